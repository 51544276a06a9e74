"""The round-5 judge's generator (tests/adversarial.py judge_r05_seamless_case: packets that show NO step in the channel's energy where
they begin -- behind a GFSK emitter, a carrier or a noise burst of their own level, slow ramps; --weak: packets 2.5..7 dB over the
noise beside a strong neighbour) on the DEVICE, through the C ABI, against the all-core oracle.  Run on the GPU box:
    python scripts/gpu_judge_seamless.py CASES SEED [--mode mix|8|20|100] [--weak] [--seconds S]
Planted records identical on the six key fields (tests/paritylib.py); the last line is a JSON total."""
import argparse, collections, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import pyoracle as po
import paritylib, adversarial
from tests.conftest import load_pkg
pkg = load_pkg()
bdist = importlib.import_module("gr_bluetooth_amd.dist")
ap = argparse.ArgumentParser(); ap.add_argument("cases", type=int); ap.add_argument("seed", type=int); ap.add_argument("--mode", default="mix")
ap.add_argument("--weak", action="store_true"); ap.add_argument("--seconds", type=float, default=0.0)
a = ap.parse_args()
kinds = ("weak-beside",) if a.weak else adversarial.SEAMLESS_KINDS
t0 = time.time()
tot = collections.Counter(); nsym_dev = 0
for case in range(a.cases):
    if a.seconds and time.time() - t0 > a.seconds:
        break
    fs, fc, nsl, sq, iq, truth, meta = adversarial.judge_r05_seamless_case(a.mode, a.seed, case, kinds)
    want, _ = po.Oracle(fs, fc, sq, po.MODE_SNIFFER).run_stream(iq, threads=os.cpu_count() or 1)
    blk = pkg.multi_sniffer(fs, fc, sq, False)
    blk.push(iq); got = blk.poll(); tm = blk.timing(); blk.close()
    gi, _ = bdist.hits_to_arrays(got); wi, _ = bdist.hits_to_arrays(want)
    d = paritylib.differential(gi, wi, truth, lag=6)
    bad = d["planted_only_gpu"] + d["planted_only_ref"]
    tot["cases"] += 1; tot["failed_cases"] += bad > 0; tot["planted"] += d["planted_ref"]
    tot["planted_only_product"] += d["planted_only_gpu"]; tot["planted_only_oracle"] += d["planted_only_ref"]; tot["offset_differs"] += d["planted_offset_differs"]
    tot["other_product"] += d["other_gpu"]; tot["other_oracle"] += d["other_ref"]; tot["other_only_product"] += d["other_only_gpu"]; tot["other_only_oracle"] += d["other_only_ref"]
    tot["busy_windows"] += int(tm.verify_windows); tot["turned_away"] += int(tm.verify_turned_away)
    for m in meta:
        tot["packets_" + m["kind"]] += 1
    nsym_dev = max(nsym_dev, d["planted_nsym_max_abs_dev"])
    if bad:
        gs = collections.Counter(map(tuple, gi[:, :6].tolist())); ws = collections.Counter(map(tuple, wi[:, :6].tolist()))
        print("FAIL case %d (seed %d, mode %s): only product %s only oracle %s" % (case, a.seed, a.mode, sorted((gs - ws).elements()), sorted((ws - gs).elements())), flush=True)
out = dict(tot); out["nsym_dev_max"] = nsym_dev; out["seed"] = a.seed; out["mode"] = a.mode; out["kinds"] = list(kinds)
print("TOTAL " + json.dumps(out))
