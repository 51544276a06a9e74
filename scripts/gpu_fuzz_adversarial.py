"""The adversarial record differential of scripts/emu_fuzz_adversarial.py (tests/adversarial.py's generator) on the DEVICE, through the
C ABI, against the all-core oracle.  Run on the GPU box:
    python scripts/gpu_fuzz_adversarial.py CASES SEED [--rates 8,8,20,100] [--wide] [--seconds S] [--no-torch]   (S: stop after S seconds of cases)
Planted records identical on the six key fields (tests/paritylib.py), LE adverts identical; the last line is a JSON total."""
import argparse, collections, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
if "--no-torch" in sys.argv:                  # libbtgpu.so then binds /opt/rocm's HIP runtime itself; saves torch's minute-long first import on a fresh box
    sys.argv.remove("--no-torch"); sys.modules["torch"] = None
import numpy as np
import pyoracle as po
import paritylib, adversarial
from tests.conftest import load_pkg
pkg = load_pkg()
bdist = importlib.import_module("gr_bluetooth_amd.dist")
ap = argparse.ArgumentParser(); ap.add_argument("cases", type=int); ap.add_argument("seed", type=int); ap.add_argument("--rates", default="8,8,20,100")
ap.add_argument("--wide", action="store_true"); ap.add_argument("--seconds", type=float, default=0.0)
a = ap.parse_args()
t0 = time.time()
rates = tuple(int(x) for x in a.rates.split(","))
rng = np.random.default_rng(a.seed)
tot = collections.Counter(); nsym_dev = 0
for case in range(a.cases):
    if a.seconds and time.time() - t0 > a.seconds:
        break
    c = adversarial.draw_case(rng, rates)
    fs, fc = c["fs"], c["fc"]; le = c["le"] and c["sniffer"]
    iq, truth, meta = adversarial.make_adversarial_capture(fs, fc, c["n_slots"], c["n_packets"], c["seed"], c["laps"], le_channels=c["le_channels"] if le else None,
                                                          n_adverts=c["n_adverts"], lag_slots=6.4 if c["sniffer"] else 1.5, wide=a.wide)
    want, _ = po.Oracle(fs, fc, c["squelch"], po.MODE_SNIFFER if c["sniffer"] else po.MODE_LAP, le=le).run_stream(iq, threads=os.cpu_count() or 1)
    blk = pkg.multi_sniffer(fs, fc, c["squelch"], False, le=le) if c["sniffer"] else pkg.multi_LAP(fs, fc, c["squelch"])
    blk.push(iq); got = blk.poll(); tm = blk.timing(); blk.close()
    gi, _ = bdist.hits_to_arrays(got); wi, _ = bdist.hits_to_arrays(want)
    d = paritylib.differential(gi, wi, truth, lag=6 if c["sniffer"] else 1)
    AA = 0x8E89BED6
    ga = collections.Counter(map(tuple, gi[(gi[:, 2] == 1) & (gi[:, 4] == AA)][:, :6].tolist())) if len(gi) else collections.Counter()
    wa = collections.Counter(map(tuple, wi[(wi[:, 2] == 1) & (wi[:, 4] == AA)][:, :6].tolist())) if len(wi) else collections.Counter()
    adv = sum(((ga - wa) + (wa - ga)).values())
    bad = d["planted_only_gpu"] + d["planted_only_ref"] + adv
    tot["cases"] += 1; tot["failed_cases"] += bad > 0; tot["planted"] += d["planted_ref"]
    tot["planted_only_product"] += d["planted_only_gpu"]; tot["planted_only_oracle"] += d["planted_only_ref"]
    tot["adverts"] += sum(wa.values()); tot["adverts_differing"] += adv
    tot["other_product"] += d["other_gpu"]; tot["other_oracle"] += d["other_ref"]; tot["other_only_product"] += d["other_only_gpu"]; tot["other_only_oracle"] += d["other_only_ref"]
    tot["tasks"] += int(tm.verify_windows); tot["turned_away"] += int(tm.verify_turned_away); tot["planted_%dM" % round(fs / 1e6)] += d["planted_ref"]
    nsym_dev = max(nsym_dev, d["planted_nsym_max_abs_dev"])
    if bad:
        gs = collections.Counter(map(tuple, gi[:, :6].tolist())); ws = collections.Counter(map(tuple, wi[:, :6].tolist()))
        print("FAIL case %d (seed %d, rates %s): only product %s only oracle %s" % (case, a.seed, a.rates, sorted((gs - ws).elements()), sorted((ws - gs).elements())), flush=True)
out = dict(tot); out["nsym_dev_max"] = nsym_dev; out["seed"] = a.seed; out["rates"] = a.rates; out["wide"] = a.wide
print("TOTAL " + json.dumps(out))
