"""Adversarial record differential on the CPU emulator (tests/emu: the product's kernel sources, lane by lane) against the oracle:
packets at random instants, per-packet levels 3..43 dB over the noise, carrier offsets to +-75 kHz, payloads to 2745 bits, stronger
packets on adjacent channels, back-to-back and overlapping packets on one channel, LE adverts, squelch 5 / 10 / 14 dB, sniffer and
LAP mode, 8 / 20 / 100 Msps (tests/adversarial.py).  A planted record counts as identical only with all six key fields equal
(slot, channel, kind, offset, LAP, ac_errors; tests/paritylib.py).

    python scripts/emu_fuzz_adversarial.py CASES SEED [FIRST STRIDE] [--rates 8,8,20,100] [--only CASE] [--min-snr 3] [--wide]
      (case index = FIRST + k * STRIDE: run STRIDE processes with FIRST = 0..STRIDE-1; the last line of each is a JSON total)
Environment: EMU_VERIFY=0 runs the polyphase trajectory alone (what the exact stage is there to repair), EMU_LIB names another
build of tests/emu/libemu_bank.so.
"""
import argparse, collections, ctypes, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import pyoracle as po
import paritylib, adversarial
from tests.conftest import load_pkg
load_pkg()

ap = argparse.ArgumentParser()
ap.add_argument("cases", type=int); ap.add_argument("seed", type=int)
ap.add_argument("first", type=int, nargs="?", default=0); ap.add_argument("stride", type=int, nargs="?", default=1)
ap.add_argument("--rates", default="8,8,20,100"); ap.add_argument("--only", type=int, default=None)
ap.add_argument("--min-snr", type=float, default=3.0); ap.add_argument("--quiet", action="store_true")
ap.add_argument("--wide", action="store_true", help="companion packets over the stretched ranges (tests/adversarial.py)")
a = ap.parse_args()
L = ctypes.CDLL(os.environ.get("EMU_LIB", os.path.join(ROOT, "tests", "emu", "libemu_bank.so")))
F, Q, D = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double)
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, F, ctypes.c_longlong, ctypes.c_int, Q, D, ctypes.c_int]
L.emu_verify_counts.argtypes = [ctypes.POINTER(ctypes.c_uint)]
L.emu_verify_tasks.restype = ctypes.c_int
L.emu_verify_tasks.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
if os.environ.get("EMU_VERIFY") is not None:
    L.emu_set_verify(int(os.environ["EMU_VERIFY"]))
rates = tuple(int(x) for x in a.rates.split(","))
rng = np.random.default_rng(a.seed)
tot = collections.Counter()
nsym_dev_max = 0
for case in range(a.cases):
    c = adversarial.draw_case(rng, rates)
    if (a.only is not None and case != a.only) or case % a.stride != a.first % a.stride:
        continue
    fs, fc = c["fs"], c["fc"]
    le = c["le"] and c["sniffer"]
    iq, truth, meta = adversarial.make_adversarial_capture(fs, fc, c["n_slots"], c["n_packets"], c["seed"], c["laps"],
                                                          le_channels=c["le_channels"] if le else None, n_adverts=c["n_adverts"],
                                                          min_snr_db=a.min_snr, lag_slots=6.4 if c["sniffer"] else 1.5, wide=a.wide)
    mode = po.MODE_SNIFFER if c["sniffer"] else po.MODE_LAP
    o = po.Oracle(fs, fc, c["squelch"], mode, le=le)
    want, _ = o.run_stream(iq, threads=1)
    x = np.ascontiguousarray(np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])).view(np.float32)
    cap = 16384
    rec = np.zeros((cap, 8), np.int64); snr = np.zeros(cap)
    n = L.emu_front_m_run(fs, fc, mode, int(le), c["squelch"], x.ctypes.data_as(F), len(x) // 2, c["n_slots"], rec.ctypes.data_as(Q), snr.ctypes.data_as(D), cap)
    assert 0 <= n <= cap, n
    vc = (ctypes.c_uint * 8)(); L.emu_verify_counts(vc)
    assert vc[2] == 0, "task list overflow"
    tw = (ctypes.c_int * 65536)(); tr = (ctypes.c_int * 65536)(); nt = L.emu_verify_tasks(tw, tr, 65536)
    gi = rec[:n, :7]
    wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
    d = paritylib.differential(gi, wi, truth, lag=6 if c["sniffer"] else 1)
    AA = 0x8E89BED6
    ga = collections.Counter(map(tuple, gi[(gi[:, 2] == 1) & (gi[:, 4] == AA)][:, :6].tolist()))
    wa = collections.Counter(map(tuple, wi[(wi[:, 2] == 1) & (wi[:, 4] == AA)][:, :6].tolist()))
    adv_diff = sum(((ga - wa) + (wa - ga)).values())
    bad = d["planted_only_gpu"] + d["planted_only_ref"] + adv_diff
    tot["cases"] += 1; tot["failed_cases"] += bad > 0
    tot["planted"] += d["planted_ref"]; tot["planted_only_product"] += d["planted_only_gpu"]; tot["planted_only_oracle"] += d["planted_only_ref"]
    tot["planted_offset_differs"] += d["planted_offset_differs"]
    tot["adverts"] += sum(wa.values()); tot["adverts_differing"] += adv_diff
    tot["other_product"] += d["other_gpu"]; tot["other_oracle"] += d["other_ref"]
    tot["other_only_product"] += d["other_only_gpu"]; tot["other_only_oracle"] += d["other_only_ref"]
    tot["packets"] += len(meta); tot["windows"] += c["n_slots"] * (o.high_ch - o.low_ch + 1)
    tot["tasks"] += int(vc[0]) + int(vc[3]); tot["task_rows"] += 1250 * (int(vc[4]) + int(vc[5])) // 11   # busy windows + second-run windows; exact rows computed
    tot["planted_%dM" % round(fs / 1e6)] += d["planted_ref"]
    nsym_dev_max = max(nsym_dev_max, d["planted_nsym_max_abs_dev"])
    # which packet each of the oracle's planted records belongs to: level band and constellation
    lag = 6 if c["sniffer"] else 1
    slot_len = 625 * int(round(fs / 1e6))
    def packet_of(r):
        cand = [m for m in meta if m["channel"] == r[1] and m["lap"] == r[4] and abs(m["start"] // slot_len - (r[0] - lag)) <= 1]
        return min(cand, key=lambda m: abs(m["start"] // slot_len - (r[0] - lag))) if cand else None
    for r in wi[paritylib.classify(wi, truth, lag)]:
        m = packet_of(r)
        if m:
            tot["band_%02d" % (int(m["snr_db"]) // 5 * 5)] += 1; tot["kind_" + m["kind"]] += 1
    if d["planted_only_gpu"] + d["planted_only_ref"]:
        pgs = collections.Counter(map(tuple, gi[paritylib.classify(gi, truth, lag)][:, :6].tolist()))
        prs = collections.Counter(map(tuple, wi[paritylib.classify(wi, truth, lag)][:, :6].tolist()))
        for side, rows in (("product", pgs - prs), ("oracle", prs - pgs)):
            for r in rows.elements():
                m = packet_of((r[0], r[1], r[2], r[3], r[4]))
                tot["onesided_band_%02d" % (int(m["snr_db"]) // 5 * 5) if m else "onesided_band_none"] += 1
                tot["onesided_kind_" + (m["kind"] if m else "none")] += 1
                print("ONESIDED %s case %d rec %s packet %s" % (side, case, r, m), flush=True)
    if bad or a.only is not None:
        gs = collections.Counter(map(tuple, gi[:, :6].tolist())); ws = collections.Counter(map(tuple, wi[:, :6].tolist()))
        print("FAIL case %d (seed %d, rates %s): fs %.0fM sniffer %d le %d squelch %.0f  planted %d only product/oracle %d/%d  adverts differing %d\n"
              "   only product: %s\n   only oracle : %s" % (case, a.seed, a.rates, fs / 1e6, c["sniffer"], le, c["squelch"], d["planted_ref"],
              d["planted_only_gpu"], d["planted_only_ref"], adv_diff, sorted((gs - ws).elements()), sorted((ws - gs).elements())), flush=True)
        if a.only is not None:
            for m in meta:
                print("   packet", m)
    if not a.quiet:
        print("case %5d fs %3.0fM sniffer %d le %d sq %2.0f packets %3d  planted %3d one-sided %d/%d nsym-dev %2d  adverts %d/%d  other %d/%d one-sided %d/%d  tasks %d rows %d"
              % (case, fs / 1e6, c["sniffer"], le, c["squelch"], len(meta), d["planted_ref"], d["planted_only_gpu"], d["planted_only_ref"],
                 d["planted_nsym_max_abs_dev"], sum(ga.values()), sum(wa.values()), d["other_gpu"], d["other_ref"], d["other_only_gpu"], d["other_only_ref"],
                 int(vc[0]) + int(vc[3]), 1250 * (int(vc[4]) + int(vc[5])) // 11), flush=True)
out = dict(tot); out["nsym_dev_max"] = nsym_dev_max; out["seed"] = a.seed; out["rates"] = a.rates; out["min_snr_db"] = a.min_snr; out["wide"] = a.wide
print("TOTAL " + json.dumps(out))
