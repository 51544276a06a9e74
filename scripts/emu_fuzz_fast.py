"""The randomised differential run of scripts/gpu_fuzz_fast.py with the CPU emulator (tests/emu: the product's kernel
sources run lane by lane on the host) in place of the device -- no GPU needed.  Same case generator, same classification
(tests/paritylib.py).  Cases 312, 626 and 743 of seed 77 were checked to give on the emulator exactly what the MI355X gives
(DESIGN.md section 5); LE is off in this run (the emulated front end takes the flag, the generator draws it as the GPU script does).
    python scripts/emu_fuzz_fast.py [cases] [seed] [first_case] [stride] [start]   (case index = first_case + k * stride: run
                                                                                    `stride` processes with first_case 0..stride-1)
"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import paritylib
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
start = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # resume: skip the cases below this index
RATES = [(100e6, 2441e6), (8e6, 2476.5e6), (20e6, 2441e6), (100e6, 2441e6)]
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float),
                              ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
L.emu_verify_counts.argtypes = [ctypes.POINTER(ctypes.c_uint)]
if os.environ.get("EMU_VERIFY") is not None:                    # 0: exact confirmation off (the round-3 behaviour), 2: hit windows only
    L.emu_set_verify(int(os.environ["EMU_VERIFY"]))
vc = (ctypes.c_uint * 8)()
tot = dict(verified_windows=0, windows=0, cases=0, planted=0, planted_differing=0, planted_offset_differs=0, other_emu=0, other_ref=0, other_only_emu=0,
           other_only_ref=0, nsym_dev_max=0, failed=0)
for case in range(cases):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 14)); snr_db = float(rng.uniform(12, 30)); occ = float(rng.uniform(0.2, 0.9))
    sq = float(rng.choice([5.0, 10.0, 14.0])); sniff = bool(rng.integers(0, 2)); le = sniff and bool(rng.integers(0, 2))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 6))
    seed = int(rng.integers(0, 1 << 30))
    if case % stride != first % stride or case < start:
        continue
    iq, truth = synth.make_capture(fs, fc, nsl, laps=laps, seed=seed, snr_db=snr_db, occupancy=occ)
    mode = po.MODE_SNIFFER if sniff else po.MODE_LAP
    o = po.Oracle(fs, fc, sq, mode, le=le)
    want, _ = o.run_stream(iq, threads=1)
    x = np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])
    xf = np.ascontiguousarray(x).view(np.float32)
    cap = 8192
    rec = np.zeros((cap, 8), np.int64); snr = np.zeros(cap, np.float64)
    n = L.emu_front_m_run(fs, fc, mode, int(le), sq, xf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x), nsl,
                          rec.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), snr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
    assert 0 <= n <= cap, n
    L.emu_verify_counts(vc); tot["verified_windows"] += int(vc[0]) + int(vc[3]); tot["windows"] += nsl * (o.high_ch - o.low_ch + 1)
    assert vc[2] == 0, "task list overflow"
    gi = rec[:n, :7]
    wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
    d = paritylib.differential(gi, wi, truth, lag=6 if sniff else 1)
    ok = d["planted_identical"] and d["planted_offset_differs"] == 0 and d["planted_nsym_max_abs_dev"] <= paritylib.NSYM_BOUND
    tot["cases"] += 1; tot["failed"] += not ok
    tot["planted"] += d["planted_ref"]; tot["planted_differing"] += d["planted_only_gpu"] + d["planted_only_ref"]
    tot["planted_offset_differs"] += d["planted_offset_differs"]
    tot["other_emu"] += d["other_gpu"]; tot["other_ref"] += d["other_ref"]
    tot["other_only_emu"] += d["other_only_gpu"]; tot["other_only_ref"] += d["other_only_ref"]
    tot["nsym_dev_max"] = max(tot["nsym_dev_max"], d["planted_nsym_max_abs_dev"])
    if not ok:
        gs, ws = set(map(tuple, gi[:, :6].tolist())), set(map(tuple, wi[:, :6].tolist()))
        print("   only emu   :", sorted(gs - ws)); print("   only oracle:", sorted(ws - gs))
    print("case %4d fs %3.0fM sniff %d le %d sq %4.1f snr %4.1f occ %.2f slots %2d  planted %3d identical %s offset-differs %d nsym-dev %d  other emu/ref %d/%d one-sided %d/%d  verified %d" %
          (case, fs / 1e6, sniff, le, sq, snr_db, occ, nsl, d["planted_ref"], d["planted_identical"], d["planted_offset_differs"],
           d["planted_nsym_max_abs_dev"], d["other_gpu"], d["other_ref"], d["other_only_gpu"], d["other_only_ref"], int(vc[0]) + int(vc[3])), flush=True)
print("TOTAL", tot)
