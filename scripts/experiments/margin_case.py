"""Margin study on ONE case of scripts/emu_fuzz_fast.py's generator: python scripts/experiments/margin_case.py seed case [nsyms]"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
rng = np.random.default_rng(int(sys.argv[1])); want_case = int(sys.argv[2]); nsyms = int(sys.argv[3]) if len(sys.argv) > 3 else 693
RATES = [(100e6, 2441e6), (8e6, 2476.5e6), (20e6, 2441e6), (100e6, 2441e6)]
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_margin_study.restype = ctypes.c_int
L.emu_margin_study.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong,
                               ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
for case in range(want_case + 1):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 14)); snr_db = float(rng.uniform(12, 30)); occ = float(rng.uniform(0.2, 0.9))
    sq = float(rng.choice([5.0, 10.0, 14.0])); sniff = bool(rng.integers(0, 2)); le = sniff and bool(rng.integers(0, 2))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 6))
    seed = int(rng.integers(0, 1 << 30))
iq, truth = synth.make_capture(fs, fc, nsl, laps=laps, seed=seed, snr_db=snr_db, occupancy=occ)
mode = po.MODE_SNIFFER if sniff else po.MODE_LAP
o = po.Oracle(fs, fc, sq, mode, le=le)
x = np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])
xf = np.ascontiguousarray(x).view(np.float32)
cap = 1 << 16
rows = np.zeros((cap, 12), np.float64)
n = L.emu_margin_study(fs, fc, mode, sq, xf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x), nsl, nsyms,
                       rows.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
rows = rows[:n]
print("fs %g sniff %d sq %g snr %.1f slots %d low_channel %d windows %d" % (fs, sniff, sq, snr_db, nsl, o.low_channel if hasattr(o, 'low_channel') else -1, n))
np.set_printoptions(linewidth=250, precision=3, suppress=False)
for r in rows:
    if r[11] != 3 or r[10] != 0:
        print("slot %d ch %d parted at %d of %d  dev out %.2e mu %.2e  min|out| %.2e minround %.3f  at parting |out| %.2e round %.4f what %d squelch %d" %
              tuple(r[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]]))
