"""One case of scripts/emu_fuzz_fast.py in detail: the records one side has and the other has not (planted or not).
    python scripts/experiments/other_case.py <seed> <case> [EMU_VERIFY]"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
rng = np.random.default_rng(int(sys.argv[1])); target = int(sys.argv[2])
RATES = [(100e6, 2441e6), (8e6, 2476.5e6), (20e6, 2441e6), (100e6, 2441e6)]
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float),
                              ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
if len(sys.argv) > 3: L.emu_set_verify(int(sys.argv[3]))
for case in range(target + 1):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 14)); snr_db = float(rng.uniform(12, 30)); occ = float(rng.uniform(0.2, 0.9))
    sq = float(rng.choice([5.0, 10.0, 14.0])); sniff = bool(rng.integers(0, 2)); le = sniff and bool(rng.integers(0, 2))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 6))
    seed = int(rng.integers(0, 1 << 30))
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
gi = rec[:n, :7]
wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
gs, ws = set(map(tuple, gi[:, :6].tolist())), set(map(tuple, wi[:, :6].tolist()))
print("fs", fs, "sniff", sniff, "le", le, "sq", sq, "snr", snr_db, "nsl", nsl, "records", len(gs), len(ws))
print("(slot, channel, kind, offset, lap, ac_errors)")
print("only emu   :", sorted(gs - ws)); print("only oracle:", sorted(ws - gs))
print("truth:", [t for t in truth][:60])
