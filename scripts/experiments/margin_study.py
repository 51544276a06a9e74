"""Round-4 design study (no GPU): how far the polyphase (FAST) demodulated stream's clock-recovery trajectory is from the
exact one, per window, and which decision margins would have announced a parting (tests/emu emu_margin_study).
    python scripts/experiments/margin_study.py [cases] [seed] [nsyms] [rates: e.g. 8,20]"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
nsyms = int(sys.argv[3]) if len(sys.argv) > 3 else 693
rates = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [8, 20]
FC = {8: 2476.5e6, 20: 2441e6, 100: 2441e6}
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_margin_study.restype = ctypes.c_int
L.emu_margin_study.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong,
                               ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
allrows = []
for case in range(cases):
    r = rates[int(rng.integers(0, len(rates)))]; fs, fc = r * 1e6, FC[r]
    nsl = int(rng.integers(8, 14)); snr_db = float(rng.uniform(12, 30)); occ = float(rng.uniform(0.2, 0.9))
    sq = float(rng.choice([5.0, 10.0, 14.0])); sniff = bool(rng.integers(0, 2))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 6))
    iq, truth = synth.make_capture(fs, fc, nsl, laps=laps, seed=int(rng.integers(0, 1 << 30)), snr_db=snr_db, occupancy=occ)
    mode = po.MODE_SNIFFER if sniff else po.MODE_LAP
    o = po.Oracle(fs, fc, sq, mode)
    x = np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])
    xf = np.ascontiguousarray(x).view(np.float32)
    cap = 1 << 16
    rows = np.zeros((cap, 12), np.float64)
    n = L.emu_margin_study(fs, fc, mode, sq, xf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x), nsl, nsyms,
                           rows.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
    assert n >= 0, n
    rows = rows[:n]
    allrows.append(rows)
    both = rows[rows[:, 11] == 3]
    parted = both[both[:, 10] != 0]
    print("case %d fs %dM sniff %d sq %.0f snr %.1f: windows %d (squelch one-sided %d) parted %d; max dev out %.2e mu %.2e" %
          (case, r, sniff, sq, snr_db, n, int((rows[:, 11] != 3).sum()), len(parted),
           both[:, 4].max() if len(both) else 0, both[:, 5].max() if len(both) else 0), flush=True)
R = np.concatenate(allrows)
np.save("/tmp/margin_rows_%s.npy" % "_".join(map(str, rates)), R)
both = R[R[:, 11] == 3]
parted = both[both[:, 10] != 0]
print("windows %d, parted %d (%.3f %%)" % (len(both), len(parted), 100.0 * len(parted) / max(1, len(both))))
print("deviation |out| quantiles 50/90/99/99.9/max:", np.quantile(both[:, 4], [.5, .9, .99, .999, 1]))
print("deviation  mu   quantiles 50/90/99/99.9/max:", np.quantile(both[:, 5], [.5, .9, .99, .999, 1]))
if len(parted):
    print("at the parting symbol: |out| quantiles", np.quantile(parted[:, 8], [0, .5, .9, .99, 1]), " rounding margin", np.quantile(parted[:, 9], [0, .5, .9, .99, 1]))
    print("what parted (1 symbol, 2 step, 4 index):", np.unique(parted[:, 10], return_counts=True))
    m = np.minimum(parted[:, 8] / 1.0, 1e9)
    for eo, em in [(1e-5, 1e-3), (3e-5, 3e-3), (1e-4, 1e-2), (3e-4, 3e-2), (1e-3, 0.1)]:
        caught = ((parted[:, 8] < eo) | (parted[:, 9] > 0.5 - em))
        flagged = ((both[:, 6] < eo) | (both[:, 7] > 0.5 - em) | (both[:, 10] != 0))
        print("eps_out %.0e eps_mu %.0e (1/128 steps): catches %d of %d partings; windows flagged before parting/end %.2f %%" %
              (eo, em, caught.sum(), len(parted), 100.0 * flagged.mean()))
