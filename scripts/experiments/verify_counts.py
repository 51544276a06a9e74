"""How many windows the exact stage takes on a capture (emulator): python scripts/experiments/verify_counts.py rateMHz slots sniff occ snr [seed]"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
r = int(sys.argv[1]); nsl = int(sys.argv[2]); sniff = int(sys.argv[3]); occ = float(sys.argv[4]); snr_db = float(sys.argv[5])
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 1
FC = {8: 2476.5e6, 20: 2441e6, 100: 2441e6}
fs, fc = r * 1e6, FC[r]
iq, truth = synth.make_capture(fs, fc, nsl, laps=(0x24D952, 0x4831DD, 0x123456), seed=seed, snr_db=snr_db, occupancy=occ, cfo_hz=float(os.environ.get("CFO", "10e3")), max_payload_bits=int(os.environ.get("MAXPAY", "240")))
mode = po.MODE_SNIFFER if sniff else po.MODE_LAP
o = po.Oracle(fs, fc, 10.0, mode)
x = np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])
xf = np.ascontiguousarray(x).view(np.float32)
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float),
                              ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
cap = 1 << 16
rec = np.zeros((cap, 8), np.int64); snr = np.zeros(cap, np.float64)
n = L.emu_front_m_run(fs, fc, mode, 0, 10.0, xf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x), nsl,
                      rec.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), snr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
vc = (ctypes.c_uint * 8)(); L.emu_verify_counts(vc)
nw = nsl * (o.high_ch - o.low_ch + 1)
print("windows %d, bursts planted %d, records %d, verified windows %d (%.1f %%), tiles %d" % (nw, len(truth), n, vc[0], 100.0 * vc[0] / nw, vc[1]))

wv = (ctypes.c_int * 65536)(); rv = (ctypes.c_int * 65536)()
nt = L.emu_verify_tasks(wv, rv, 65536)
nch = o.high_ch - o.low_ch + 1
lag = 6 if sniff else 1
tset = {}
for t in truth:
    tset.setdefault((t["slot"], t["channel"]), []).append(t)
recs = {(int(r[0]), int(r[1])) for r in rec[:n]}
kinds = {"burst": 0, "burst+rec": 0, "adjacent": 0, "early": 0, "other": 0}
for i in range(nt):
    k, c = wv[i] // nch, wv[i] % nch + o.low_ch
    near = [(dk, dc) for dk in (-1, 0, 1) for dc in (-1, 0, 1) if (k - lag + dk, c + dc) in tset]
    if any(dc == 0 for dk, dc in near): kind = "burst+rec" if (k, c) in recs else "burst"
    elif near: kind = "adjacent"
    elif k <= lag: kind = "early"
    else: kind = "other"
    kinds[kind] += 1
    if kind == "burst" and os.environ.get("SHOW_BURST"): print("   burst-no-rec slot %d ch %d rows %d truth" % (k, c, rv[i]), [(dk, tset[(k - lag + dk, c)][0]["start"] % int(o.slot), tset[(k - lag + dk, c)][0]["nbits"]) for dk in (-2, -1, 0, 1) if (k - lag + dk, c) in tset])
    if kind in ("other", "adjacent") and i < 400: print("   flagged slot %d ch %d rows %d (%s)" % (k, c, rv[i], kind))
print(kinds)
if os.environ.get("SHOW_RECS"):
    for r_ in rec[:min(n, 6)]: print("   rec", r_[:7], "w =", int(r_[0]) * nch + int(r_[1]) - o.low_ch)
tasks = {int(wv[i]) for i in range(nt)}
miss = [t for t in truth if t["slot"] + lag < nsl and (t["slot"] + lag) * nch + t["channel"] - o.low_ch not in tasks]
print("planted bursts whose window is not a task:", len(miss))
for t in miss[:12]: print("   ", t, "w =", (t["slot"] + lag) * nch + t["channel"] - o.low_ch)
if os.environ.get("SHOW_TASKS"):
    print(sorted((int(wv[i]) // nch, int(wv[i]) % nch, int(rv[i])) for i in range(nt))[:40])
