"""Randomised captures with PLANTED LE advertising packets (the fuzz generator plants classic packets only) through the emulated
polyphase front end with the exact stage against the oracle: every record at a planted advert -- kind 1, access address
0x8E89BED6 -- must be identical on both sides (slot, channel, offset, AA); the classic planted records as in emu_fuzz_fast.py.
    python scripts/experiments/emu_le_adverts.py [cases] [seed] [first] [stride]"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import paritylib
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu_bank.so"))
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float),
                              ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
RATES = [(8e6, 2476.5e6, {78: 39}), (100e6, 2441e6, {0: 37, 24: 38, 78: 39}), (8e6, 2476.5e6, {78: 39})]
tot = dict(cases=0, adverts_ref=0, adverts_emu=0, adverts_differing=0, classic_planted=0, classic_differing=0, other_only_emu=0, other_only_ref=0)
for case in range(cases):
    fs, fc, lech = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 13)); snr_db = float(rng.uniform(12, 28)); occ = float(rng.uniform(0.1, 0.6))
    sq = float(rng.choice([5.0, 10.0])); seed = int(rng.integers(0, 1 << 30)); nadv = int(rng.integers(4, 10))
    cfo = float(rng.uniform(0, 60e3))                      # add_burst draws the offset in [-cfo, cfo]
    amp = float(10 ** (rng.uniform(-6, 6) / 20))
    if case % stride != first % stride:
        continue
    iq, truth = synth.make_capture(fs, fc, nsl, laps=(0x24D952, 0x4831DD, 0x9E8B33), seed=seed, snr_db=snr_db, occupancy=occ)
    r2 = np.random.default_rng(seed + 1)
    slot = int(round(fs * 625e-6))
    for _ in range(nadv):
        ch = int(r2.choice(list(lech)))
        k = int(r2.integers(0, nsl - 1))
        synth.add_burst(iq, synth.le_advert_bits(lech[ch], r2, payload_bytes=int(r2.integers(6, 30))),
                        k * slot + int(r2.integers(0, slot - 1)), fs, fc, ch, r2, cfo_hz=cfo, amplitude=amp)
    o = po.Oracle(fs, fc, sq, po.MODE_SNIFFER, le=True)
    want, _ = o.run_stream(iq, threads=1)
    x = np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])
    xf = np.ascontiguousarray(x).view(np.float32)
    cap = 8192
    rec = np.zeros((cap, 8), np.int64); snr = np.zeros(cap, np.float64)
    n = L.emu_front_m_run(fs, fc, po.MODE_SNIFFER, 1, sq, xf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x), nsl,
                          rec.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), snr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cap)
    assert 0 <= n <= cap
    gi = rec[:n, :7]
    wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
    d = paritylib.differential(gi, wi, truth, lag=6)
    AA = 0x8E89BED6
    ga = set(map(tuple, gi[(gi[:, 2] == 1) & (gi[:, 4] == AA)][:, :6].tolist()))
    wa = set(map(tuple, wi[(wi[:, 2] == 1) & (wi[:, 4] == AA)][:, :6].tolist()))
    tot["cases"] += 1; tot["adverts_ref"] += len(wa); tot["adverts_emu"] += len(ga); tot["adverts_differing"] += len(ga ^ wa)
    tot["classic_planted"] += d["planted_ref"]; tot["classic_differing"] += d["planted_only_gpu"] + d["planted_only_ref"] + d["planted_offset_differs"]
    tot["other_only_emu"] += d["other_only_gpu"]; tot["other_only_ref"] += d["other_only_ref"]
    print("case %3d fs %3.0fM sq %4.1f snr %4.1f amp %+.1f dB cfo <= %2.0f kHz adverts planted %d found emu/ref %d/%d differing %d  classic %d identical %s  other one-sided %d/%d"
          % (case, fs / 1e6, sq, snr_db, 20 * np.log10(amp), cfo / 1e3, nadv, len(ga), len(wa), len(ga ^ wa), d["planted_ref"],
             d["planted_identical"] and d["planted_offset_differs"] == 0, d["other_only_gpu"], d["other_only_ref"]), flush=True)
    if ga ^ wa:
        print("   only emu:", sorted(ga - wa), " only oracle:", sorted(wa - ga))
print("TOTAL", tot)
