"""Added by the round-4 JUDGE (not by the builder).  Near-far differential on the CPU emulator: every capture is noise plus
packets at random instants (not slot-aligned), random per-packet levels, and -- for 70 % of them -- a packet 14-32 dB STRONGER on
an adjacent channel that overlaps it in time.  Product kernel sources (tests/emu) against the oracle, tests/paritylib.py.
    python3 scripts/experiments/judge_r04_nearfar_emu.py 100 36 21 35     # 100 Msps, seed 21, only case 35: one planted record
                                                                          #   (slot 6, channel 44, offset 235, LAP a06302, 4 errors)
                                                                          #   found by the oracle only -- kernels.hip.h:678 (leak rule)
    python3 scripts/experiments/judge_r04_nearfar_emu.py mix 959 11 958   # 8 / 20 Msps, seed 11, case 958: nsym 10 apart (bound says 8)
    python3 scripts/experiments/judge_r04_nearfar_emu.py mix 2400 11      # 2400 captures, 15 346 planted records, none deviating (~6 min, 1 core)
"""
import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np, pyoracle as po, paritylib
from tests.conftest import load_pkg
load_pkg(); synth = importlib.import_module("gr_bluetooth_amd.synth")
L = ctypes.CDLL(ROOT + "/tests/emu/libemu_bank.so"); F = ctypes.POINTER(ctypes.c_float); Q = ctypes.POINTER(ctypes.c_longlong)
D = ctypes.POINTER(ctypes.c_double)
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, F, ctypes.c_longlong, ctypes.c_int, Q, D, ctypes.c_int]
mode = sys.argv[1] if len(sys.argv) > 1 else "100"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 21)
only = int(sys.argv[4]) if len(sys.argv) > 4 else None
RATES = [(100e6, 2441e6)] if mode == "100" else [(8e6, 2476.5e6), (8e6, 2476.5e6), (20e6, 2441e6)]
tot = dict(cases=0, planted=0, only_product=0, only_oracle=0, offset_differs=0, nsym_dev_max=0)
for case in range(cases):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 12)); base = float(rng.uniform(14, 30)); sq = float(rng.choice([5.0, 10.0]))
    nb = int(rng.integers(40, 120)) if mode == "100" else int(rng.integers(10, 40))
    seed = int(rng.integers(0, 1 << 30)); cfo = float(rng.choice([10e3, 40e3, 60e3]))
    spread = float(rng.choice([0.0, 6.0, 12.0])); laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 5))
    if only is not None and case != only:
        continue
    sps = int(round(fs / 1e6)); slot = 625 * sps; lo, hi = synth.visible_channels(fs, fc)
    r2 = np.random.default_rng(seed); truth = []
    iq, _ = synth.make_capture(fs, fc, nsl, laps=laps, seed=seed, snr_db=base, occupancy=0.0)       # noise only
    for _ in range(nb):
        lap = int(r2.choice(laps)); ch = int(r2.integers(lo, hi + 1)); start = int(r2.integers(0, (nsl - 1) * slot))
        a = float(r2.uniform(-spread, 0.0)) + float(r2.choice([0.0, 0.0, 6.0])); bits = synth.packet_bits(lap, r2, int(r2.integers(0, 1200)))
        synth.add_burst(iq, bits, start, fs, fc, ch, r2, cfo_hz=cfo, amplitude=10 ** (a / 20))
        truth.append(dict(slot=start // slot, channel=ch, lap=lap))
        if r2.random() < 0.7:                                                 # a stronger packet on an adjacent channel, overlapping in time
            ch2 = ch + (1 if (ch < hi and (ch == lo or r2.random() < 0.5)) else -1); lap2 = int(r2.choice(laps)); up = float(r2.uniform(14, 32))
            st2 = max(0, start + int(r2.integers(-60 * sps, 60 * sps))); b2 = synth.packet_bits(lap2, r2, int(r2.integers(0, 600)))
            synth.add_burst(iq, b2, st2, fs, fc, ch2, r2, cfo_hz=cfo, amplitude=10 ** ((a + up) / 20))
            truth.append(dict(slot=st2 // slot, channel=ch2, lap=lap2))
    o = po.Oracle(fs, fc, sq, po.MODE_SNIFFER, le=False); want, _ = o.run_stream(iq, threads=1)
    x = np.ascontiguousarray(np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])).view(np.float32)
    rec = np.zeros((8192, 8), np.int64); snr = np.zeros(8192)
    n = L.emu_front_m_run(fs, fc, po.MODE_SNIFFER, 0, sq, x.ctypes.data_as(F), len(x) // 2, nsl, rec.ctypes.data_as(Q), snr.ctypes.data_as(D), 8192)
    wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
    d = paritylib.differential(rec[:n, :7], wi, truth, lag=6)
    tot["cases"] += 1; tot["planted"] += d["planted_ref"]; tot["only_product"] += d["planted_only_gpu"]; tot["only_oracle"] += d["planted_only_ref"]
    tot["offset_differs"] += d["planted_offset_differs"]; tot["nsym_dev_max"] = max(tot["nsym_dev_max"], d["planted_nsym_max_abs_dev"])
    bad = d["planted_only_gpu"] or d["planted_only_ref"] or d["planted_offset_differs"] or d["planted_nsym_max_abs_dev"] > 8
    if bad or only is not None:
        gs, ws = set(map(tuple, rec[:n, :6].tolist())), set(map(tuple, wi[:, :6].tolist()))
        print("case %d fs %.0fM planted %d only product/oracle %d/%d offset differs %d nsym dev %d\n   only product: %s\n   only oracle : %s" %
              (case, fs / 1e6, d["planted_ref"], d["planted_only_gpu"], d["planted_only_ref"], d["planted_offset_differs"],
               d["planted_nsym_max_abs_dev"], sorted(gs - ws), sorted(ws - gs)), flush=True)
print("TOTAL", tot)
