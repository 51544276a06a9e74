"""Added by the round-6 JUDGE (not by the builder).  Differential on the CPU emulator (tests/emu: the product's kernel sources) against
the oracle for the classes DESIGN.md section 4.4 "What presence does not see" names itself:
  faint-id     a 68-symbol ID packet (access code alone: 68 us on the air) 4..12 dB over the noise at a random instant -- at the rates
               whose presence tiles are 125 us (4 / 10 / 16 Msps) it can straddle two tiles with a third of a tile's energy in each
  faint-pair   the same between TWO long neighbours 22..40 dB over the noise on the channels below and above (threshold 3.0 x there)
  faint-long   a faint packet (4..10 dB) with a payload, alone
    python3 scripts/experiments/judge_r06_faint_emu.py RATES CASES SEED [ONLY]      # RATES e.g. 4,10,16 | 8,20 | 100
"""
import os, sys, ctypes, importlib, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np, pyoracle as po, paritylib, adversarial
from tests.conftest import load_pkg
load_pkg(); synth = importlib.import_module("gr_bluetooth_amd.synth")
L = ctypes.CDLL(os.environ.get("EMU_LIB", ROOT + "/tests/emu/libemu_bank.so")); F = ctypes.POINTER(ctypes.c_float); Q = ctypes.POINTER(ctypes.c_longlong)
D = ctypes.POINTER(ctypes.c_double)
L.emu_front_m_run.restype = ctypes.c_int
L.emu_front_m_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, F, ctypes.c_longlong, ctypes.c_int, Q, D, ctypes.c_int]
if os.environ.get("EMU_VERIFY") is not None:
    L.emu_set_verify(int(os.environ["EMU_VERIFY"]))
rates = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,10,16").split(",")]
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
only = int(sys.argv[4]) if len(sys.argv) > 4 else None
KINDS = tuple(os.environ.get("JUDGE_KINDS", "faint-id,faint-pair,faint-long").split(","))
TOP = 43.0


def put(iq, bb, start, fs, fc, ch, f_off, ph):
    f = (synth.BASE_FREQUENCY + ch * 1e6 - fc) + f_off
    start = int(start)
    if start < 0:
        bb = bb[-start:]; start = 0
    m = np.arange(len(bb))
    bb = bb * np.exp(1j * (2 * np.pi * f / fs * m + ph))
    end = min(start + len(bb), len(iq))
    if end > start:
        iq[start:end] += bb[:end - start].astype(np.complex64)


tot = collections.Counter()
for case in range(cases):
    r = int(rng.choice(rates)); fs, fc, _le = adversarial.RATES[r]
    nsl = int(rng.integers(9, 13)); sq = float(rng.choice([5.0, 10.0])); seed = int(rng.integers(0, 1 << 30))
    npk = int(rng.integers(40, 90)) if r == 100 else int(rng.integers(10, 28))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 5))
    if only is not None and case != only:
        continue
    sps = int(round(fs / 1e6)); slot = 625 * sps; lo, hi = synth.visible_channels(fs, fc)
    r2 = np.random.default_rng(seed); truth = []; meta = []
    iq, _ = synth.make_capture(fs, fc, nsl, laps=laps, seed=seed, snr_db=TOP, occupancy=0.0)
    reach = int((nsl - 6.4) * slot)
    used = collections.defaultdict(list)
    for _ in range(npk):
        kind = str(r2.choice(KINDS)); lap = int(r2.choice(laps)); ch = int(r2.integers(lo, hi + 1))
        start = int(r2.integers(700 * sps, max(reach, 701 * sps)))
        if any(abs(start - s) < 1500 * sps for c2 in (ch - 1, ch, ch + 1) for s in used[c2]):
            continue
        used[ch].append(start)
        LO = float(os.environ.get("JUDGE_LO", "4")); HI = float(os.environ.get("JUDGE_HI", "12")); level = float(r2.uniform(LO, HI))
        amp = 10 ** ((level - TOP) / 20); cfo = float(r2.uniform(-40e3, 40e3))
        if kind == "faint-long":
            bits = synth.packet_bits(lap, r2, int(r2.integers(0, 600)))
        else:
            bits = synth.packet_bits(lap, r2, 0)[:68]
        if kind == "faint-pair":
            for c2 in (ch - 1, ch + 1):
                if lo <= c2 <= hi:
                    nb = synth.packet_bits(int(r2.choice(laps)), r2, 2745)
                    put(iq, synth.gfsk_baseband(nb, sps) * 10 ** ((float(r2.uniform(22, 40)) - TOP) / 20), start - int(r2.integers(300, 1500)) * sps, fs, fc, c2,
                        float(r2.uniform(-60e3, 60e3)), float(r2.uniform(0, 2 * np.pi)))
        put(iq, synth.gfsk_baseband(bits, sps) * amp, start, fs, fc, ch, cfo, float(r2.uniform(0, 2 * np.pi)))
        truth.append(dict(slot=start // slot, channel=ch, lap=lap))
        meta.append(dict(kind=kind, level=round(level, 2), start=start, channel=ch, lap=lap))
    o = po.Oracle(fs, fc, sq, po.MODE_SNIFFER, le=False); want, _ = o.run_stream(iq, threads=1)
    x = np.ascontiguousarray(np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])).view(np.float32)
    rec = np.zeros((8192, 8), np.int64); snr = np.zeros(8192)
    n = L.emu_front_m_run(fs, fc, po.MODE_SNIFFER, 0, sq, x.ctypes.data_as(F), len(x) // 2, nsl, rec.ctypes.data_as(Q), snr.ctypes.data_as(D), 8192)
    wi = np.array([[h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors, h.nsym] for h in want], np.int64).reshape(-1, 7)
    # planted = a record whose (channel, LAP) is a packet of this capture in the slot it must be reported in (+-1), of the LAP of THAT packet
    d = paritylib.differential(rec[:n, :7], wi, truth, lag=6)
    tot["cases"] += 1; tot["packets"] += len(meta); tot["planted"] += d["planted_ref"]; tot["only_product"] += d["planted_only_gpu"]; tot["only_oracle"] += d["planted_only_ref"]
    tot["offset_differs"] += d["planted_offset_differs"]; tot["nsym_dev_max"] = max(tot["nsym_dev_max"], d["planted_nsym_max_abs_dev"])
    tot["other_product"] += d["other_gpu"]; tot["other_oracle"] += d["other_ref"]; tot["other_only_product"] += d["other_only_gpu"]; tot["other_only_oracle"] += d["other_only_ref"]
    tot["planted_%dM" % r] += d["planted_ref"]
    for rr in wi[paritylib.classify(wi, truth, 6)]:
        c = [m for m in meta if m["channel"] == rr[1] and m["lap"] == rr[4] and abs(m["start"] // slot - (rr[0] - 6)) <= 1]
        if c:
            tot["planted_" + c[0]["kind"]] += 1; tot["planted_band_%02d" % (int(c[0]["level"]) // 2 * 2)] += 1
    if d["planted_only_gpu"] or d["planted_only_ref"] or only is not None:
        gs = collections.Counter(map(tuple, rec[:n, :6][paritylib.classify(rec[:n, :7], truth, 6)].tolist()))
        ws = collections.Counter(map(tuple, wi[:, :6][paritylib.classify(wi, truth, 6)].tolist()))
        print("case %d fs %.0fM sq %.0f planted %d only product/oracle %d/%d\n   only product: %s\n   only oracle : %s" %
              (case, fs / 1e6, sq, d["planted_ref"], d["planted_only_gpu"], d["planted_only_ref"], sorted((gs - ws).elements()), sorted((ws - gs).elements())), flush=True)
        for side in ((gs - ws), (ws - gs)):
            for rr in side.elements():
                for m in meta:
                    if m["channel"] == rr[1] and m["lap"] == rr[4] and abs(m["start"] // slot - (rr[0] - 6)) <= 1:
                        tot["onesided_" + m["kind"]] += 1
                        print("      packet", m, flush=True)
print("TOTAL " + json.dumps(dict(tot)))
