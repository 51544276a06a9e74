"""What the handlers of the drop-in sniffer block are given on the DEFAULT path (polyphase banks + exact stage, BTGPU_FLAG_SYMBOLS):
the sliced symbols of every record's window from the hit on, compared with the oracle's symbol by symbol -- the input of every
header / payload decode and CRC (lib/multi_sniffer_impl.cc:118-123, lib/packet_impl.cc:1066-1160).  Emulator, adversarial captures.
    python scripts/emu_symbol_parity.py CASES SEED [FIRST STRIDE] [--rates 8,20] [--exact-payload] [--wide]
Per record: symbols compared, first differing symbol (relative to the hit), number differing within the packet's own length."""
import argparse, collections, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import pyoracle as po
import adversarial
from tests.conftest import load_pkg
load_pkg()
ap = argparse.ArgumentParser(); ap.add_argument("cases", type=int); ap.add_argument("seed", type=int); ap.add_argument("--rates", default="8,20")
ap.add_argument("first", type=int, nargs="?", default=0); ap.add_argument("stride", type=int, nargs="?", default=1)
ap.add_argument("--exact-payload", action="store_true", help="BTGPU_FLAG_EXACT_PAYLOAD: long tasks to the end of each burst")
ap.add_argument("--wide", action="store_true", help="the generator's stretched ranges and non-packet interferers (tests/adversarial.py)")
a = ap.parse_args()
L = ctypes.CDLL(os.environ.get("EMU_LIB", os.path.join(ROOT, "tests", "emu", "libemu_bank.so")))
F, Q, D = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double)
L.emu_front_m_syms_run.restype = ctypes.c_int
L.emu_front_m_syms_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, F, ctypes.c_longlong, ctypes.c_int, Q, D, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_uint32)]
KW = 120
L.emu_set_exact_payload(1 if a.exact_payload else 0)
L.emu_long_counts.argtypes = [ctypes.POINTER(ctypes.c_uint)]
rng = np.random.default_rng(a.seed)
tot = collections.Counter()
for case in range(a.cases):
    c = adversarial.draw_case(rng, tuple(int(x) for x in a.rates.split(",")))
    if case % a.stride != a.first % a.stride:
        continue
    fs, fc = c["fs"], c["fc"]
    iq, truth, meta = adversarial.make_adversarial_capture(fs, fc, c["n_slots"], c["n_packets"], c["seed"], c["laps"], lag_slots=6.4, wide=a.wide)
    o = po.Oracle(fs, fc, c["squelch"], po.MODE_SNIFFER, le=False)
    want, _ = o.run_stream(iq, threads=1)
    x = np.ascontiguousarray(np.concatenate([np.zeros(o.history - 1, np.complex64), iq.astype(np.complex64)])).view(np.float32)
    cap = 4096
    rec = np.zeros((cap, 8), np.int64); snr = np.zeros(cap); sym = np.zeros((cap, KW), np.uint32)
    n = L.emu_front_m_syms_run(fs, fc, po.MODE_SNIFFER, 0, c["squelch"], x.ctypes.data_as(F), len(x) // 2, c["n_slots"], rec.ctypes.data_as(Q), snr.ctypes.data_as(D), cap,
                               sym.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    lc = (ctypes.c_uint * 4)(); L.emu_long_counts(lc); tot["long_tasks"] += int(lc[0]); tot["long_tiles"] += int(lc[1]); tot["long_turned_away"] += int(lc[2])
    got = {tuple(int(v) for v in rec[i, :6]): i for i in range(n)}
    for h in want:
        key = (h.slot, h.channel, h.kind, h.offset, h.lap, h.ac_errors)
        if h.kind != 0 or key not in got:
            continue
        # the packet this record belongs to: same channel and LAP, its first sample where the record's access code begins
        sps = int(round(fs / 1e6))
        est = h.slot * o.slot - (o.history - 1) + o.first_ch + (o.ntaps_ch - 1) // 2 + h.offset * sps
        cand = [m for m in meta if m["channel"] == h.channel and m["lap"] == h.lap and abs(m["start"] - est) < 12 * sps]
        if not cand:
            continue
        pk = min(cand, key=lambda m: (abs(m["start"] - est), -m["snr_db"]))      # (two of one LAP at one instant: the record is the stronger one's)
        win = o.window(iq, h.slot)
        osym, _ = o.channel_symbols(o.channel_samples(win, h.channel)[0] if isinstance(o.channel_samples(win, h.channel), tuple) else o.channel_samples(win, h.channel))
        bits = np.unpackbits(sym[got[key]].view(np.uint8), bitorder="little")
        m = min(len(osym), len(bits)) - h.offset
        nb = min(pk["nbits"], m)                                   # the packet's own symbols: access code, header, payload
        diff = np.nonzero(osym[h.offset:h.offset + nb] != bits[h.offset:h.offset + nb])[0]
        tot["records"] += 1; tot["symbols"] += nb
        tot["records_with_a_differing_symbol"] += len(diff) > 0; tot["differing_symbols"] += len(diff)
        if len(diff):
            print("case %d rec %s packet bits %d: %d symbols differ, first at %d" % (case, key, nb, len(diff), diff[0]), flush=True)
tot["exact_payload"] = int(a.exact_payload)
print("TOTAL " + json.dumps(dict(tot)))
