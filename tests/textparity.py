"""Text-level parity of the drop-in multi_sniffer on its DEFAULT path (test infrastructure): captures of one piconet's traffic --
ID, POLL / NULL, DM1, DH1, FHS, DH3 and DH5 packets with real headers, whitening and CRCs, plus LE adverts -- and the text the
reference's handlers print for the oracle's records on them (lib/multi_sniffer_impl.cc:169-318, lib/packet_impl.cc:1066-1160),
to be compared with btrx_amd -S run WITHOUT BTGPU_AUTO=direct."""
import importlib

import numpy as np


def make_piconet_capture(fs, fc, n_slots, seed, le_channels=None, snr_db=24.0):
    synth = importlib.import_module("gr_bluetooth_amd.synth")
    rng = np.random.default_rng(seed)
    sps = int(round(fs / 1e6)); slot = 625 * sps
    lo, hi = synth.visible_channels(fs, fc)
    lap, uap = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256))
    clk0 = int(rng.integers(0, 1 << 20))
    iq, _ = synth.make_capture(fs, fc, n_slots, laps=(lap,), seed=int(rng.integers(0, 1 << 30)), snr_db=snr_db, occupancy=0.0)
    truth = []
    busy_until = 0
    for s in range(n_slots - 7):
        if s * slot < busy_until or rng.random() < 0.25:
            continue
        clk = clk0 + s
        kind = str(rng.choice(["id", "poll", "null", "dm1", "dh1", "fhs", "dh3", "dh5"], p=[0.08, 0.12, 0.05, 0.2, 0.2, 0.05, 0.15, 0.15]))
        if kind == "id":
            bits = synth.access_code_bits(lap)[:68]
        elif kind in ("poll", "null"):
            bits = synth.classic_poll_bits(lap, uap, clk, lt_addr=int(rng.integers(1, 8)), ptype=1 if kind == "poll" else 0)
        elif kind == "dm1":
            bits = synth.classic_dm1_bits(lap, uap, clk, bytes(rng.integers(0, 256, int(rng.integers(1, 18)), dtype=np.uint8)), lt_addr=int(rng.integers(1, 8)))
        elif kind == "dh1":
            bits = synth.classic_dh1_bits(lap, uap, clk, bytes(rng.integers(0, 256, int(rng.integers(1, 28)), dtype=np.uint8)), lt_addr=int(rng.integers(1, 8)))
        elif kind == "fhs":
            bits = synth.classic_fhs_bits(lap, uap, clk, int(rng.integers(0, 1 << 24)), int(rng.integers(0, 256)), int(rng.integers(0, 1 << 16)), int(rng.integers(0, 1 << 26)))
        else:
            pt = 11 if kind == "dh3" else 15
            nmax = 183 if pt == 11 else 339
            bits = synth.classic_dh_multislot_bits(lap, uap, clk, bytes(rng.integers(0, 256, int(rng.integers(nmax // 2, nmax + 1)), dtype=np.uint8)), ptype=pt,
                                                   lt_addr=int(rng.integers(1, 8)))
        ch = int(rng.integers(lo, hi + 1))
        start = s * slot + int(rng.integers(5 * sps, 15 * sps))
        synth.add_burst(iq, bits, start, fs, fc, ch, rng, cfo_hz=15e3, amplitude=10 ** (float(rng.uniform(-8, 4)) / 20))
        truth.append(dict(slot=s, channel=ch, lap=lap))
        busy_until = start + len(bits) * sps + 100 * sps
    if le_channels:
        for _ in range(int(rng.integers(2, 6))):
            ch = int(rng.choice(list(le_channels)))
            synth.add_burst(iq, synth.le_advert_bits(le_channels[ch], rng, payload_bytes=int(rng.integers(6, 30)), pdu_type=int(rng.choice([0, 2, 4, 6]))),
                            int(rng.integers(0, (n_slots - 7) * slot)), fs, fc, ch, rng, cfo_hz=15e3)
    return iq, truth, lap, uap


def sniffer_text(po, o, iq, hits):
    """stdout of multi_sniffer for the oracle's record list: the oracle's packet handlers (multi_sniffer_impl::ac and below) fed with
    the symbols each record hands over, LE lines as aa() prints (the same helper as tests/test_host_block_gpu.py)."""
    sn = po.Sniffer(tun=False)
    text = ""
    for h in hits:
        ch_iq, _ = o.channel_samples(o.window(iq, h.slot), h.channel)
        sym, _ = o.channel_symbols(ch_iq)
        if h.kind != 0:
            text += "time %6d, snr=%.1f, " % (h.slot, h.snr) + po.le_print(sym[h.offset:h.offset + max(h.nsym, 0)], 2402e6 + 1e6 * h.channel)
            continue
        text += sn.ac(sym[h.offset:h.offset + min(h.nsym, 3125)], h.slot, h.channel, h.snr)
    return text
