"""Oracle, classic header path: pinned to the answer the compiled reference
gave on samples/channel37.dem: basic_rate_piconet::UAP_from_header resolves
UAP = 0xaf, CLK1-6 offset 38, after 3 packets."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _channel37():
    n = json.load(open(os.path.join(G, "channel37_hits.json")))["n_symbols"]
    return np.unpackbits(np.load(os.path.join(G, "channel37.bits.npy")))[:n]


def test_uap_from_header_known_answer_channel37(po):
    bits = _channel37()
    pn = po.Piconet(0x24D952)
    seen, lines = 0, []
    for off, lap, errs in po.scan_symbols(bits):
        s = bits[off:off + 3125]
        if lap != 0x24D952 or not po.header_present(s, len(s)):
            continue
        seen += 1
        done, log = pn.uap_from_header(s, off // 625, 37)      # clkn = slot index of a 1 Msym/s stream
        lines.append(log)
        if done:
            break
    assert seen == 3 and pn.st.have_uap and pn.st.have_clk6
    assert pn.st.uap == 0xAF and pn.st.clk_offset == 38
    assert lines[0] == "reduced from 64 to 52 CLK1-6 candidates\n"
    assert lines[2].endswith("We have a winner! UAP = 0xaf found after 3 total packets.\n")


def test_try_clock_is_consistent_with_the_resolved_clock(po):
    """Every 24d952 packet of the capture unwhitens to UAP 0xaf at a clock close to
    (clkn + 38) % 64 (the symbol clock of the capture drifts by 4 slots over its 6300), and only two
    of the 64 clocks give that UAP (the HEC reversal is a bijection per clock)."""
    bits = _channel37()
    n = 0
    for off, lap, errs in po.scan_symbols(bits):
        s = bits[off:off + 3125]
        if lap != 0x24D952 or not po.header_present(s, len(s)):
            continue
        good = [c for c in range(64) if po.try_clock(s, c)[0] == 0xAF]
        assert len(good) == 2
        assert any((c - off // 625 - 38) % 64 <= 4 for c in good)
        n += 1
    assert n >= 30


def test_unfec23_never_corrects_data_bits(po):
    """Quirk Q12: two or more parity mismatches always fail; zero or one pass with the data untouched."""
    import ctypes
    rng = np.random.default_rng(3)
    L = po.lib()

    def parity(d):
        g = [1, 1, 0, 1, 0, 1]
        reg = [0] * 5
        for i in range(9, -1, -1):
            fb = d[i] ^ reg[4]
            reg = [fb & g[0]] + [reg[j - 1] ^ (fb if g[j] else 0) for j in range(1, 5)]
        return reg

    for _ in range(200):
        d = [int(x) for x in rng.integers(0, 2, 10)]
        cw = np.array(d + parity(d), np.uint8)
        out = ctypes.create_string_buffer(16)
        assert L.bto_unfec23(cw.tobytes(), 10, out) == 1 and list(out.raw[:10]) == d
        e1 = cw.copy(); e1[10 + rng.integers(0, 5)] ^= 1                    # one parity bit wrong: accepted
        assert L.bto_unfec23(e1.tobytes(), 10, out) == 1 and list(out.raw[:10]) == d
        e2 = cw.copy(); e2[rng.integers(0, 10)] ^= 1                         # one data bit wrong: >= 2 parity mismatches
        assert L.bto_unfec23(e2.tobytes(), 10, out) == 0


def test_sniffer_handlers_on_channel37(po):
    """multi_sniffer_impl::ac/discover/recall/decode over the capture's hits (clkn = symbol offset / 625):
    the ID line for a header-less hit, the discovery dialogue, the queued packets decoded after the
    winner, and the re-discovery after the capture's clock drift breaks the HEC."""
    bits = _channel37()
    sn = po.Sniffer()
    text = "".join(sn.ac(bits[off:off + 3125], off // 625, 37, 20.0) for off, lap, errs in po.scan_symbols(bits)[:9])
    lines = text.split("\n")
    assert lines[0] == "time    105, snr=20.0, channel 37, LAP f2f57b ID"
    assert lines[1] == "time    330, snr=20.0, channel 37, LAP 24d952 working on UAP/CLK1-6"
    assert "We have a winner! UAP = 0xaf found after 3 total packets." in lines
    i = lines.index("Decoding queued packets")
    assert lines[i + 1] == "time    330, channel 37, LAP 24d952 HV3/EV3/3-EV3"
    assert lines[i + 4] == "Finished decoding queued packets"
    assert "time    998, snr=20.0, channel 37, LAP 24d952 POLL" in lines
    assert any(l.endswith("bad HEC! fd af 11 failed to decode header") for l in lines) and "lost clock!" in lines
