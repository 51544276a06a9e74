"""Oracle (integer half) against the reference's known answers: SURVEY.md F3, section 8(c), A.4/A.4b.
These are what pin oracle/bt_oracle.c to lib/packet_impl.cc (acgen :309-364, check_ac :471-510,
sniff_ac :247-268, sniff_aa :1452-1527)."""
import collections
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_acgen_known_answers(po):
    vec = json.load(open(os.path.join(G, "ac_vectors.json")))
    assert len(vec) == 5
    for lap, ac in vec.items():
        assert po.acgen(int(lap, 16)).hex() == ac


def test_ac_air_order_layout_24d952(po):
    # preamble . 34 parity . LAP LSB-first . Barker . trailer
    want = ("0101" "0101111101101111011110110111110101" "010010101001101100100100" "001101" "0101")
    got = "".join(str(b) for b in po.ac_bits(0x24D952))
    assert got == want


def test_three_independent_acgens_agree(po, pkg, synth):
    rng = np.random.default_rng(5)
    for lap in [0, 0xFFFFFF, 0x9E8B33, 0x9E8B00] + [int(x) for x in rng.integers(0, 1 << 24, 300)]:
        a = po.acgen(lap)
        assert pkg.acgen(lap) == a                                   # product host code (design.cc)
        assert np.packbits(synth.access_code_bits(lap)).tobytes() == a   # generator's encoder


def test_access_code_is_affine_in_lap_and_tables_match(po, pkg):
    rng = np.random.default_rng(11)
    for lap in [0, 1, 0x800000, 0xFFFFFF] + [int(x) for x in rng.integers(0, 1 << 24, 200)]:
        _, _, lo, hi = pkg.debug_tables(8e6, 2476.5e6, lap)
        bits = po.ac_bits(lap)[:68]
        want_lo = sum(int(b) << i for i, b in enumerate(bits[:64]))
        want_hi = sum(int(b) << i for i, b in enumerate(bits[64:68]))
        assert (lo, hi) == (want_lo, want_hi)


def test_channel37_dem_hit_list(po):
    dem = np.unpackbits(np.load(os.path.join(G, "channel37.bits.npy")))
    gold = json.load(open(os.path.join(G, "channel37_hits.json")))
    dem = dem[:gold["n_symbols"]]
    hits = po.scan_symbols(dem)
    assert len(hits) == 33
    laps = collections.Counter("%06x" % h[1] for h in hits)
    assert laps == {"24d952": 31, "133bec": 1, "f2f57b": 1}
    assert [h[0] for h in hits[:4]] == [66136, 206587, 225314, 506941]
    assert hits[0][1] == 0xF2F57B
    assert [[h[0], "%06x" % h[1], h[2]] for h in hits] == gold["hits"]


def _embed(po, lap, offset, n=400):
    s = np.zeros(n, np.uint8)
    s[offset:offset + 72] = po.ac_bits(lap)
    return s


def test_embedded_ac_error_tolerance(po):
    # accept up to 6 flipped parity bits, reject 7 and 8
    lap = 0x24D952
    pos = [6, 9, 12, 15, 18, 21, 24, 27]
    for k in range(9):
        s = _embed(po, lap, 100)
        for p in pos[:k]:
            s[100 + p] ^= 1
        got = po.sniff_ac(s, 300)
        assert got == (100 if k <= 6 else -1), k
    # Barker gate: 0..2 flips pass, 3 fail (gate <= 2 is checked first)
    for k in range(4):
        s = _embed(po, lap, 100)
        for p in (62, 63, 64)[:k]:
            s[100 + p] ^= 1
        assert po.sniff_ac(s, 300) == (100 if k <= 2 else -1), k


def test_sniff_ac_limit_is_exclusive(po):
    s = _embed(po, 0x9E8B33, 50)
    assert po.sniff_ac(s, 50) == -1
    assert po.sniff_ac(s, 51) == 50
    assert po.sniff_ac(np.zeros(200, np.uint8), 100) == -1
    assert po.sniff_ac(np.zeros(10, np.uint8), 0) == -1


def _whitening():
    w = [1, 1, 1, 0, 0, 0, 1]
    for n in range(7, 127):
        w.append(w[n - 7] ^ w[n - 3])
    return w


def test_le_channel_index_mapping(po):
    L = po.lib()
    for f, idx in [(2402e6, 37), (2404e6, 0), (2426e6, 38), (2428e6, 11), (2480e6, 39), (2439e6, -1),
                   (2400e6, -1), (2482e6, -1)]:
        assert L.bto_le_freq2index(f) == idx


def test_sniff_aa_known_answer(po):
    # AA D6 BE 89 8E 00 09, LSB first at offset 50, header whitened for index 37
    w = _whitening()
    assert w[:12] == [1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1]
    s = np.zeros(400, np.uint8)
    by = [0xAA, 0xD6, 0xBE, 0x89, 0x8E, 0x00, 0x09]
    bits = [(b >> i) & 1 for b in by for i in range(8)]
    start = 8                              # le_packet::INDICES[37] (derived by the LFSR rule)
    for i in range(16):
        bits[40 + i] ^= w[(start + i) % 127]
    s[50:50 + 56] = bits
    assert po.sniff_aa(s, 300, 2402e6) == 50
    assert po.sniff_aa(s, 300, 2439e6) == -1          # odd MHz: not an LE channel
    s2 = s.copy(); s2[50 + 10] ^= 1; s2[50 + 20] ^= 1  # 2 AA bit errors tolerated on advertising channels
    assert po.sniff_aa(s2, 300, 2402e6) == 50
    s2[50 + 30] ^= 1
    assert po.sniff_aa(s2, 300, 2402e6) == -1


def test_btbb_find_ac_restatement(po):
    """[EXT libbtbb, unpinned] btbb_find_ac as multi_LAP calls it (lib/multi_LAP_impl.cc:55,93:
    btbb_init(1), max_ac_errors 1, LAP_ANY).  Properties of the published algorithm: the offset is
    the sync-word start (preamble start + 4); one error over sync bits 0..57 is corrected and
    counted, a LAP bit included; two are rejected; one Barker-field error passes the gate and is
    not counted; agreement with the in-tree correlator on clean codes."""
    rng = np.random.default_rng(5)
    for lap in (0x24D952, 0x9E8B33, 0x000000, 0xFFFFFF, 0x800000, 0x7FFFFF):
        st = np.zeros(400, np.uint8)
        st[100:172] = po.ac_bits(lap)
        assert po.sniff_ac(st, 300) == 100
        assert po.btbb_find_ac(st, 300) == (104, lap, 0)
        for pos in list(range(104, 104 + 57)):                      # parity bits and LAP bits 0..22
            s1 = st.copy(); s1[pos] ^= 1
            assert po.btbb_find_ac(s1, 300) == (104, lap, 1), pos
        for _ in range(40):                                          # any two errors in bits 0..56
            a, b = rng.choice(57, 2, replace=False)
            s2 = st.copy(); s2[104 + a] ^= 1; s2[104 + b] ^= 1
            off, got, errs = po.btbb_find_ac(s2, 300)
            assert off != 104 or got != lap                         # never the true code at its place
        for pos in range(104 + 57, 104 + 64):                        # 7-bit Barker field (LAP msb + Barker): fixed, not counted
            s3 = st.copy(); s3[pos] ^= 1
            assert po.btbb_find_ac(s3, 300) == (104, lap, 0)
        # the search range is exclusive, like the loop `count < search_length`
        assert po.btbb_find_ac(st, 104)[0] == -1 and po.btbb_find_ac(st, 105)[0] == 104
    # channel37.dem: every clean in-tree hit is a btbb hit 4 symbols later with the same LAP
    bits = np.unpackbits(np.load(os.path.join(G, "channel37.bits.npy")))[: json.load(open(os.path.join(G, "channel37_hits.json")))["n_symbols"]]
    n = 0
    for off, lap, errs in po.scan_symbols(bits):
        if errs == 0:
            assert po.btbb_find_ac(bits[off + 4: off + 4 + 64 + 1], 1) == (0, lap, 0)
            n += 1
    assert n >= 20
