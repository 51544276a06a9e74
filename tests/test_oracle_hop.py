"""Oracle, hop reversal (lib/piconet_impl.cc:96-338).  The reference holds no
hop-sequence vectors (PARITY UNPINNED): what is checked are two independent restatements against
each other (gen_hops nested loops vs single_hop closed form) and the structural properties of the
Bluetooth hop selection kernel."""
import numpy as np


def test_gen_hops_equals_single_hop_and_is_uniform(po):
    hp = po.Hopper((0xAF << 24) | 0x24D952)
    tab = hp.table()
    rng = np.random.default_rng(4)
    idx = np.concatenate([rng.integers(0, 1 << 27, 4000), [0, 1, 2, 63, 64, (1 << 27) - 1]])
    assert all(tab[i] == hp.single_hop(int(i) << 1) for i in idx)       # table index = CLK27..1
    counts = np.bincount(tab[:79 * 64 * 512], minlength=79)
    assert counts.min() > 0 and tab.max() == 78
    assert abs(int(counts.max()) - int(counts.min())) <= 0.02 * counts.mean()   # every channel equally often
    # within a 32-hop segment of even slots the kernel visits 32 different channels (the 5-bit permutation)
    seg = tab[0:64:2]
    assert len(set(seg.tolist())) == 32


def test_afh_sequence_repeats_each_master_slot(po):
    hp = po.Hopper((0x12 << 24) | 0x4831DD, afh=True)
    tab = hp.table()
    assert np.array_equal(tab[0:4096:2], tab[1:4096:2])
    plain = po.Hopper((0x12 << 24) | 0x4831DD).table()
    assert np.array_equal(tab[0:4096:2], plain[0:4096:2]) and not np.array_equal(tab[1:4096:2], plain[1:4096:2])


def test_winnowing_converges_on_the_true_clock(po):
    addr = (0xAF << 24) | 0x24D952
    hp = po.Hopper(addr)
    tab = hp.table()
    clk = 0x2B5C1A7
    n0 = hp.init_candidates(int(tab[clk]), clk & 0x3F)
    assert (1 << 21) / 79 * 0.9 < n0 < (1 << 21) / 79 * 1.1
    n = n0
    for off in (3, 17, 40, 122, 500, 1021):
        n = hp.winnow(off, int(tab[(clk + off) % (1 << 27)]))
        assert n >= 1
    assert n == 1 and int(hp.candidates()[0]) == clk
    # aliased receiver (25 observable channels): more initial candidates, same convergence
    hp2 = po.Hopper(addr)
    al = po.lib().bto_aliased_channel
    m0 = hp2.init_candidates(al(int(tab[clk])), clk & 0x3F, aliased=True)
    assert m0 > 2.5 * n0
    for off in (3, 17, 40, 122, 500, 1021, 2000, 3001):
        m = hp2.winnow(off, al(int(tab[(clk + off) % (1 << 27)])), aliased=True)
    assert m == 1 and int(hp2.candidates()[0]) == clk


def _hopper_text(po, o, iq, hits, lap, n_slots, aliased=False):
    """stdout of multi_hopper for the oracle front end's hit list (first classic hit of each channel
    per slot, ascending channels) through the oracle's hopper block."""
    hb = po.HopperBlock(lap, aliased=aliased)
    by_slot = {}
    for h in hits:
        if h.kind == 0:
            by_slot.setdefault(h.slot, {}).setdefault(h.channel, h)
    text = ""
    for k in range(n_slots):
        lst = []
        for ch in sorted(by_slot.get(k, {})):
            h = by_slot[k][ch]
            ch_iq, _ = o.channel_samples(o.window(iq, h.slot), h.channel)
            sym, _ = o.channel_symbols(ch_iq)
            lst.append((ch, sym[h.offset:h.offset + min(h.nsym, 3125)]))
        if lst:
            text += hb.slot(k, lst, o.low_ch, o.high_ch)
    return text, hb


def test_hopper_block_acquires_the_master_clock(po, synth):
    """A master hopping by the real selection kernel (generator written from the specification,
    independent of the oracle), 8 of 79 channels visible: UAP / CLK1-6 from header consistency,
    CLK1-27 from the hop pattern, then hopalong decodes every visible packet with the true master
    clock.  CLK offset = clk0 - 6: a packet sent in slot s is reported by work() call s + 6."""
    fs, fc = 8e6, 2476.5e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x3A5C7E1, 700
    iq, truth = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=5, dh1_fraction=0.0)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    hits, _ = o.run_stream(iq, threads=16)
    text, hb = _hopper_text(po, o, iq, hits, lap, nsl)
    pn = hb.piconet
    assert pn.have_clk27 and pn.uap == uap and pn.clk_offset == (clk0 - 6) & 0x7FFFFFF
    assert "We have a winner! UAP = 0xaf" in text and "\nCalculating complete hopping sequence.\n" in text
    assert "\nAcquired CLK1-27 offset = 0x%07x\n" % ((clk0 - 6) & 0x7FFFFFF) in text
    tail = text.split("Acquired CLK1-27 offset")[1].splitlines()[1:]
    sent = {clk: ch for k, clk, ch in truth}
    assert len(tail) > 20
    for line in tail:                               # "clock 0x%07x, channel %2d: POLL"
        clk, ch = int(line[6:15], 16), int(line[25:27])
        assert sent[clk] == ch and line.split(": ")[1] in ("POLL", "NULL")


def test_hopper_block_crc_success_quirk(po, synth):
    """Reference behaviour kept: when the very first packet already passes its payload CRC,
    UAP_from_header returns before d_got_first_packet is set (lib/piconet_impl.cc:482-493 vs :498), every
    later packet gets pattern index 0 and the hop reversal starts over with each new channel."""
    fs, fc = 8e6, 2476.5e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x3A5C7E1, 260
    iq, _ = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=5, dh1_fraction=1.0)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    hits, _ = o.run_stream(iq, threads=16)
    text, hb = _hopper_text(po, o, iq, hits, lap, nsl)
    assert "Correct CRC! UAP = 0xaf found after 1 total packets." in text
    assert "no candidates remaining! starting over . . ." in text and not hb.piconet.have_clk27


def test_hopper_block_with_an_aliasing_receiver(po, synth):
    """multi_hopper's aliased mode (apps/btrx -a, lib/piconet_impl.cc:520-523): a 25 Msps capture folds
    all 79 channels into channels 26..50, every packet of the piconet is seen, the hop reversal runs on
    aliased channel numbers, and hopalong reports each followed packet on the channel it was observed
    on -- which is aliased_channel() of the channel it was sent on."""
    fs, fc = 25e6, 2440e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x1B3C5D2, 200
    iq, truth = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=8, dh1_fraction=0.0, aliased=True)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    assert (o.low_ch, o.high_ch) == (26, 50)
    hits, _ = o.run_stream(iq, threads=16)
    text, hb = _hopper_text(po, o, iq, hits, lap, nsl, aliased=True)
    pn = hb.piconet
    assert pn.have_clk27 and pn.uap == uap and pn.clk_offset == (clk0 - 6) & 0x7FFFFFF, text[-600:]
    tail = text.split("Acquired CLK1-27 offset")[1].splitlines()[1:]
    sent = {clk: ch for k, clk, ch in truth}
    followed = 0
    for line in tail:
        if not line.startswith("clock 0x"):
            continue
        clk, ch = int(line[6:15], 16), int(line[25:27])
        assert ch == ((sent[clk] + 24) % 25) + 26
        followed += 1
    assert followed > 10
