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
