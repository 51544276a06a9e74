"""N > 1 path on CPU: world_size-2 and world_size-8 gloo runs of the time partition + hit gather
(gr_bluetooth_amd/dist.py).  The per-rank processor here is the oracle (checker standing in
for the GPU, which does not exist in this container); the partition, halo and gather logic is
the product's and must reproduce the single-process hit list."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_and_segment_bounds(pkg):
    import importlib
    d = importlib.import_module("gr_bluetooth_amd.dist")
    for total, world in ((20, 2), (1600, 8), (7, 3), (3, 8)):
        parts = [d.partition_slots(total, world, r) for r in range(world)]
        assert sum(n for _, n in parts) == total
        pos = 0
        for first, n in parts:
            assert first == pos
            pos += n
    start, n = d.segment_bounds(10, 5, 31601, 5000)
    assert (start, n) == (10 * 5000 - 31600, 31601 + 4 * 5000)
    start, n = d.segment_bounds(10, 5, 31601, 5000, left_margin=4096)      # staged squelch: margin in front
    assert (start, n) == (10 * 5000 - 31600 - 4096, 4096 + 31601 + 4 * 5000)
    assert d.segment_bounds(0, 0, 31601, 5000)[1] == 0


def _worker(rank, world, port, tmp, total=18):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch.distributed as dist
    from conftest import load_pkg
    load_pkg()
    bd = importlib.import_module("gr_bluetooth_amd.dist")
    synth = importlib.import_module("gr_bluetooth_amd.synth")
    import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, total, laps=(0x24D952, 0x4831DD), seed=21, snr_db=24, occupancy=0.5)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    H, slot = o.history, o.slot
    first, n = bd.partition_slots(total, world, rank)
    start, cnt = bd.segment_bounds(first, n, H, slot)
    full = np.concatenate([np.zeros(H - 1, np.complex64), iq])       # GNU Radio pre-fill
    seg = full[start + (H - 1): start + (H - 1) + cnt]               # this rank's samples incl. halo
    hits = []
    for k in range(n):                                               # one work() per slot on the segment
        for h in o.work(seg[k * slot: k * slot + H], first + k):
            hits.append(h)
    ints, snr = bd.hits_to_arrays(hits)
    # the bench's path: one fixed-size asynchronous all_gather per batch, here with a tiny capacity so
    # that records spill into further rounds, posted in two batches
    g = bd.HitGatherer(cap=3, device="cpu")
    half = len(ints) // 2
    g.post(ints[:half], snr[:half])
    a_i, a_s = g.collect()
    g.post(ints[half:], snr[half:])
    b_i, b_s = g.collect(drain=True)
    gi, gs = bd.sort_hits(np.concatenate([a_i, b_i], axis=0), np.concatenate([a_s, b_s], axis=0))
    assert g.rounds >= 3                                             # (the busiest rank's records / capacity: every rank runs every round)
    # ... and in one round with room for everything: the same records
    g2 = bd.HitGatherer(cap=4096, device="cpu")
    g2.hold(ints[:half], snr[:half])                                 # (a cadence of several batches: held, then posted together)
    g2.post(ints[half:], snr[half:])
    hi, hs = g2.collect(drain=True)
    assert g2.rounds == 1 and np.array_equal(hi, gi) and np.array_equal(hs, gs)
    # ... and the bench's cadence: two rounds in flight, the oldest collected only when a third is due, thirds of the
    # records per round with a capacity that makes the middle one spill
    g3 = bd.HitGatherer(cap=2, device="cpu")                         # (the same capacity on every rank: it sizes the collective)
    t1, t2 = len(ints) // 3, 2 * len(ints) // 3
    got = []
    for lo, hi_ in ((0, t1), (t1, t2), (t2, len(ints))):
        if g3.full:
            got.append(g3.collect())
        g3.post(ints[lo:hi_], snr[lo:hi_])
        assert len(g3.inflight) <= 2
    assert g3.full                                                   # rounds 2 and 3 still in flight
    got.append(g3.collect(drain=True))
    di, ds = bd.sort_hits(np.concatenate([g_[0] for g_ in got], axis=0), np.concatenate([g_[1] for g_ in got], axis=0))
    assert np.array_equal(di, gi) and np.array_equal(ds, gs)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), gi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_time_partition_matches_single_process(pkg, po, synth, tmp_path):
    import importlib
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    gi = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, 18, laps=(0x24D952, 0x4831DD), seed=21, snr_db=24, occupancy=0.5)
    want, _ = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER).run_stream(iq)
    bd = importlib.import_module("gr_bluetooth_amd.dist")
    wi, _ = bd.sort_hits(*bd.hits_to_arrays(want))
    assert len(want) > 0
    assert np.array_equal(gi, wi)


def test_eight_rank_time_partition_and_gather(pkg, po, synth, tmp_path):
    """The node's real shape: EIGHT ranks, 40 slots -- five per rank, halo and gather as above, spill rounds and
    the two-in-flight cadence with eight senders per collective -- against the single-process oracle."""
    import importlib
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(8, port, str(tmp_path), 40), nprocs=8, join=True)
    gi = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, 40, laps=(0x24D952, 0x4831DD), seed=21, snr_db=24, occupancy=0.5)
    want, _ = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER).run_stream(iq, threads=4)
    bd = importlib.import_module("gr_bluetooth_amd.dist")
    wi, _ = bd.sort_hits(*bd.hits_to_arrays(want))
    assert len(want) > 20
    assert np.array_equal(gi, wi)
