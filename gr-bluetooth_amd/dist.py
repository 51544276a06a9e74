"""Time-partitioned multi-GPU layer.

The IQ stream is split into contiguous slot ranges, one per rank, each with a left halo of
history()-1 samples (exactly what the reference's GNU Radio history provides at a window
start: lib/multi_block.cc:102-119).  No collective touches the data path; the only
exchange is the gather of the fixed-size hit records (RCCL over xGMI on the GPU box via
backend "nccl", gloo in the CPU tests): HitGatherer posts ONE fixed-size all_gather per batch
(count and records in one buffer, no host round trip for sizes), asynchronously, and the
records of batch n are unpacked while batch n+1 computes.
"""
import numpy as np

HIT_INT_FIELDS = ("slot", "channel", "kind", "offset", "lap", "ac_errors", "nsym")


def partition_slots(total_slots, world_size, rank):
    """Contiguous, balanced slot ranges: returns (first_slot, n_slots) of `rank`."""
    base, rem = divmod(int(total_slots), int(world_size))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def segment_bounds(first_slot, n_slots, history, samples_per_slot, left_margin=0):
    """Absolute sample range [start, start+n) a rank needs for its slots: the history()-1 halo
    plus `left_margin` (btgpu_design.left_margin: what the staged squelch reads in front of a
    segment; pass it to btgpu_process_device as left_margin).  start may be negative (the zeros
    GNU Radio pre-fills before the stream)."""
    if n_slots <= 0:
        return first_slot * samples_per_slot, 0
    start = first_slot * samples_per_slot - (history - 1) - int(left_margin)
    return start, int(left_margin) + history + (n_slots - 1) * samples_per_slot


def hits_to_arrays(hits):
    """List of hit records (btgpu Hit or oracle Hit) -> (int64 [n,7], float64 [n])."""
    n = len(hits)
    ints = np.zeros((n, len(HIT_INT_FIELDS)), np.int64)
    snr = np.zeros(n, np.float64)
    for i, h in enumerate(hits):
        for j, f in enumerate(HIT_INT_FIELDS):
            ints[i, j] = getattr(h, f)
        snr[i] = getattr(h, "snr_db", getattr(h, "snr", 0.0))
    return ints, snr


def struct_to_arrays(rec):
    """numpy structured hit array (multi_*.poll_arrays) -> (int64 [n,7], float64 [n])."""
    ints = np.stack([rec[f].astype(np.int64) for f in HIT_INT_FIELDS], axis=1) if len(rec) else \
        np.zeros((0, len(HIT_INT_FIELDS)), np.int64)
    return ints, rec["snr_db"].astype(np.float64)


def sort_hits(ints, snr):
    """Order the reference's loops print in: slot, channel, kind, offset."""
    if len(ints) == 0:
        return ints, snr
    order = np.lexsort((ints[:, 3], ints[:, 2], ints[:, 1], ints[:, 0]))
    return ints[order], snr[order]


class HitGatherer:
    """One collective per round (a round = one batch, or every few batches: the caller's cadence -- all ranks must use the
    same one): every rank contributes a fixed-size int64 block
    [1 + cap, 8] -- row 0 = (number of records this round, number still to come), rows 1.. =
    the seven integer fields and the bits of snr_db -- to one asynchronous all_gather into ONE
    stacked receive buffer [world, 1 + cap, 8].  Nothing about sizes crosses the host beforehand (no
    counts exchange, no .item()); post() returns at once and collect() of the PREVIOUS post is
    called while the next batch computes.  More than `cap` records in a batch (rare) spill into
    extra rounds that every rank agrees on from the headers it received.

    With a device (backend nccl = RCCL) the whole exchange lives on a stream of its own: the block is
    packed into a pinned host buffer, copied to the device, gathered, and the stacked result comes
    back with a single device-to-host copy -- the compute streams of the handle are never touched.
    force=True runs the collective even in a one-rank group (the single-GPU first-contact test of
    the RCCL path)."""

    def __init__(self, cap=8192, device="cpu", group=None, force=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.cap, self.device = int(cap), device
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.on = inited and (self.world > 1 or force)
        self.pending = None                 # (work handle, stacked receive buffer)
        self.backlog_i = np.zeros((0, len(HIT_INT_FIELDS)), np.int64)
        self.backlog_s = np.zeros((0,), np.float64)
        self.rounds = 0
        self.on_device = str(device) != "cpu"
        self.stream = None
        self.done = None
        if self.on and self.on_device:
            self.stream = torch.cuda.Stream(device=device)
            self.h_send = torch.zeros((1 + self.cap, 8), dtype=torch.int64).pin_memory()
            self.h_recv = torch.zeros((self.world * (1 + self.cap), 8), dtype=torch.int64).pin_memory()
            self.d_send = torch.zeros((1 + self.cap, 8), dtype=torch.int64, device=device)
            self.d_recv = torch.zeros((self.world * (1 + self.cap), 8), dtype=torch.int64, device=device)

    def _pack(self):
        n = min(len(self.backlog_i), self.cap)
        blockh = np.zeros((1 + self.cap, 8), np.int64)
        blockh[0, 0] = n
        blockh[0, 1] = len(self.backlog_i) - n
        if n:
            blockh[1:1 + n, :7] = self.backlog_i[:n]
            blockh[1:1 + n, 7] = self.backlog_s[:n].view(np.int64)
        self.backlog_i, self.backlog_s = self.backlog_i[n:], self.backlog_s[n:]
        return blockh

    def hold(self, ints, snr):
        """Queue this rank's new records for the next round (no collective)."""
        if len(ints):
            self.backlog_i = np.concatenate([self.backlog_i, np.ascontiguousarray(ints, dtype=np.int64)], axis=0)
            self.backlog_s = np.concatenate([self.backlog_s, np.ascontiguousarray(snr, dtype=np.float64)], axis=0)

    def post(self, ints, snr):
        """Queue this rank's new records and start the gather of one round."""
        torch = self.torch
        self.hold(ints, snr)
        if not self.on:
            return
        assert self.pending is None, "collect() the previous round first"
        blockh = self._pack()
        if self.on_device:
            # the whole round is enqueued here, on the gatherer's stream: pack -> device, the collective, the stacked blocks
            # back to pinned memory, one event.  collect() then only waits for that event -- a whole batch later, when
            # it has long fired -- instead of enqueueing a copy and waiting for it on the spot.
            if self.done is not None:
                self.done.synchronize()                  # (the previous round is through with the pinned buffers)
            self.h_send.numpy()[...] = blockh
            with torch.cuda.stream(self.stream):
                self.d_send.copy_(self.h_send, non_blocking=True)
                work = self.dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group, async_op=True)
                work.wait()                              # orders this stream behind the collective (no host wait)
                self.h_recv.copy_(self.d_recv, non_blocking=True)    # ONE copy of the stacked blocks
                self.done = torch.cuda.Event()
                self.done.record(self.stream)
            self.pending = (work, self.d_recv)
        else:
            send = torch.from_numpy(blockh)
            recv = torch.empty((self.world * (1 + self.cap), 8), dtype=torch.int64)      # the blocks of all ranks, concatenated
            work = self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
            self.pending = (work, recv, send)
        self.rounds += 1

    def collect(self, drain=False):
        """Records of the posted round from every rank, sorted (slot, channel, kind, offset).  With
        drain=True keeps going (synchronously) until no rank has records left."""
        torch = self.torch
        if not self.on:
            i, s = self.backlog_i, self.backlog_s
            self.backlog_i, self.backlog_s = i[:0], s[:0]
            return sort_hits(i, s)
        out_i, out_s = [], []
        while True:
            if self.pending is None:
                self.post(self.backlog_i[:0], self.backlog_s[:0])
            work, recv = self.pending[0], self.pending[1]
            if self.on_device:
                self.done.synchronize()
                blocks = self.h_recv.numpy().reshape(self.world, 1 + self.cap, 8)
            else:
                work.wait()
                blocks = recv.numpy().reshape(self.world, 1 + self.cap, 8)
            self.pending = None
            more = False
            for r in range(self.world):
                blk = blocks[r]
                n = int(blk[0, 0])
                more = more or blk[0, 1] > 0
                if n:
                    out_i.append(blk[1:1 + n, :7].copy())
                    out_s.append(blk[1:1 + n, 7].copy().view(np.float64))
            if not (drain and more):
                break
        if not out_i:
            return np.zeros((0, len(HIT_INT_FIELDS)), np.int64), np.zeros((0,), np.float64)
        return sort_hits(np.concatenate(out_i, axis=0), np.concatenate(out_s, axis=0))
