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
    counts exchange, no .item()); post() returns at once.  Up to `depth` rounds are in flight: the caller collects round
    r when it is about to post round r + depth (`full`), a whole cadence period or more after that round was posted.
    More than `cap` records in a round (rare) spill into extra rounds that every rank agrees on from the headers it received.

    With a device (backend nccl = RCCL) the whole exchange lives on a stream of its own: the block is
    packed into a pinned host buffer, copied to the device, gathered, and the stacked result comes
    back with a single device-to-host copy -- the compute streams of the handle are never touched.
    Why depth 2: beside kernels that fill every CU, each of the three device operations of a round (copy in, collective,
    copy out) only gets onto the device at the next kernel boundary of the compute stream -- measured on one MI355X, a round
    posted beside 1.9 ms batches was still ~1 ms from done four batches later, and a collect() at that point stalled
    the host for that millisecond (3-18 % of the cadence period).  One more period of slack and collect() never waits.
    force=True runs the collective even in a one-rank group (the single-GPU first-contact test of
    the RCCL path)."""

    def __init__(self, cap=8192, device="cpu", group=None, force=False, depth=2):
        import collections
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.cap, self.device, self.depth = int(cap), device, max(1, int(depth))
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.on = inited and (self.world > 1 or force)
        self.inflight = collections.deque()     # rounds posted and not yet collected, oldest first
        self.backlog_i = np.zeros((0, len(HIT_INT_FIELDS)), np.int64)
        self.backlog_s = np.zeros((0,), np.float64)
        self.rounds = 0
        self.on_device = str(device) != "cpu"
        self.stream = None
        self.slots = []                         # device path: `depth` buffer sets, used round-robin
        if self.on and self.on_device:
            self.stream = torch.cuda.Stream(device=device)
            for _ in range(self.depth):
                self.slots.append(dict(
                    h_send=torch.zeros((1 + self.cap, 8), dtype=torch.int64).pin_memory(),
                    h_recv=torch.zeros((self.world * (1 + self.cap), 8), dtype=torch.int64).pin_memory(),
                    d_send=torch.zeros((1 + self.cap, 8), dtype=torch.int64, device=device),
                    d_recv=torch.zeros((self.world * (1 + self.cap), 8), dtype=torch.int64, device=device),
                    done=None))

    @property
    def pending(self):
        """The oldest round in flight (None if there is none)."""
        return self.inflight[0] if self.inflight else None

    @property
    def full(self):
        """True when the next post() needs a collect() first."""
        return len(self.inflight) >= self.depth

    def _pack(self, out=None):
        """This rank's block of the next round, written straight into `out` (the pinned send buffer) when given: only the
        header and the n record rows are touched -- rows behind them are never read by anybody (zeroing and copying the
        whole 1 MB block was half of post()'s 0.2 ms)."""
        n = min(len(self.backlog_i), self.cap)
        blockh = np.zeros((1 + self.cap, 8), np.int64) if out is None else out
        blockh[0, :] = 0
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
        assert not self.full, "collect() the oldest round first"
        if self.on_device:
            # the whole round is enqueued here, on the gatherer's stream: pack -> device, the collective, the stacked blocks
            # back to pinned memory, one event.  collect() then only waits for that event.
            sl = self.slots[self.rounds % self.depth]
            if sl["done"] is not None:
                sl["done"].synchronize()                 # (the round that used this buffer set is through with it)
            self._pack(sl["h_send"].numpy())
            with torch.cuda.stream(self.stream):
                sl["d_send"].copy_(sl["h_send"], non_blocking=True)
                work = self.dist.all_gather_into_tensor(sl["d_recv"], sl["d_send"], group=self.group, async_op=True)
                work.wait()                              # orders this stream behind the collective (no host wait)
                sl["h_recv"].copy_(sl["d_recv"], non_blocking=True)     # ONE copy of the stacked blocks
                sl["done"] = torch.cuda.Event()
                sl["done"].record(self.stream)
            self.inflight.append((work, sl))
        else:
            send = torch.from_numpy(self._pack())
            recv = torch.empty((self.world * (1 + self.cap), 8), dtype=torch.int64)      # the blocks of all ranks, concatenated
            try:
                work = self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
            except (RuntimeError, NotImplementedError):
                # (ProcessGroupGloo only has _allgather_base in recent PyTorch releases: the list form works everywhere)
                work = self.dist.all_gather(list(recv.view(self.world, 1 + self.cap, 8).unbind(0)), send, group=self.group, async_op=True)
            self.inflight.append((work, recv, send))
        self.rounds += 1

    def collect(self, drain=False, sort=True):
        """Records of the OLDEST posted round from every rank, sorted (slot, channel, kind, offset) unless sort=False
        (rank order then: a caller that merges many rounds sorts once at the end -- np.lexsort of the 12 000 records of
        four C79 batches is 0.8 ms of host time per round).  With drain=True: every round in flight, then
        (synchronously) further rounds until no rank has records left."""
        if not self.on:
            i, s = self.backlog_i, self.backlog_s
            self.backlog_i, self.backlog_s = i[:0], s[:0]
            return sort_hits(i, s) if sort else (i, s)
        out_i, out_s = [], []
        while True:
            if not self.inflight:
                self.post(self.backlog_i[:0], self.backlog_s[:0])
            rnd = self.inflight.popleft()
            if self.on_device:
                rnd[1]["done"].synchronize()
                blocks = rnd[1]["h_recv"].numpy().reshape(self.world, 1 + self.cap, 8)
            else:
                rnd[0].wait()
                blocks = rnd[1].numpy().reshape(self.world, 1 + self.cap, 8)
            # (`more` is re-derived from every round's headers: a later round's "still to come" counts supersede an earlier
            # round's -- each rank reports what is left AFTER that round -- so the last round collected is the authoritative one)
            more = False
            for r in range(self.world):
                blk = blocks[r]
                n = int(blk[0, 0])
                more = more or blk[0, 1] > 0
                if n:
                    out_i.append(blk[1:1 + n, :7].copy())
                    out_s.append(blk[1:1 + n, 7].copy().view(np.float64))
            if not drain or not (self.inflight or more):
                break
        if not out_i:
            return np.zeros((0, len(HIT_INT_FIELDS)), np.int64), np.zeros((0,), np.float64)
        oi, os_ = np.concatenate(out_i, axis=0), np.concatenate(out_s, axis=0)
        return sort_hits(oi, os_) if sort else (oi, os_)
