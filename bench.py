#!/usr/bin/env python3
"""bench.py -- complex-IQ Msamples/s of the multi_sniffer hot path on MI355X.

One "step" = one pass of the hot path (channel bank -> squelch -> demod -> M&M -> slicer ->
access-code search -> hit records on the host) over one batch of synthetic wideband IQ that
is already resident in HBM.  Workload = BASELINE.json configs[2]: multi_sniffer, 79 channels,
100 Msps, centre 2441 MHz (C79), synthetic capture per SURVEY.md section 8(d) (8 piconets,
30 % slot occupancy, payload 0-2745 bits, CFO +-75 kHz).

N > 1 (`--gpus N`): the stream is time-partitioned, rank r owns slots [r S, (r+1) S) plus a left
halo of history()-1 (+ left_margin) samples, and the hit records travel to every rank with ONE
asynchronous fixed-size all_gather_into_tensor per round -- every --gather-every batches and at the
flush (RCCL over xGMI; gr-bluetooth_amd/dist.py HitGatherer, on a stream of its own) -- weak
scaling, per-rank slots fixed.  Started under
torchrun (WORLD_SIZE / RANK / LOCAL_RANK in the environment) each process is one rank; started
plainly with --gpus N > 1 this script spawns its own N ranks, one device each, and FAILS if it
cannot see N devices.

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` is measured in the configuration of the
drop-in block, gr::bluetooth::multi_sniffer::work (lib/multi_sniffer_impl.cc:107-149): classic search, LE pass,
the symbols of every hit handed over with the GPU header sweep -- BTGPU_FLAG_LE | HEADERS (what host/blocks.cc
sets).  Added objects: `roofline` (the step's dominant kernel: since round 6 exact_rows_kernel on the fp32 matrix
pipe), `roofline_bank` (the channel bank against the HBM roofline, rounds 1-5's `roofline`), `cpu_baseline`,
`parity` (a differential of the records of one step against the CPU oracle run on all host cores over the first
--parity-slots slots of the same capture, tests/paritylib.py), and at N = 1 `classic_only` (the same K steps
without LE pass and symbols: rounds 1-5's headline), `c8` (BASELINE configs[1], 8 channels at 8 Msps, the
headline's flags), `verify` (exact rows per step, the A/B without them), `host_fed`.
"""
import argparse
import collections
import hashlib
import gc
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3

WORKLOADS = {
    "c79": dict(sample_rate=100e6, center_freq=2441e6, name="multi_sniffer 79-channel classic BT, 100 Msps synthetic wideband IQ"),
    "c8": dict(sample_rate=8e6, center_freq=2476.5e6, name="multi_sniffer 8-channel (8 MHz span) synthetic IQ"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c79", choices=sorted(WORKLOADS))
    ap.add_argument("--slots", type=int, default=0, help="slots per rank per step (0 = workload default)")
    ap.add_argument("--squelch", type=float, default=10.0, help="SNR squelch threshold in dB (btrx -t default 10.0)")
    ap.add_argument("--snr", type=float, default=25.0, help="burst SNR in 1 MHz (dB) of the synthetic capture")
    ap.add_argument("--piconets", type=int, default=8)
    ap.add_argument("--occupancy", type=float, default=0.3)
    ap.add_argument("--cfo-hz", type=float, default=75e3, help="carrier offset range of the bursts (SURVEY 8(d): +-75 kHz)")
    ap.add_argument("--max-payload-bits", type=int, default=2745, help="payload length range (SURVEY 8(d): 0-2745)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="approximate budget of the single-thread cpu_baseline sample")
    ap.add_argument("--parity-slots", type=int, default=1600,
                    help="slots of the capture the all-core oracle differential covers (N = 1; 0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and parity legs")
    ap.add_argument("--no-ab", action="store_true", help="skip the A/B region without the exact stage (BTGPU_FLAG_NO_VERIFY)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (btgpu_process_host, PCIe-inclusive) leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--all-on-device0", action="store_true", help="dry run of the N > 1 path on a 1-GPU box (use with --backend gloo)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--headers", action="store_true", help="(default since round 6) export hit symbols and run the GPU header sweep (BTGPU_FLAG_HEADERS), as the C++ multi_sniffer block does")
    ap.add_argument("--classic-only", action="store_true", help="the headline region WITHOUT the LE pass and the symbol hand-over (rounds 1-5's headline; now the extra key classic_only)")
    ap.add_argument("--no-exact-all", action="store_true", help="skip the short region with BTGPU_FLAG_EXACT_ALL (no selection: every row of every channel exact)")
    ap.add_argument("--no-c8", action="store_true", help="skip the short run of BASELINE configs[1] (8 channels, 8 Msps) that fills the c8 object")
    ap.add_argument("--synth-device", default=None, help="where the synthetic capture is generated (default: the GPU).  'cpu': torch CPU ops + one "
                    "upload -- for the rocprofv3 --pmc passes of the C8 workload, which crash inside torch's own randn launches")
    ap.add_argument("--exact-payload", action="store_true", help="with --headers: BTGPU_FLAG_EXACT_PAYLOAD, as the C++ multi_sniffer block sets it")
    ap.add_argument("--le", action="store_true", help="also run the le_packet::sniff_aa pass (BTGPU_FLAG_LE), as the C++ multi_sniffer block does")
    ap.add_argument("--no-block-config", action="store_true", help="skip the second timed region (the classic-only configuration)")
    ap.add_argument("--full-timing", action="store_true", help="HIP events around every kernel in the headline region too (kernel_avg_ms of all kernels)")
    ap.add_argument("--no-timing", action="store_true", help="no per-kernel HIP events (BTGPU_FLAG_TIMING off): kernel traces without event records; roofline then has no kernel time")
    ap.add_argument("--gather-every", type=int, default=4, help="N > 1: batches per record-gather round (one fixed-size all_gather every that "
                    "many batches and at the flush; measured on one GPU, scripts/gather_cost.py: a round per 2 ms batch costs 5-30 %% "
                    "of the step depending on how RCCL's kernel lands beside the bank kernel, one per several batches nothing; the block "
                    "holds 4096 records per batch of the cadence, more spill into extra rounds)")
    ap.add_argument("--force-gather", action="store_true", help="N = 1: still push the records through the HitGatherer collective (single-rank process group; first contact of the RCCL path on one GPU)")
    ap.add_argument("--sync", action="store_true", help="harvest every batch before the next one is enqueued (no tail overlap)")
    ap.add_argument("--channelizer", type=int, default=0)
    ap.add_argument("--squelch-mode", type=int, default=0, help="0 auto, 1 direct, 2 staged")
    ap.add_argument("--pmc-json", default="", help="rocprofv3 PMC summary (scripts/pmc_hbm_json.py) of THIS build: its HBM bytes "
                    "become roofline.traffic; refused when its build id differs from libbtgpu.so's")
    return ap.parse_args(argv)


def build_id():
    """sha256 (16 hex) of the loaded libbtgpu.so: ties PMC files to the kernels they measured."""
    so = os.path.join(ROOT, "gr-bluetooth_amd", "libbtgpu.so")
    return hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]


def traffic_by_device_code(key, S):
    """roofline.traffic when profiles/ holds no PMC summary of THIS build of libbtgpu.so.  A summary collected on ANOTHER build still
    measures the dominant kernel if that kernel's instructions are the same in both (a change elsewhere in the library does not
    un-measure the bank kernel; a change to it does): the summaries carry a per-kernel id of the device code they ran
    (scripts/device_code_ids.py: sha256 of the normalised gfx950 disassembly, stamped from a bit-identical rebuild of the measured
    build).  Returns the roofline fields on an equal id, else None.  key: the kernel-name prefix, S: slots per launch."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import device_code_ids as dci
    ids = dci.kernel_code_ids(os.path.join(ROOT, "gr-bluetooth_amd", "libbtgpu.so"))
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm.json")), reverse=True):
        try:
            pj = json.load(open(cand))
        except (OSError, ValueError):
            continue
        if pj.get("slots") != S or not isinstance(pj.get("kernel_code_sha"), dict):
            continue
        match = sorted([k for k in pj["kernels"] if key and k.startswith(key)], key=lambda k: -pj["kernels"][k]["hbm_bytes"])
        want_id = pj["kernel_code_sha"].get(match[0]) if match else None
        if want_id and want_id == dci.lookup(ids, match[0]):
            return {"_summary": pj, "_kernel": match[0],
                    "traffic": pj["kernels"][match[0]]["hbm_bytes"], "traffic_source": os.path.basename(cand),
                    "traffic_basis": "PMC passes ran on build %s; %s is instruction-identical in this build (device code id %s, "
                                     "scripts/device_code_ids.py)" % (pj.get("build_id"), match[0], want_id)}
    return None


def _spawn_entry(rank, args, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run_rank(args)


def main():
    args = parse_args()
    if "WORLD_SIZE" in os.environ:                  # torchrun / the driver's launcher: one rank per process
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        return run_rank(args)
    if args.gpus > 1:                               # plain start: spawn our own ranks, one device each
        import torch
        import torch.multiprocessing as mp
        have = torch.cuda.device_count()
        if not args.all_on_device0 and have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (no silent fallback to fewer ranks; "
                             "--all-on-device0 --backend gloo is the 1-GPU dry run)" % (args.gpus, have))
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_spawn_entry, args=(args, args.gpus, port), nprocs=args.gpus, join=True)
        return
    os.environ.pop("RANK", None)
    return run_rank(args)


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        args.no_cpu = True                          # cpu_baseline / oracle differential: rank 0 at N = 1 only
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.all_on_device0:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: device %d not visible (%d GPUs)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if args.backend == "nccl" else torch.device("cpu")
    if world > 1 or args.force_gather:
        if world == 1:                              # a one-rank group: the collective path end to end on one GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    pkg = load_pkg()
    import importlib
    synth = importlib.import_module("gr_bluetooth_amd.synth")
    bdist = importlib.import_module("gr_bluetooth_amd.dist")

    wl = WORKLOADS[args.workload]
    fs, fc = wl["sample_rate"], wl["center_freq"]
    # C79: 1.44 s of signal per step = 768 window-kernel workgroups of three slots, exactly the number
    # that is resident at once (three per CU).  C8: the same number of input samples per second of
    # signal is 12.5x smaller, so a step takes more slots
    S = args.slots or (2304 if args.workload == "c79" else 16384)
    laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(args.piconets))
    gen = dict(laps=laps, seed=args.seed, snr_db=args.snr, occupancy=args.occupancy, cfo_hz=args.cfo_hz,
               max_payload_bits=args.max_payload_bits)

    base_flags = 0 if args.sync else pkg.FLAG_ASYNC
    # HIP events in the timed region: around the channel-bank kernel only (the roofline's kernel; two records per batch) in
    # the headline region, around every kernel (--full-timing, and always in the classic_only region: eleven records per
    # batch, measured at 2-3 % of the step)
    head_timing = 0 if args.no_timing else (pkg.FLAG_TIMING if args.full_timing else pkg.FLAG_TIMING_BANK)

    def make_block(extra, timing=None):
        return pkg.multi_sniffer(fs, fc, args.squelch, False, device=local_rank, max_batch_slots=S,
                                 channelizer=args.channelizer, squelch=args.squelch_mode,
                                 flags=base_flags | extra | (head_timing if timing is None else timing))
    # The headline is what gr::bluetooth::multi_sniffer::work does (lib/multi_sniffer_impl.cc:107-149): the LE pass after the classic
    # one, the symbols of every hit handed to the packet handlers -- the flags host/blocks.cc creates its handle with.
    if not args.classic_only:
        args.le = args.headers = args.exact_payload = True
    head_flags = (pkg.FLAG_HEADERS if args.headers else 0) | (pkg.FLAG_LE if args.le else 0) | (pkg.FLAG_EXACT_PAYLOAD if args.exact_payload else 0)
    blk = make_block(head_flags)
    des = blk.design
    H, slot = des.history, des.samples_per_slot
    nch = des.high_channel - des.low_channel + 1

    first, _ = bdist.partition_slots(world * S, world, rank)        # rank r owns slots [r S, (r+1) S)
    margin = des.left_margin
    a0, n_need = bdist.segment_bounds(first, S, H, slot, margin)
    seg, truth = synth.make_segment_torch(fs, fc, first, first + S, args.synth_device or device, left_pad=H - 1 + margin, **gen)
    seg = seg.to(device).contiguous()
    n_complex = seg.shape[0]
    assert n_complex >= n_need and a0 == first * slot - (H - 1) - margin   # (the generator runs to the end of the last slot)
    torch.cuda.synchronize()
    gatherer = bdist.HitGatherer(cap=4096 * max(1, args.gather_every), device=coll_device, force=args.force_gather)
    gathering = world > 1 or args.force_gather

    def step(last=False, gather=True, blk=None, every=False):
        """One pass of the hot path over the rank's batch.  In the (default) pipelined mode the
        records of a batch are harvested while the next batch runs; the last step of a timed
        region flushes, so every record of every step is on the host inside the timed region.
        N > 1: the records this rank has ready are posted to one asynchronous all_gather; what
        comes back here is the post of two rounds ago (HitGatherer keeps two in flight: a round only gets onto a
        device full of compute at kernel boundaries, and is collected when it has long finished)."""
        blk = blk or head_blk
        dump = os.environ.get("BENCH_DUMP_STEPS")
        tt = [time.perf_counter()]
        blk.process_device(seg.data_ptr(), n_complex, first, S, left_margin=margin)
        tt.append(time.perf_counter())
        if last:
            blk.flush()
        tt.append(time.perf_counter())
        ints, snr = bdist.struct_to_arrays(blk.poll_arrays())
        tt.append(time.perf_counter())
        if not gathering or not gather:
            if dump and last:
                print("last step ms: process %.2f flush %.2f poll %.2f" % tuple(1e3 * (b - a) for a, b in zip(tt, tt[1:])), file=sys.stderr)
            return ints, snr
        # one round every --gather-every batches (every rank counts the same batches) and at the flush
        step.count = getattr(step, "count", 0) + 1
        if not last and not every and step.count % max(1, args.gather_every):
            gatherer.hold(ints, snr)
            return ints[:0], snr[:0]
        got = gatherer.collect(sort=False) if gatherer.full else (ints[:0], snr[:0])      # the round posted two cadence periods ago
        tt.append(time.perf_counter())
        gatherer.post(ints, snr)
        tt.append(time.perf_counter())
        if last:
            more = gatherer.collect(drain=True, sort=False)
            got = (np.concatenate([got[0], more[0]], axis=0), np.concatenate([got[1], more[1]], axis=0))
        tt.append(time.perf_counter())
        if dump:
            print("gather step ms (last %d): process %.2f flush %.2f poll %.2f collect %.2f post %.2f drain %.2f" % ((int(last),) + tuple(1e3 * (b - a) for a, b in zip(tt, tt[1:]))), file=sys.stderr)
        return got

    head_blk = blk

    def tdiff(a, b):
        return np.array(list(b.kernel_ms)) - np.array(list(a.kernel_ms)), \
            np.array(list(b.kernel_launches), dtype=np.float64) - np.array(list(a.kernel_launches), dtype=np.float64)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Bring the device to its steady power state first: an idle MI355X sits at 500 MHz and needs
    # some tens of milliseconds of load before the clocks settle (measured: the first ~5 batches
    # after idle run ~1.5x slower).  Untimed, like the W warm-up steps that follow.
    # No cyclic-GC pass inside the timed region (a generation-2 sweep of the interpreter heap with torch
    # loaded costs tens of milliseconds, i.e. more than the whole region at the default K) -- and none
    # between the warm-up and the timed region either: the device clocks sag during any idle gap and
    # take ~10 batches to come back.
    def timed_region(b, gather=True):
        """prewarm + W warm-up steps + K timed steps on block `b`; returns (elapsed s, ints, snr, marks, fence ms, kernel ms,
        kernel launches)."""
        gc.collect()
        gc.disable()
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:      # per-rank clock: no collectives in here
            step(last=False, gather=False, blk=b)
        step(last=True, gather=False, blk=b)
        fence()
        # (warm-up with a gather round per step: the first collective posted while the device is full of queued batches has
        # been seen to block the host for 5-8 ms, one-off -- it belongs to the warm-up, not to any steady state)
        for i in range(args.warmup):
            step(last=(i == args.warmup - 1), gather=gather, blk=b, every=True)
        fence()
        tm0 = b.timing()
        t0 = time.perf_counter()
        got_i, got_s = [], []
        marks = [t0]
        for i in range(args.steps):
            ints, snr = step(last=(i == args.steps - 1), gather=gather, blk=b)
            got_i.append(ints); got_s.append(snr)
            marks.append(time.perf_counter())
        ints = np.concatenate(got_i, axis=0)
        snr = np.concatenate(got_s, axis=0)
        t_loop = time.perf_counter()
        if os.environ.get("BENCH_DUMP_STEPS"):
            print("step_ms", [round(float(v) * 1e3, 2) for v in np.diff(np.array(marks))], file=sys.stderr)
        fence()
        elapsed = time.perf_counter() - t0
        fence_ms = (time.perf_counter() - t_loop) * 1e3
        gc.enable()
        tm1 = b.timing()
        kms, kl = tdiff(tm0, tm1)
        # exact confirmation of the polyphase path's records (always counted by the library): windows re-run per step etc.
        b._long_stats = {"tasks_per_step": round((tm1.long_tasks - tm0.long_tasks) / max(1, args.steps), 1),
                         "rows_per_step": round((tm1.long_rows - tm0.long_rows) / max(1, args.steps), 1),
                         "turned_away": int(tm1.long_turned_away - tm0.long_turned_away)}
        b._verify_stats = {"windows_per_step": round((tm1.verify_windows - tm0.verify_windows) / max(1, args.steps), 1),
                           "rows_per_step": round((tm1.verify_rows - tm0.verify_rows) / max(1, args.steps), 1),
                           "turned_away": int(tm1.verify_turned_away - tm0.verify_turned_away)}
        return elapsed, ints, snr, marks, fence_ms, kms, kl

    elapsed, ints, snr, marks, fence_ms, kernel_ms, kernel_launches = timed_region(blk)
    # the exact stage re-runs every window that can carry a packet's record through the direct-form arithmetic:
    # 4 * ntaps multiply-adds per recomputed demodulated row
    blk_stats, blk_long = dict(blk._verify_stats), dict(blk._long_stats)
    verify_obj = dict(blk._verify_stats)
    verify_obj["second_run"] = dict(blk._long_stats)
    verify_obj["enabled"] = bool(verify_obj["windows_per_step"] > 0 or os.environ.get("BTGPU_VERIFY", "1") != "0")
    verify_obj["gfma_per_step"] = round(verify_obj["rows_per_step"] * 4.0 * ((des.ntaps_channel + 7) // 8 * 8) / 1e9, 3)
    rank_elapsed = [elapsed]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=coll_device)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # every rank's own time: a slow rank is visible in the line
        rank_elapsed = [float(v) for v in t.cpu()]
        elapsed = max(rank_elapsed)
    def one_copy(ints, snr):
        # every step processes the same resident batch: keep one copy of the (identical) record set
        if args.steps > 1 and len(ints):
            ints_u, idx = np.unique(ints, axis=0, return_index=True)
            assert len(ints_u) * args.steps == len(ints), "steps produced different record sets"
            return bdist.sort_hits(ints_u, snr[idx])
        return ints, snr
    ints, snr = one_copy(ints, snr)

    # ---- the drop-in block's configuration (N = 1): gr::bluetooth::multi_sniffer always runs the LE pass after the
    # classic one and hands the symbols of every hit to its packet handlers (lib/multi_sniffer_impl.cc:107-149), so
    # host/blocks.cc creates its handle with BTGPU_FLAG_LE | BTGPU_FLAG_HEADERS: the same K steps in that configuration
    # ---- A/B of the exact stage (N = 1): the same K steps with BTGPU_FLAG_NO_VERIFY -- what the confirmation of the records costs
    if world == 1 and not args.no_ab and verify_obj["windows_per_step"] > 0:
        blk.close()
        blk = make_block(head_flags | pkg.FLAG_NO_VERIFY)
        a_el, a_ints, a_snr, _m, _f, _k1, _k2 = timed_region(blk, gather=False)
        a_ints, a_snr = one_copy(a_ints, a_snr)
        verify_obj["ab_no_verify"] = {"flags": "the same + BTGPU_FLAG_NO_VERIFY", "value": round(float(S) * slot * args.steps / a_el / 1e6, 3),
                                      "unit": "Msamples/s", "ms_per_step": round(a_el / args.steps * 1e3, 3), "hits": int(len(a_ints)),
                                      "records_equal_on_6_fields": bool(len(a_ints) == len(ints) and np.array_equal(a_ints[:, :6], ints[:, :6]))}
        verify_obj["cost_frac_of_step"] = round(1.0 - a_el / elapsed, 4)

    # ---- the classic-only configuration (N = 1): rounds 1-5's headline -- no LE pass, no symbol hand-over: the LAP list alone ----
    block_cfg = None
    if world == 1 and not args.no_block_config and not args.classic_only:
        blk.close()
        blk = make_block(0, timing=pkg.FLAG_TIMING)
        b_el, b_ints, b_snr, _m, _f, b_kms, b_kl = timed_region(blk, gather=False)
        b_ints, b_snr = one_copy(b_ints, b_snr)
        head_ac = ints[ints[:, 2] == 0]
        block_cfg = {"flags": "ASYNC|TIMING (no LE pass, no symbols: the classic LAP list; every kernel bracketed, ~3 % slower than the light form)",
                     "value": round(float(S) * slot * args.steps / b_el / 1e6, 3), "unit": "Msamples/s",
                     "ms_per_step": round(b_el / args.steps * 1e3, 3), "hits": int(len(b_ints)),
                     "kernel_avg_ms": {pkg.KERNEL_NAMES[i]: round(float(b_kms[i] / b_kl[i]), 4) if b_kl[i] else 0.0
                                       for i in range(len(pkg.KERNEL_NAMES))},
                     "verify": dict(blk._verify_stats),
                     "ac_records_equal_headline": bool(np.array_equal(b_ints, head_ac)),
                     "ac_records_equal_headline_on_6_fields": bool(len(b_ints) == len(head_ac) and np.array_equal(b_ints[:, :6], head_ac[:, :6]))}

    # ---- BASELINE configs[1] (N = 1): 8 channels at 8 Msps, the same flags, driver-timed ----
    c8 = None
    if world == 1 and not args.no_c8 and args.workload == "c79":
        blk.close()
        w8 = WORKLOADS["c8"]
        S8 = 16384
        b8 = pkg.multi_sniffer(w8["sample_rate"], w8["center_freq"], args.squelch, False, device=local_rank, max_batch_slots=S8,
                               flags=base_flags | head_flags | head_timing)
        d8 = b8.design
        m8 = d8.left_margin
        seg8, truth8 = synth.make_segment_torch(w8["sample_rate"], w8["center_freq"], 0, S8, args.synth_device or device, left_pad=d8.history - 1 + m8, **gen)
        seg8 = seg8.to(device).contiguous()
        torch.cuda.synchronize()
        def step8(last):
            b8.process_device(seg8.data_ptr(), seg8.shape[0], 0, S8, left_margin=m8)
            if last:
                b8.flush()
            return bdist.struct_to_arrays(b8.poll_arrays())
        # the headline's regime (timed_region): no cyclic-GC pass inside the timed loop, prewarm_ms of load and W steps in front of it.
        # (Until the library kept a destroyed handle's streams for the next one -- btgpu.hip g_stream_pool -- this leg read 14.8-15.1 G where
        # the same block measures 19-20 G in a process of its own: streams created after others were destroyed ran every kernel 4-19 %
        # longer, profiles/r06_z_c8_leg_*, r06_zz_second_handle.txt.)
        gc.collect()
        gc.disable()
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:
            step8(False)
        for i in range(max(2, args.warmup)):
            step8(i == max(2, args.warmup) - 1)
        torch.cuda.synchronize()
        tm0 = b8.timing()
        K8 = max(10, 2 * args.steps)
        t0 = time.perf_counter()
        n8 = 0
        for i in range(K8):
            n8 += len(step8(i == K8 - 1)[0])
        torch.cuda.synchronize()
        el8 = time.perf_counter() - t0
        gc.enable()
        tm1 = b8.timing()
        k8, l8 = tdiff(tm0, tm1)
        bank8 = float(k8[0] / l8[0]) if l8[0] else 0.0
        ex8 = float(k8[7] / l8[7]) if l8[7] else 0.0
        c8 = {"workload": w8["name"], "flags": "the headline's", "slots_per_step": S8, "steps": K8,
              "value": round(float(S8) * d8.samples_per_slot * K8 / el8 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(el8 / K8 * 1e3, 3),
              "hits_per_step": n8 // K8,
              "bank_ms": round(bank8, 4), "bank_frac_of_hbm_roofline": round(8.0 * S8 * d8.samples_per_slot / (bank8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if bank8 > 0 else None,
              "exact_ms": round(ex8, 4),
              "exact_rows_per_step": round((tm1.verify_rows - tm0.verify_rows) / K8, 1), "busy_windows_per_step": round((tm1.verify_windows - tm0.verify_windows) / K8, 1)}
        b8.close()
        del seg8
        blk = make_block(head_flags)                       # (something to close at the end)

    # ---- no selection at all (N = 1): BTGPU_FLAG_EXACT_ALL -- every row of every channel through the reference's arithmetic on the
    # matrix pipe; every field of every record then equals the oracle's (nsym and the noise-born records included) ----
    exact_all = None
    x_ints = None
    if world == 1 and not args.no_exact_all and not args.classic_only:
        blk.close()
        blk = make_block(head_flags | pkg.FLAG_EXACT_ALL)
        keep_steps, keep_warm = args.steps, args.warmup
        args.steps, args.warmup = max(4, args.steps // 2), 2
        x_el, x_ints, x_snr, _m, _f, x_kms, x_kl = timed_region(blk, gather=False)
        x_ints, x_snr = one_copy(x_ints, x_snr)
        exact_all = {"flags": "the headline's + BTGPU_FLAG_EXACT_ALL", "steps": args.steps,
                     "value": round(float(S) * slot * args.steps / x_el / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(x_el / args.steps * 1e3, 3),
                     "hits": int(len(x_ints)), "exact_ms": round(float(x_kms[7] / x_kl[7]), 3) if x_kl[7] else None,
                     "rows_per_step": blk._verify_stats["rows_per_step"],
                     "records_equal_headline_on_6_fields_where_planted": None}
        args.steps, args.warmup = keep_steps, keep_warm

    # ---- the host-fed rate (N = 1): what btrx_amd and the GNU Radio block see -- the batch lies in HOST memory and goes through
    # btgpu_process_host (PCIe-inclusive; never `value`).  Source page-locked (a block can register the scheduler's buffer once):
    # copied to the device as it lies; pageable: through the library's pinned staging buffers.  H2D alone = the same bytes with
    # hipMemcpyAsync from page-locked memory, the ceiling of this box's link.
    host_fed = None
    if world == 1 and not args.no_host_fed:
        hb = 8
        seg_host = seg.cpu()
        pinned = seg_host.pin_memory()
        pageable = seg_host.numpy()
        def fed(src):
            b = make_block(0, timing=0)
            for _ in range(2):                                                              # warm-up: both staging / device buffer pairs get allocated
                b.process_host(src, first, S, left_margin=margin)
            b.flush(); b.poll_arrays()
            t0 = time.perf_counter()
            nrec = 0
            for i in range(hb):
                b.process_host(src, first, S, left_margin=margin)
                nrec += len(b.poll_arrays())
            b.flush(); nrec += len(b.poll_arrays())
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            b.close()
            return el, nrec
        el_pin, n_pin = fed((pinned.data_ptr(), n_complex))
        el_page, n_page = fed((pageable.ctypes.data, n_complex))
        dst = torch.empty_like(seg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(hb):
            dst.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        el_copy = time.perf_counter() - t0
        nbytes = float(n_complex) * 8.0
        host_fed = {"entry": "btgpu_process_host", "batches": hb, "slots_per_batch": S,
                    "pinned_source": {"value": round(S * slot * hb / el_pin / 1e6, 1), "unit": "Msamples/s", "h2d_GBps": round(nbytes * hb / el_pin / 1e9, 2)},
                    "pageable_source": {"value": round(S * slot * hb / el_page / 1e6, 1), "unit": "Msamples/s", "GBps": round(nbytes * hb / el_page / 1e9, 2)},
                    "h2d_alone_GBps": round(nbytes * hb / el_copy / 1e9, 2),
                    "frac_of_h2d_alone": round(el_copy / el_pin, 3),
                    "records_per_batch_equal_device_fed": (bool(n_pin == n_page == hb * len(ints)) if not (args.le or args.headers) else None)}
        del dst, pinned, seg_host

    total_samples = float(world) * S * slot * args.steps
    value = total_samples / elapsed / 1e6

    # ---- correctness gate 1: ground-truth bursts reported (recall of the synthetic capture) ----
    got = set((int(r[0]), int(r[1]), int(r[4])) for r in ints)           # (slot, channel, lap)
    lo_slot, hi_slot = 0, world * S
    if world > 1:                                   # a rank only knows its own truth; regenerate all
        truth, _ = synth.burst_schedule(fs, fc, 0, world * S, laps, args.seed, args.occupancy, args.cfo_hz,
                                        args.max_payload_bits)
    expected = found = 0
    for tr in truth:
        det = tr["slot"] + 6                        # sniffer window lag: (history()-1)/slot = 6.3
        if det + 1 >= hi_slot or tr["slot"] < lo_slot:
            continue
        expected += 1
        if any((det + d, tr["channel"], tr["lap"]) in got for d in (-1, 0, 1)):
            found += 1

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events inside libbtgpu, same stream) ----
        names = pkg.KERNEL_NAMES
        NK = len(names)
        avg = [kernel_ms[i] / kernel_launches[i] if kernel_launches[i] else 0.0 for i in range(NK)]
        # the dominant kernel is looked for on the critical path: in pipelined mode the tail
        # (finish_kernel) of batch n runs on its own stream underneath batch n+1
        crit = [i for i in range(NK) if not (names[i] in ("finish", "verify") and not args.sync)]
        dom_all = max(crit, key=lambda i: avg[i])              # (light timing: the channel bank and the exact rows' kernel are bracketed)
        bank_like = [i for i in crit if names[i] != "exact"]
        dom = max(bank_like, key=lambda i: avg[i])             # the HBM-side figure: the dominant STREAMING kernel (the channel bank)
        bytes_per_launch = 8.0 * S * slot                      # 8 B per complex input sample, read once
        ach = bytes_per_launch / (avg[dom] * 1e-3) / 1e9 if avg[dom] > 0 else 0.0
        # algorithmic FMA per input sample of the direct-form banks
        fma_ch = nch * des.ntaps_channel * 2.0 / des.decimation * 2
        fma_noise = nch * des.ntaps_noise * 2.0 / des.decimation * 2
        roof = {"bound": "hbm", "kernel": names[dom], "achieved": round(ach, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": round(avg[dom], 4),
                "kernel_avg_ms": {names[i]: round(avg[i], 4) for i in range(NK)},
                "note": "ddc_channel = channel bank (+ noise stage 1 when fused); ddc_noise = 0 then; without --full-timing only "
                        "ddc_channel and exact are bracketed in the headline region (classic_only.kernel_avg_ms has every kernel)"}
        direct = (names[dom] == "ddc_channel" and int(des.channelizer) == 1) or \
                 (names[dom] == "ddc_noise" and int(des.squelch) == 1)
        if direct:                                  # direct-form banks are ALU-bound: report the fp32 rate too
            fl = (fma_noise if names[dom] == "ddc_noise" else fma_ch) * 2.0 * S * slot
            roof["fp32_tflops"] = round(fl / (avg[dom] * 1e-3) / 1e12, 3)
            roof["fp32_frac"] = round(roof["fp32_tflops"] / FP32_PEAK_TFLOPS, 4)
        roof["build_id"] = build_id()
        if not args.pmc_json:
            # self-carrying evidence: a PMC summary under profiles/ is used iff it was collected on THIS build of
            # libbtgpu.so with this batch size (scripts/pmc_hbm_json.py stamps both); anything else leaves traffic null
            import glob
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm.json")), reverse=True):
                try:
                    pj = json.load(open(cand))
                except (OSError, ValueError):
                    continue
                if pj.get("build_id") == roof["build_id"] and pj.get("slots") == S:
                    args.pmc_json = cand
                    break
        pmc_loaded = None
        if args.pmc_json:
            # HBM bytes of the dominant kernel from rocprofv3 PMC passes of the SAME build and batch size
            # (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; scripts/collect_profiles.sh stamps the build id)
            pmc = json.load(open(args.pmc_json))
            if pmc.get("build_id") != roof["build_id"] or pmc.get("slots") != S:
                raise SystemExit("--pmc-json %s was collected on build %s / %s slots, this run is build %s / %d slots"
                                 % (args.pmc_json, pmc.get("build_id"), pmc.get("slots"), roof["build_id"], S))
            pmc_loaded = pmc
            key = {"ddc_channel": "pfb", "window": "window_kernel", "noise_energy": "noise_stage2_kernel"}.get(names[dom])
            match = [k for k in pmc["kernels"] if key and k.startswith(key)]
            # (the small-M banks run the same kernel template twice -- channel bank and squelch stage 1: the channel bank is the one
            # with the larger traffic, it writes the demodulated stream)
            match.sort(key=lambda k: -pmc["kernels"][k]["hbm_bytes"])
            if match:
                roof["traffic"] = pmc["kernels"][match[0]]["hbm_bytes"]
                roof["traffic_source"] = os.path.basename(args.pmc_json)
        else:
            # no summary of this very build: one of another build counts for the dominant kernel on an equal device-code id
            key = {"ddc_channel": "pfb", "window": "window_kernel", "noise_energy": "noise_stage2_kernel"}.get(names[dom])
            try:
                tb = traffic_by_device_code(key, S) or {}
                tb.pop("_summary", None); tb.pop("_kernel", None)
                roof.update(tb)
            except Exception as e:                      # (no disassembler on the box, ...: traffic stays null)
                roof["traffic_note"] = "no PMC summary of this build; device-code match not possible: %r" % (e,)

        # The dominant kernel of the step since round 6 is exact_rows_kernel (the reference's direct-form DDC on the fp32 matrix pipe
        # over every busy window's rows): MFMA-bound.  Algorithmic flops per launch = rows recomputed in line x 4 real multiply-adds
        # per tap x ntaps x 2; peak = the dense f32-MFMA rate (MI355X_MICROARCH.md: 157.3 TFLOP/s).  The channel bank's HBM figure
        # (rounds 1-5's `roofline`) stays beside it as roofline_bank.
        roof_bank = roof
        if names[dom_all] == "exact" and avg[dom_all] > 0:
            rows_inline = float(blk_stats["rows_per_step"] - blk_long["rows_per_step"])
            fl = rows_inline * 8.0 * des.ntaps_channel
            roof = {"bound": "mfma", "kernel": "exact (exact_rows_kernel<%d>)" % des.decimation, "achieved": round(fl / (avg[dom_all] * 1e-3) / 1e12, 3),
                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (avg[dom_all] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5), "traffic": None,
                    "algorithmic_flops_per_launch": fl, "rows_per_launch": rows_inline, "avg_launch_ms": round(avg[dom_all], 4),
                    "executed_tflops": round(fl / (avg[dom_all] * 1e-3) / 1e12 / (28.0 / 32.0 * des.ntaps_channel / (14.0 * des.decimation) * 1250.0 / (11 * 128)), 3),
                    "note": "executed = the matrix pipe's own rate: 28 of 32 rows, 667 of 700 taps, 1250 of 1408 columns carry results (DESIGN.md 4.4); "
                            "F12: the f32 MFMA runs on the SIMD's vector lanes -- its time and every other instruction's add up",
                    "build_id": roof_bank.get("build_id")}
            pmc_basis = None
            if pmc_loaded is None:
                # no summary of this very build: one whose exact_rows_kernel AND channel bank are instruction-identical to this build's
                try:
                    te, tk = traffic_by_device_code("exact_rows", S), traffic_by_device_code("pfb", S)
                    if te and tk and te["traffic_source"] == tk["traffic_source"]:
                        pmc_loaded, pmc_basis, args.pmc_json = te["_summary"], te["traffic_basis"], te["traffic_source"]
                except Exception:
                    pass
            if pmc_loaded is not None:
                # HBM bytes of the step's exact_rows_kernel launches (two per step: over presence's marks, then the second run's ~4 %):
                # per-launch average x its launches / the steps of the PMC run (= launches of the channel bank)
                ks = pmc_loaded["kernels"]
                ex = [v for k, v in ks.items() if k.startswith("exact_rows")]
                bank = [v for k, v in ks.items() if k.startswith("pfb")]
                if ex and bank and bank[0].get("launches"):
                    roof["traffic"] = round(sum(v["hbm_bytes"] * v.get("launches", 1) for v in ex) / max(v.get("launches", 1) for v in bank), 1)
                    roof["traffic_source"] = os.path.basename(args.pmc_json)
                    if pmc_basis:
                        roof["traffic_basis"] = pmc_basis

        # ---- cpu_baseline + parity: the oracle (a port, NOT the upstream binary) ----
        cpu = None
        parity = {"note": "truth_detected / truth_expected is the recall of the synthetic ground truth; the misses are the "
                          "reference algorithm's -- zero-threshold slicer, no carrier-offset removal (lib/multi_block.cc:171-178): "
                          "bursts beyond about +-40 kHz of offset are lost by the CPU oracle and the GPU alike "
                          "(tests/test_gpu_parity.py::test_cfo_sweep_gpu_equals_oracle, DESIGN.md section 5)",
                  "truth_detected": found, "truth_expected": expected, "hits": int(len(ints)),
                  "records_sha256": hashlib.sha256(np.ascontiguousarray(ints, dtype=np.int64).tobytes()).hexdigest()[:16],
                  # the six key fields (slot, channel, kind, offset, LAP, ac_errors): what the parity contract fixes -- nsym, the seventh
                  # column, depends on how many rows of a window the exact stage recomputed, which batch and range boundaries move
                  "records6_sha256": hashlib.sha256(np.ascontiguousarray(np.asarray(ints, dtype=np.int64).reshape(-1, ints.shape[1] if len(ints) else 7)[:, :6]).tobytes()).hexdigest()[:16]}
        if not args.no_cpu:
            import pyoracle as po
            import paritylib
            # with the block configuration measured, the oracle runs its LE pass too (one run serves both differentials:
            # the LE pass only adds access-address records, lib/multi_sniffer_impl.cc:129-149)
            le_on = bool(args.le)
            o = po.Oracle(fs, fc, args.squelch, po.MODE_SNIFFER, le=le_on)
            ncores = os.cpu_count() or 1
            P = max(2, min(S, args.parity_slots)) if args.parity_slots > 0 else 0
            nhost = max(P, min(S, 64))
            host = seg[margin + H - 1:margin + H - 1 + nhost * slot].cpu().numpy().reshape(-1)
            # single thread (what GNU Radio gives one block), bounded sample
            probe = 2
            t1 = time.perf_counter()
            o.run_stream(host[:2 * probe * slot])
            per_slot = (time.perf_counter() - t1) / probe
            cs = int(max(2, min(nhost, args.cpu_seconds / max(per_slot, 1e-6))))
            t1 = time.perf_counter()
            o.run_stream(host[:2 * cs * slot])
            dt = time.perf_counter() - t1
            cpu = {"value": round(cs * slot / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
                   "sample": "first %d slots (%d samples) of the same capture, oracle/bt_oracle.c single thread" % (cs, cs * slot)}
            if P > 128:
                # keep the all-core leg near a minute on hosts with fewer cores than the 256 of the GPU boxes
                t1 = time.perf_counter()
                o.run_stream(host[:2 * 128 * slot], max_hits=1 << 20, threads=ncores)
                rate = 128 * slot / (time.perf_counter() - t1)
                P = int(max(128, min(P, 80.0 * rate / slot)))
            if P:
                # all host cores over the first P slots: the full-size differential AND the all-core rate
                t1 = time.perf_counter()
                ohits, done = o.run_stream(host[:2 * P * slot], max_hits=1 << 20, threads=ncores)
                dt2 = time.perf_counter() - t1
                cpu["all_cores"] = {"value": round(P * slot / dt2 / 1e6, 4), "cores": ncores, "slots": P,
                                    "seconds": round(dt2, 2)}
                oi_all, _ = bdist.sort_hits(*bdist.hits_to_arrays(ohits))
                oi = oi_all if args.le else oi_all[oi_all[:, 2] == 0]      # headline without --le: classic records only
                gi = ints[ints[:, 0] < P]
                tr = [t for t in truth if t["slot"] < P]
                parity["oracle_slots"] = P
                parity["differential"] = paritylib.differential(gi, oi, tr)
                if exact_all is not None:
                    xi = x_ints[x_ints[:, 0] < P]
                    exact_all["differential"] = paritylib.differential(xi, oi_all, tr)
                    exact_all["records_identical_all_fields"] = bool(exact_all["differential"]["records_identical_all_fields"])
                if block_cfg is not None:
                    bi = b_ints[b_ints[:, 0] < P]
                    block_cfg["differential"] = paritylib.differential(bi, oi_all[oi_all[:, 2] == 0], tr)
                if args.le:
                    ga = collections.Counter(tuple(int(v) for v in r[[0, 1, 3, 4]]) for r in gi if r[2] == 1)
                    ra = collections.Counter(tuple(int(v) for v in r[[0, 1, 3, 4]]) for r in oi_all if r[2] == 1)
                    parity["aa_records"] = {"gpu": int(sum(ga.values())), "ref": int(sum(ra.values())),
                                            "common": int(sum((ga & ra).values())),
                                            "note": "access-address records of a classic-only capture are born from noise symbols"}
                parity["lap_list_equal_ref"] = ("planted records identical" if parity["differential"]["planted_identical"]
                                                else "PLANTED RECORDS DIFFER") + \
                    "; %d / %d other records on one side only" % (
                        parity["differential"]["other_only_gpu"] + parity["differential"]["other_only_ref"],
                        parity["differential"]["other_gpu"] + parity["differential"]["other_ref"])

        out = {
            "metric": "complex-IQ Msamples/s @ 79 ch" if args.workload == "c79" else "complex-IQ Msamples/s @ 8 ch",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "sample_rate": fs, "center_freq": fc, "channels": nch,
                       "slots_per_rank_per_step": S, "samples_per_rank_per_step": S * slot,
                       "squelch_db": args.squelch, "burst_snr_db": args.snr, "piconets": args.piconets,
                       "occupancy": args.occupancy, "cfo_hz": args.cfo_hz, "max_payload_bits": args.max_payload_bits,
                       "mode": "multi_sniffer",
                       "partition": "time x%d, halo %d + margin %d samples" % (world, H - 1, margin),
                       "gather": ("one async all_gather_into_tensor per %d batches (%s, own stream), %d rounds" % (args.gather_every, args.backend, gatherer.rounds)) if gathering else "none",
                       "flags": "ASYNC%s%s%s" % ("|LE" if args.le else "", "|HEADERS" if args.headers else "", ("|EXACT_PAYLOAD" if args.exact_payload else "") + ("" if args.no_timing else ("|TIMING" if args.full_timing else "|TIMING_BANK"))),
                       "channelizer": {1: "direct", 2: "polyphase"}[int(des.channelizer)],
                       "squelch_filter": {1: "direct", 2: "staged"}[int(des.squelch)]},
            "ms_per_step_by_rank": [round(v / args.steps * 1e3, 3) for v in rank_elapsed],
            "classic_only": block_cfg,
            "c8": c8,
            "exact_all": exact_all,
            "verify": verify_obj,
            "host_fed": host_fed,
            "fence_ms": round(fence_ms, 3),
            "step_enqueue_ms": [round(float(v), 3) for v in np.percentile(np.diff(np.array(marks)) * 1e3, [0, 50, 100])],
            "roofline": roof,
            "roofline_bank": roof_bank,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    blk.close()


if __name__ == "__main__":
    main()
