"""Where the per-batch cost of the record gather goes (single-rank nccl group on one GPU): host time of post() / collect(),
and the step time with the gather every batch, every 8th batch, and off."""
import os, sys, time, socket, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import load_pkg
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
pkg = load_pkg(); synth = importlib.import_module("gr_bluetooth_amd.synth"); bd = importlib.import_module("gr_bluetooth_amd.dist")
fs, fc, S = 100e6, 2441e6, 2304
blk = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S, flags=pkg.FLAG_ASYNC)
des = blk.design
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
seg, _ = synth.make_segment_torch(fs, fc, 0, S, dev, laps=laps, seed=1, snr_db=25.0, cfo_hz=75e3, max_payload_bits=2745,
                                  left_pad=des.history - 1 + des.left_margin)
seg = seg.contiguous(); torch.cuda.synchronize()
def run(every, cap, steps=40):
    g = bd.HitGatherer(cap=cap, device=dev, force=True)
    tp = tc = 0.0
    for i in range(10):
        blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin); blk.poll_arrays()
    blk.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for i in range(steps):
        blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin)
        ints, snr = bd.struct_to_arrays(blk.poll_arrays())
        if every and i % every == every - 1:
            a = time.perf_counter()
            if g.full: r = g.collect(); n += len(r[0])
            b = time.perf_counter()
            g.post(ints, snr)
            c = time.perf_counter(); tc += b - a; tp += c - b
        elif every:
            g.backlog_i = np.concatenate([g.backlog_i, ints]); g.backlog_s = np.concatenate([g.backlog_s, snr])
    blk.flush(); blk.poll_arrays(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("gather every %d batches, cap %5d: %.3f ms/step  (host: collect %.3f ms, post %.3f ms per call)" %
          (every, cap, el / steps * 1e3, tc / max(1, steps // max(every, 1)) * 1e3, tp / max(1, steps // max(every, 1)) * 1e3))
run(0, 8192); run(1, 8192); run(1, 1024); run(8, 8192); run(0, 8192)
dist.destroy_process_group()
