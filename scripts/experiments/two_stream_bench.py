"""Does the device overlap the latency-bound window kernel of one batch with the bank kernel of
another?  Two independent blocks (own streams) fed alternately vs one block, same total work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
fs, fc = 100e6, 2441e6
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 20
dev = torch.device("cuda", 0)
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
blks = [pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S, flags=pkg.FLAG_ASYNC) for _ in range(nblk)]
des = blks[0].design
H, slot, margin = des.history, des.samples_per_slot, des.left_margin
seg, _ = synth.make_segment_torch(fs, fc, 0, S, dev, laps=laps, seed=1, snr_db=25.0, left_pad=H - 1 + margin)
seg = seg.contiguous(); n = seg.shape[0]
torch.cuda.synchronize()


def run(k):
    nh = 0
    for i in range(k):
        for b in blks:
            b.process_device(seg.data_ptr(), n, 0, S, left_margin=margin)
        for b in blks:
            nh += len(b.poll_arrays())
    for b in blks:
        b.flush(); nh += len(b.poll_arrays())
    torch.cuda.synchronize()
    return nh


t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    run(2)
t0 = time.perf_counter(); nh = run(steps); el = time.perf_counter() - t0
print("blocks %d x %d slots: %.3f ms per %d slots, %.1f Gsamples/s, %d records" %
      (nblk, S, el / steps * 1e3, nblk * S, nblk * S * slot * steps / el / 1e9, nh))
