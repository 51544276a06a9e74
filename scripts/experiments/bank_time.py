"""Average duration of the bank kernel on the C79 workload, no result checks (kernel experiments).
    python scripts/bank_time.py [slots] [batches]"""
import os as _os; _os.environ.setdefault("BTGPU_TIMING", "1")   # btgpu_last_timing is opt-in
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from tests.conftest import load_pkg
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
fs, fc = 100e6, 2441e6
blk = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S, flags=pkg.FLAG_ASYNC)
des = blk.design
seg, _ = synth.make_segment_torch(fs, fc, 0, S, torch.device("cuda", 0), laps=(0x24D952, 0x4831DD), seed=1, snr_db=25.0,
                                  left_pad=des.history - 1 + des.left_margin)
seg = seg.contiguous(); torch.cuda.synchronize()
for i in range(nb):
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin)
    blk.poll_arrays()
    if i == nb // 2:
        blk.flush(); t0 = blk.timing(); k0, n0 = t0.kernel_ms[0], t0.kernel_launches[0]
blk.flush(); t1 = blk.timing()
print("ddc_channel avg ms %.4f over %d launches" % ((t1.kernel_ms[0] - k0) / (t1.kernel_launches[0] - n0), t1.kernel_launches[0] - n0))
