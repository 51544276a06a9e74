"""Which call of the end-of-region fence is slow, and is it a per-process or a per-fence effect?  GPU only.
    python scripts/fence_probe.py [rounds]
"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.conftest import load_pkg

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
fs, fc, S = 8e6, 2476.5e6, 16384
blk = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S, flags=pkg.FLAG_ASYNC)
des = blk.design
dev = torch.device("cuda", 0)
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
seg, _ = synth.make_segment_torch(fs, fc, 0, S, dev, laps=laps, seed=1, snr_db=25.0, left_pad=des.history - 1 + des.left_margin)
seg = seg.contiguous()
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
mode = os.environ.get("PROBE", "torch")
for r in range(rounds):
    t0 = time.perf_counter()
    for i in range(20):
        blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin)
        if i == 19:
            blk.flush()
        blk.poll_arrays()
    t1 = time.perf_counter()
    if mode == "hip":
        hip.hipDeviceSynchronize()
    else:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%s round %2d loop %.2f ms  sync1 %.3f ms  sync2 %.3f ms" % (mode, r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
