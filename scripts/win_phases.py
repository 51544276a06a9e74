import os, sys
os.environ["BTGPU_PFB_PROF"] = "1"; os.environ["BTGPU_WIN_PROF"] = "1"
sys.path.insert(0, "/root/repo")
import importlib, numpy as np, torch
from tests.conftest import load_pkg
pkg = load_pkg(); synth = importlib.import_module("gr_bluetooth_amd.synth")
S = 1600; fs, fc = 100e6, 2441e6
blk = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S)
des = blk.design; dev = torch.device("cuda", 0)
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
seg, _ = synth.make_segment_torch(fs, fc, 0, S, dev, laps=laps, seed=1, snr_db=25.0, left_pad=des.history - 1 + des.left_margin)
seg = seg.contiguous(); torch.cuda.synchronize()
for _ in range(3):
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin); blk.flush(); blk.poll_arrays()
c = blk.debug_fetch(9, 0, 0, 1 << 24).astype(np.float64).reshape(-1, 8)
w = c[:800].sum(axis=0)   # window kernel uses blocks 0..799 (pfb marks are in the same buffer: subtract not possible) 
print(w / w.sum() * 100)
