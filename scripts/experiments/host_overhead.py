"""Where does the wall time of a pipelined bench step go on the host?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib, torch
synth = importlib.import_module("gr_bluetooth_amd.synth")
bdist = importlib.import_module("gr_bluetooth_amd.dist")
fs, fc = 100e6, 2441e6
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
S = 1600
blk = pkg.multi_sniffer(fs, fc, 10.0, False, max_batch_slots=S, flags=pkg.FLAG_ASYNC)
H, slot, mg = blk.history(), blk.output_multiple(), blk.design.left_margin
seg, truth = synth.make_segment_torch(fs, fc, 0, S, "cuda", laps=laps, seed=1, left_pad=H - 1 + mg)
torch.cuda.synchronize()
for _ in range(2):
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=mg)
blk.flush(); blk.poll_arrays()
torch.cuda.synchronize()
K = 12
tp, tq, tc = [], [], []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=mg)
    b = time.perf_counter()
    r = blk.poll_arrays()
    c = time.perf_counter()
    ints, snr = bdist.struct_to_arrays(r)
    d = time.perf_counter()
    tp.append(b - a); tq.append(c - b); tc.append(d - c)
blk.flush(); r = blk.poll_arrays()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("wall/step %.3f ms" % ((t1 - t0) / K * 1e3))
print("process_device ms:", " ".join("%.2f" % (x * 1e3) for x in tp))
print("poll_arrays    ms:", " ".join("%.2f" % (x * 1e3) for x in tq))
print("to_arrays      ms:", " ".join("%.2f" % (x * 1e3) for x in tc))
