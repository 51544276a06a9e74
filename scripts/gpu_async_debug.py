import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib, torch
synth = importlib.import_module("gr_bluetooth_amd.synth")
fs, fc = 100e6, 2441e6
laps = tuple(0x24D952 + 0x10101 * i for i in range(8))
S = 400
blk = pkg.multi_sniffer(fs, fc, 10.0, False, max_batch_slots=S, flags=0)
H, slot, mg = blk.history(), blk.output_multiple(), blk.design.left_margin
seg, truth = synth.make_segment_torch(fs, fc, 0, S, "cuda", laps=laps, seed=1, left_pad=H - 1 + mg)
torch.cuda.synchronize()
prev = None
for i in range(3):
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=mg)
    r = blk.poll_arrays()
    cur = {}
    for c in (0, 40, 64, 78):
        d = blk.debug_fetch(1, c, 0, 1 << 22); d2 = blk.debug_fetch(6, c, 0, 1 << 22)
        cur[c] = (d, d2)
        print("run", i, "ch", c, "d==d2:", np.array_equal(d[1:], d2[1:len(d)]), len(d), len(d2), "nan", np.isnan(d).sum())
        if prev is not None:
            print("   d same as previous run:", np.array_equal(prev[c][0], d), " d2 same:", np.array_equal(prev[c][1], d2))
    wl = blk.debug_fetch(7, 0, 0, S * 79)
    fr = blk.debug_fetch(8, 0, 0, len(r) + 64)
    fr = fr[:int((wl > 0).sum())] if False else fr
    nf = len(r)
    frs = np.sort(fr[:nf], order='w')
    if prev is not None:
        print("   records same:", np.array_equal(r, prevr), " win_len same:", np.array_equal(wl, pwl))
        n = min(len(frs), len(pfrs))
        same = frs[:n] == pfrs[:n]
        print("   finish recs same:", same.all(), "ndiff", (~same).sum())
        bad = np.nonzero(wl != pwl)[0][:5]
        for wdx in bad:
            a = frs[frs['w'] == wdx]; b2 = pfrs[pfrs['w'] == wdx]
            print("   w", wdx, "len", wl[wdx], pwl[wdx], "rec", a, b2)
    prev, prevr, pwl, pfrs = cur, r, wl, frs
blk.close()
