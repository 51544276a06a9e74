"""Per-phase cycle split of the fused bank kernel (BTGPU_PFB_PROF diagnostics).

Runs the C79 bench workload synchronously for a few batches with the in-kernel cycle marks
enabled and prints the share of wave-cycles per phase.  GPU only.
    python scripts/pfb_phases.py [slots] [batches]
"""
import os as _os; _os.environ.setdefault("BTGPU_TIMING", "1")   # btgpu_last_timing is opt-in
import os, sys
os.environ["BTGPU_PFB_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
import torch
from tests.conftest import load_pkg

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
fs, fc = 100e6, 2441e6
blk = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S)
des = blk.design
dev = torch.device("cuda", 0)
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
seg, _ = synth.make_segment_torch(fs, fc, 0, S, dev, laps=laps, seed=1, snr_db=25.0,
                                  left_pad=des.history - 1 + des.left_margin)
seg = seg.contiguous()
torch.cuda.synchronize()
for _ in range(nb):
    blk.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=des.left_margin)
    blk.flush()
    blk.poll_arrays()
raw = blk.debug_fetch(9, 0, 0, 1 << 24).astype(np.float64).reshape(-1, 8)
variant = os.environ.get("BTGPU_BANK", "run256")
if variant in ("legacy", "wide"):
    c = raw.sum(axis=0)
    names = ["0 stage input", "A branch FIR (+noise)", "B1 DFT pass + twiddle", "B2 DFT pass -> Y", "C' noise bin loads",
             "C epilogue (demod, sums) + Z stores + barrier", "copy-out d / dcol + tile sums", "-"]
    tot = c.sum()
    for n, v in zip(names, c):
        print("%-24s %6.2f %%" % (n, 100.0 * v / tot))
else:
    # pfb100f_kernel: [workgroup][wave][8]; marks 0..4 = cycles up to the barrier that ends: copy-out + staging | A | B1 | B2 | C
    nw = 5 if variant == "run320" else 4
    kt = 10 if variant in ("run256", "run256a", "run256d", "run256e") else 5   # tiles per workgroup (bank_launch.h)
    psl = 16 if kt >= 10 else 8                                            # slots per wave (pfb100f.hip.h)
    raw = raw.reshape(-1)
    used = raw[: (len(raw) // (nw * psl)) * nw * psl].reshape(-1, nw, psl)
    used = used[used[:, 0, :].sum(axis=1) > 0]
    n = len(used) * kt * nb
    if psl == 16:
        # marks 5..11 end the WORK of an interval, marks 0..4 the barrier behind it
        rows = [("stage span -> LDS", 5, None), ("flush tile n-1 (Z, d, dcol, sums)", 6, 0), ("prefetch + rot loads issue", 7, None),
                ("A march", 8, 1), ("B1", 9, 2), ("B2", 10, 3), ("C epilogue", 11, 4)]
        print("cycles per tile and wave: work | barrier wait (mean over %d workgroups x %d tiles)" % (len(used), kt * nb))
        for name, kw, kb in rows:
            print("%-36s " % name + " ".join("w%d %6.0f | %5.0f" % (w, used[:, w, kw].sum() / n, (used[:, w, kb].sum() / n) if kb is not None else 0) for w in range(nw)))
        tot = used[:, :, :12].sum(axis=(0, 2))
    else:
        names = ["copy-out(n-1) + stage + barrier", "A march + barrier", "B1 + barrier", "B2 + barrier", "C epilogue + Z + barrier"]
        tot = used[:, :, :5].sum(axis=(0, 2))
        print("cycles per tile and wave (mean over %d workgroups x %d tiles):" % (len(used), kt * nb))
        for k, nm in enumerate(names):
            print("%-34s " % nm + " ".join("w%d %7.0f (%4.1f %%)" % (w, used[:, w, k].sum() / n, 100 * used[:, w, k].sum() / tot[w]) for w in range(nw)))
    print("tile life (cycles), per wave:", [round(float(t) / n) for t in tot])
tm = blk.timing()
print("ddc_channel avg ms %.4f (with marks enabled)" % (tm.kernel_ms[0] / max(tm.kernel_launches[0], 1)))
