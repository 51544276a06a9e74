"""First-contact GPU check: C8 sniffer on a synthetic capture, hits and intermediates vs the oracle."""
import os as _os; _os.environ.setdefault("BTGPU_TIMING", "1")   # btgpu_last_timing is opt-in
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib
synth = importlib.import_module("gr_bluetooth_amd.synth")
import pyoracle as po

fs, fc = 8e6, 2476.5e6
iq, truth = synth.make_capture(fs, fc, 24, laps=(0x24D952, 0x4831DD), seed=1, snr_db=25)
o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
t = time.time(); ohits, done = o.run_stream(iq); print("oracle", time.time() - t, "s", len(ohits), "hits")
b = pkg.multi_sniffer(fs, fc, 10.0)
print("design", b.design.history, b.design.ddc_out)
t = time.time(); b.push(iq); print("gpu push", time.time() - t)
ghits = b.poll()
print("gpu hits", len(ghits))
ok = [h.key() for h in ohits] == [h.key() for h in ghits]
print("HIT LIST IDENTICAL:", ok)
if not ok:
    for h in ohits: print("O", h.key(), h.snr)
    for h in ghits: print("G", h.key(), h.snr_db)
tm = b.timing(); print("timing ms", list(tm.kernel_ms)[:5], tm.total_ms, tm.slots)
# intermediates of the last batch: Y / d for one channel, window 0..
S = done
for ch in (71, 75, 78):
    Y = b.debug_fetch(0, ch, 0, 1 << 24)
    d = b.debug_fetch(1, ch, 0, 1 << 24)
    worst = 0.0; dd = 0.0
    for k in (0, 7, 13, S - 1):
        win = o.window(iq, k)
        oy, e = o.channel_samples(win, ch)
        g0 = 1250 * k
        gy = Y[g0:g0 + len(oy)]
        sgn = 1.0 if np.allclose(gy[100:110], oy[100:110], atol=1e-6) else -1.0
        worst = max(worst, float(np.max(np.abs(gy * sgn - oy))))
        exact = np.array_equal((gy * sgn).view(np.float32), oy.view(np.float32))
        od = o.demod(oy)
        gd = d[g0:g0 + len(od)].copy(); gd[0] = 0
        dd = max(dd, float(np.max(np.abs(gd - od))))
        print("ch", ch, "slot", k, "Y exact:", exact, "sign", sgn, "demod exact:", np.array_equal(gd, od))
    print("ch", ch, "max |dY|", worst, "max |dd|", dd)
eon = b.debug_fetch(2, 0, 0, 1 << 20); eoff = b.debug_fetch(3, 0, 0, 1 << 20); snr = b.debug_fetch(4, 0, 0, 1 << 20)
nch = b.design.high_channel - b.design.low_channel + 1
for k in (0, 7, S - 1):
    win = o.window(iq, k)
    for ch in (71, 78):
        oy, e = o.channel_samples(win, ch)
        okk, osnr, ooff = o.check_snr(win, ch, e)
        i = k * nch + (ch - b.design.low_channel)
        print("slot", k, "ch", ch, "E_on rel", abs(eon[i] - e) / max(e, 1e-300), "E_off rel", abs(eoff[i] - ooff) / max(ooff, 1e-300), "snr", snr[i], osnr)
