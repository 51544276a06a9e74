"""Fast path (polyphase channel bank + staged squelch) vs direct path vs oracle at C79."""
import os as _os; _os.environ.setdefault("BTGPU_TIMING", "1")   # btgpu_last_timing is opt-in
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg
pkg = load_pkg()
import importlib
synth = importlib.import_module("gr_bluetooth_amd.synth")
import pyoracle as po
fs, fc = 100e6, 2441e6
laps = tuple(0x24D952 + 0x10101 * i for i in range(6))
S = 9
iq, truth = synth.make_capture(fs, fc, S, laps=laps, seed=79, snr_db=25, occupancy=0.6)
o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
t = time.time(); ohits, done = o.run_stream(iq, threads=32); print("oracle", round(time.time() - t, 2), "s", len(ohits), "hits")
res = {}
for name, ch, sq in (("direct", pkg.CHANNELIZER_DIRECT, pkg.SQUELCH_DIRECT), ("pfb+direct", pkg.CHANNELIZER_POLYPHASE, pkg.SQUELCH_DIRECT),
                     ("direct+staged", pkg.CHANNELIZER_DIRECT, pkg.SQUELCH_STAGED), ("fast", pkg.CHANNELIZER_POLYPHASE, pkg.SQUELCH_STAGED)):
    b = pkg.multi_sniffer(fs, fc, 10.0, False, channelizer=ch, squelch=sq, flags=pkg.FLAG_DEBUG_Y)
    b.push(iq)
    hits = b.poll()
    tm = b.timing()
    nch = 79
    res[name] = dict(keys=[h.key() for h in hits], Y={c: b.debug_fetch(0, c, 0, 1 << 22) for c in (0, 39, 78)},
                     d={c: b.debug_fetch(1, c, 0, 1 << 22) for c in (0, 39, 78)},
                     eon=b.debug_fetch(2, 0, 0, S * nch), eoff=b.debug_fetch(3, 0, 0, S * nch), snr=b.debug_fetch(4, 0, 0, S * nch))
    print(name, "hits", len(hits), "== oracle:", res[name]["keys"] == [h.key() for h in ohits], "kernel ms", [round(x, 3) for x in list(tm.kernel_ms)[:5]])
    b.close()
ref = res["direct"]
for name in ("pfb+direct", "direct+staged", "fast"):
    r = res[name]
    for c in (0, 39, 78):
        ny = min(len(r["Y"][c]), len(ref["Y"][c]))
        ey = np.linalg.norm(r["Y"][c][:ny] - ref["Y"][c][:ny]) / np.linalg.norm(ref["Y"][c][:ny])
        nd = min(len(r["d"][c]), len(ref["d"][c]))
        ed = np.max(np.abs(r["d"][c][1:nd] - ref["d"][c][1:nd]))
        print(name, "ch", c, "Y rel-L2", ey, "max|dd|", ed)
    m = np.isfinite(ref["snr"]) & (ref["eoff"] > 0)
    print(name, "E_on max rel", np.max(np.abs(r["eon"][m] - ref["eon"][m]) / ref["eon"][m]), "E_off max rel", np.max(np.abs(r["eoff"][m] - ref["eoff"][m]) / ref["eoff"][m]),
          "snr max abs dB", np.max(np.abs(r["snr"][m] - ref["snr"][m])))
ok = [h.key() for h in ohits]
for name in ("pfb+direct", "fast"):
    for a, b in zip(ok, res[name]["keys"]):
        if a != b:
            print(name, "oracle", a, "gpu", b)
