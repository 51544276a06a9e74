"""Does the SECOND handle of a process run slower than the first?  (bench.py's c8 leg reads 15 G where --workload c8 reads 19 G.)
python scripts/experiments/second_handle.py [c8|c79 ...]   -- one handle per argument, in order, each: create, 150 ms of load, 30 timed steps, close."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
W = {"c8": (8e6, 2476.5e6, 16384), "c79": (100e6, 2441e6, 2304), "c8s": (8e6, 2476.5e6, 2048)}
laps = tuple((0x24D952 + 0x10101 * i) & 0xFFFFFF for i in range(8))
gen = dict(laps=laps, seed=1, snr_db=25.0, occupancy=0.3, cfo_hz=75e3, max_payload_bits=2745)
keep = os.environ.get("KEEP_SEG") == "1"
segs = []
for name in sys.argv[1:] or ["c8", "c8"]:
    fs, fc, S = W[name]
    b = pkg.multi_sniffer(fs, fc, 10.0, False, device=0, max_batch_slots=S, flags=pkg.FLAG_ASYNC | pkg.FLAG_LE | pkg.FLAG_HEADERS | pkg.FLAG_TIMING_BANK)
    d = b.design
    seg, _ = synth.make_segment_torch(fs, fc, 0, S, "cuda:0", left_pad=d.history - 1 + d.left_margin, **gen)
    seg = seg.to("cuda:0").contiguous(); torch.cuda.synchronize()
    def step(last):
        b.process_device(seg.data_ptr(), seg.shape[0], 0, S, left_margin=d.left_margin)
        if last: b.flush()
        return len(b.poll_arrays())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: step(False)
    step(True); torch.cuda.synchronize()
    tm0 = b.timing(); K = 30; t0 = time.perf_counter()
    for i in range(K): step(i == K - 1)
    torch.cuda.synchronize(); el = time.perf_counter() - t0; tm1 = b.timing()
    ex = (tm1.kernel_ms[7] - tm0.kernel_ms[7]) / max(1, tm1.kernel_launches[7] - tm0.kernel_launches[7])
    bk = (tm1.kernel_ms[0] - tm0.kernel_ms[0]) / max(1, tm1.kernel_launches[0] - tm0.kernel_launches[0])
    print("%-4s handle: %.3f ms per step = %.2f Gsamples/s, exact rows %.3f ms, bank %.3f ms" % (name, el / K * 1e3, S * d.samples_per_slot * K / el / 1e9, ex, bk), flush=True)
    b.close()
    if keep: segs.append(seg)
    else: del seg
