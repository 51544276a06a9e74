"""Time the hop-reversal primitives on the GPU (btgpu_hopseq_*): table generation, candidate
initialisation, one winnowing step.  GPU only; run under rocprofv3 --kernel-trace --stats for the
kernel durations.   python scripts/hop_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pkg
pkg = load_pkg()
addr = (0xAF << 24) | 0x24D952
t0 = time.perf_counter(); seq = pkg.HopSequence(addr); t1 = time.perf_counter()
seqs = [seq]
for i in range(5):                      # steady state (allocation reused by the driver's pool)
    a5 = time.perf_counter(); s = pkg.HopSequence(addr ^ (i + 1)); b5 = time.perf_counter()
    seqs.append(s)
ch = int(seq.lookup([12345])[0])
a = time.perf_counter(); n0 = seq.init_candidates(ch, 12345 & 0x3F); c = time.perf_counter()
n1 = seq.winnow(77, int(seq.lookup([12345 + 77])[0])); d = time.perf_counter()
print("create (first, incl. hipMalloc of 128 MiB) %.2f ms, create (6th) %.2f ms, init_candidates %.3f ms -> %d, winnow %.3f ms -> %d"
      % ((t1 - t0) * 1e3, (b5 - a5) * 1e3, (c - a) * 1e3, n0, (d - c) * 1e3, n1))
