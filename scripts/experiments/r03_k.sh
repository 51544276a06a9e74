#!/bin/bash
# A/B of the lean epilogue / 32-bit store offsets (bank) and the six-output stage 2: bank times, kernel stats, parity subset
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_k}
mkdir -p "$OUT"
cd $R
for v in run256 run256e run256d run256 run256e; do echo -n "$v: "; BTGPU_BANK=$v python scripts/bank_time.py 2304 40 2>&1 | tail -1; done | tee "$OUT/bank_times.txt"
python bench.py --no-cpu --no-block-config 2>/dev/null | grep "^{" | tee "$OUT/bench_quick.json" | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"; head -8 "$OUT/kernel_stats.csv"
cd $R
if [ "${2:-tests}" = "tests" ]; then python -m pytest tests -m gpu -x -q -k "c79 or fast_path or smoke or full_size or silence or cfo" 2>&1 | tail -3 | tee "$OUT/pytest_subset.log"; fi
