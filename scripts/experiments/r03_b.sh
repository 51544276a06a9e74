#!/bin/bash
# Round 3, run-kernel A/B: tests, then bench with each bank variant, kernel stats of the default.
set -u
TAG=${1:-r03_b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
ab() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["parity"]["records_sha256"], d["parity"]["hits"], d["parity"]["truth_detected"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
ab run320 BTGPU_BANK=run320
ab run256 BTGPU_BANK=run256
ab legacy BTGPU_BANK=legacy
ab run256_b BTGPU_BANK=run256
ab run320_b BTGPU_BANK=run320
python bench.py --no-cpu --workload c8 --steps 100 > "$OUT/bench_c8.json" 2> "$OUT/bench_c8.err"; cut -c1-160 "$OUT/bench_c8.json"
cd /tmp && export TMPDIR=/tmp
for v in run256 run320; do
rm -rf /tmp/kt
BTGPU_BANK=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu > "$OUT/bench_under_rocprof_$v.json" 2> "$OUT/rocprof_$v.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_$v.csv"
head -8 "$OUT/kernel_stats_$v.csv"
done
python $R/scripts/timeline.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" > "$OUT/timeline.txt" 2>&1
