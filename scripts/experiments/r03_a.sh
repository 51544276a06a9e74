#!/bin/bash
# Round 3, first contact of the three-stage pipeline: tests, smoke, bench, A/B runs of the pipeline knobs,
# kernel trace.  bash scripts/r03_a.sh [tag]
set -u
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
ab() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["parity"]["records_sha256"], d["parity"]["hits"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
ab default BENCH_DUMP_STEPS=1
ab pipe0 BTGPU_PIPE=0
ab ctx2 BTGPU_CTX=2
ab postprio0 BTGPU_POST_PRIO=0
ab wide512 BTGPU_BANK_THREADS=512
ab default2 A=1
python bench.py --no-cpu --headers > "$OUT/bench_headers.json" 2> "$OUT/bench_headers.err"; cut -c1-160 "$OUT/bench_headers.json"
python bench.py --no-cpu --workload c8 --steps 100 > "$OUT/bench_c8.json" 2> "$OUT/bench_c8.err"; cut -c1-160 "$OUT/bench_c8.json"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
TR=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python $R/scripts/timeline.py "$TR" > "$OUT/timeline.txt" 2>&1
# keep the last 400 kernel records of the trace (the timed region's tail) for an overlap picture
tail -400 "$TR" | cut -d, -f1-20 > "$OUT/kernel_trace_tail.csv"
cat "$OUT/kernel_stats.csv" | head -12
cd $R
python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; cut -c1-400 "$OUT/bench_full.json"
