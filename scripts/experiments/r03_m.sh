#!/bin/bash
# A/B of the window kernel's chunk rows (29: four workgroups per CU; 38: three), kernel stats of both, then the GPU tests
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_m}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for rows in 29 38 29; do
  rm -rf /tmp/kt
  BTGPU_WIN_ROWS=$rows rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config > "$OUT/bench_rows$rows.json" 2> "$OUT/bench.err"
  python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_rows$rows.csv"
  echo "== rows $rows: $(python -c "import json;d=json.loads(open('$OUT/bench_rows$rows.json').read().strip().split(chr(10))[-1]);print(d['value'], d['ms_per_step'])")"; sed -n 2,5p "$OUT/kernel_stats_rows$rows.csv"
done
cd $R
for rows in 29 38 29 38; do echo -n "rows $rows (no profiler): "; BTGPU_WIN_ROWS=$rows python bench.py --no-cpu --no-block-config 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print(d['value'], d['ms_per_step'])"; done | tee "$OUT/ab.txt"
if [ "${2:-tests}" = "tests" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee "$OUT/pytest_gpu.log"; fi
