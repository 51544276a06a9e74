#!/bin/bash
# Round-2 evidence in one GPU call:  bash scripts/collect_r02.sh r02_a [tests]
# -> gpurun_out/prof_<tag>/ (C79 bench + rocprof + PMC passes), c8_*, two_rank_*, pytest / smoke logs
set -u
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd $R
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
fi
bash scripts/collect_profiles.sh $TAG > "$OUT/collect.log" 2>&1
tail -1 "$OUT/bench.json" | cut -c1-300
# C8 (BASELINE configs[1]): bench line + kernel stats
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload c8 --steps 100 > "$OUT/c8_bench.json" 2> "$OUT/c8.err"
rm -rf /tmp/kt8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o kt -- python $R/bench.py --workload c8 --steps 100 --no-cpu > "$OUT/c8_bench_under_rocprof.json" 2>> "$OUT/c8.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt8 -name '*kernel_stats.csv' | head -1)" "$OUT/c8_kernel_stats.csv"
cut -c1-200 "$OUT/c8_bench.json"
# N > 1 path on this one device: two ranks, time-partitioned, gathered once per batch
cd $R
python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
cut -c1-200 "$OUT/two_rank_on_one_device_bench.json"
