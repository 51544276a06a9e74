#!/bin/bash
# window kernel alone (--sync: no tail overlap) with the exact stage off / hits only / on
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_c; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 2 1; do
  rm -rf /tmp/kt$v; BTGPU_VERIFY=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$v -o kt -- python $R/bench.py --no-cpu --no-block-config --sync --steps 10 > $OUT/bench_sync_$v.json 2>> $OUT/err
  python $R/scripts/summarize_rocprof.py "$(find /tmp/kt$v -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_sync_$v.csv; echo "verify $v"; head -9 $OUT/kernel_stats_sync_$v.csv
done
