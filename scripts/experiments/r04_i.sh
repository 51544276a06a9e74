#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_i; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in "--le --headers" ""; do
  for v in 1 0; do
  echo "== verify $v cfg '$cfg'"; BTGPU_VERIFY=$v python $R/bench.py --no-cpu --no-block-config --full-timing $cfg 2>> $OUT/err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --le --headers --no-cpu --no-block-config > $OUT/bench_block.json 2>> $OUT/err
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_block.csv; head -10 $OUT/kernel_stats_block.csv
