#!/bin/bash
# block configuration (LE | HEADERS) under the kernel trace, exact stage on / off
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_f; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/kt$v; BTGPU_VERIFY=$v rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/kt$v -o kt -- python $R/bench.py --le --headers --no-cpu --no-block-config > $OUT/bench_block_$v.json 2>> $OUT/err
  python $R/scripts/summarize_rocprof.py "$(find /tmp/kt$v -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_block_$v.csv; echo "verify $v"; head -12 $OUT/kernel_stats_block_$v.csv
  python $R/scripts/timeline.py "$(find /tmp/kt$v -name '*kernel_trace.csv' | head -1)" 40 > $OUT/timeline_block_$v.txt 2>&1
  cp "$(find /tmp/kt$v -name '*memory_copy_trace.csv' | head -1)" $OUT/memcopy_$v.csv 2>/dev/null
  python - $OUT/bench_block_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_ms"), d.get("verify"))
PY
done
tail -42 $OUT/timeline_block_1.txt
