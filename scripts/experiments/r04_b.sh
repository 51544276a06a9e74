#!/bin/bash
# exact stage: per-kernel durations (rocprofv3 kernel trace) with the stage on, bench lines on / off
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_b; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 1 0 2; do BTGPU_VERIFY=$v timeout 300 python $R/bench.py --no-cpu --no-block-config --full-timing > $OUT/bench_nocpu_verify$v.json 2>> $OUT/bench.err; done
rm -rf /tmp/kt; BTGPU_VERIFY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv
python $R/scripts/timeline.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" 40 > $OUT/timeline.txt 2>&1; tail -45 $OUT/timeline.txt
for f in $OUT/bench_nocpu_*.json; do echo "== $f"; python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_ms"), d.get("verify"))
PY
done
