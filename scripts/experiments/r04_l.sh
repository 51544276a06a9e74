#!/bin/bash
# deferred squelch A/B + the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_l; mkdir -p $OUT; cd $R
for df in 1 0; do for cfg in "" "--le --headers"; do
  echo "== defer $df cfg '$cfg'"; BTGPU_DEFER=$df python bench.py --no-cpu --no-block-config --no-host-fed --full-timing $cfg 2>> $OUT/err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['parity'])"
done; done
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config --no-host-fed > /dev/null 2>> $OUT/err
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv
python $R/scripts/timeline.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" 24 > $OUT/timeline.txt 2>&1; tail -26 $OUT/timeline.txt
