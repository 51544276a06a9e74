#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_h; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pr in 3 1 0; do
  for cfg in "--le --headers" ""; do
    echo "== fin_prio $pr cfg '$cfg'"; BTGPU_FIN_PRIO=$pr python $R/bench.py --no-cpu --no-block-config --full-timing $cfg 2>> $OUT/err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done
