#!/bin/bash
# pipelined bench under different values of one environment variable:  bash scripts/env_sweep.sh VAR "v1 v2 ..." [runs]   ("-" = unset)
for r in $(seq ${3:-2}); do
  for v in $2; do
    if [ "$v" = "-" ]; then unset $1; else export $1=$v; fi
    python bench.py --no-cpu ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['roofline']['kernel_avg_ms']; print(sys.argv[1], sys.argv[2], j['value'], j['ms_per_step'], 'bank', k['ddc_channel'], 'finish', k['finish'], 'fence', j['fence_ms'], 'steps', j['step_enqueue_ms'])" $1 $v
  done
done
