#!/bin/bash
# pipelined A/B of library builds, alternating:  bash scripts/ab_pipe.sh "base old" [runs]
cp gr-bluetooth_amd/libbtgpu.so /tmp/libbtgpu_base.so
for r in $(seq ${2:-3}); do
  for v in $1; do
    if [ "$v" = base ]; then cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so; else cp gr-bluetooth_amd/libbtgpu_$v.so gr-bluetooth_amd/libbtgpu.so; fi
    python bench.py --no-cpu ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(sys.argv[1], j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms'])" $v
  done
done
cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so
