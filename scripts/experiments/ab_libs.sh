#!/bin/bash
# A/B of library builds on one box:  bash scripts/ab_libs.sh "base exp1 exp2" [runs]   (gr-bluetooth_amd/libbtgpu_<name>.so; base = the built one)
cp gr-bluetooth_amd/libbtgpu.so /tmp/libbtgpu_base.so
for r in $(seq ${2:-2}); do
  for v in $1; do
    if [ "$v" = base ]; then cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so; else cp gr-bluetooth_amd/libbtgpu_$v.so gr-bluetooth_amd/libbtgpu.so; fi
    python bench.py --no-cpu --sync ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(sys.argv[1], 'sync bank ms', j['roofline']['kernel_avg_ms']['ddc_channel'])" $v
    python bench.py --no-cpu ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(sys.argv[1], 'pipelined', j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms']['ddc_channel'])" $v
  done
done
cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so
