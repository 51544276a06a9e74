#!/bin/bash
# A/B of exact_rows_kernel builds on one box:  bash scripts/experiments/ab_exact.sh "base fulla" [runs]
# (gr-bluetooth_amd/libbtgpu_<name>.so beside the built one; scripts/ubench/exact_mfma_e0 / _e1 = the kernel alone)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for u in e0 e1; do [ -x scripts/ubench/exact_mfma_$u ] && for pt in 4 9 20; do echo "== ubench $u, $pt channels per tile"; ./scripts/ubench/exact_mfma_$u $pt 2304 0 2048 | grep -v "^D = 4"; done; done
cp gr-bluetooth_amd/libbtgpu.so /tmp/libbtgpu_base.so
for r in $(seq ${2:-2}); do
  for v in $1; do
    if [ "$v" = base ]; then cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so; else cp gr-bluetooth_amd/libbtgpu_$v.so gr-bluetooth_amd/libbtgpu.so; fi
    python bench.py --no-cpu --no-block-config --no-c8 --no-exact-all --no-ab --no-host-fed 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(sys.argv[1], 'value', j['value'], 'ms', j['ms_per_step'], 'exact', j['roofline']['avg_launch_ms'], 'bank', j['roofline_bank']['avg_launch_ms'], 'records', j['parity']['records_sha256'], j['parity']['hits'])" $v
  done
done
cp /tmp/libbtgpu_base.so gr-bluetooth_amd/libbtgpu.so
