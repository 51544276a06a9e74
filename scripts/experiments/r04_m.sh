#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_m; mkdir -p $OUT; cd $R
for v in 1 0; do echo "== c8 verify $v"; BTGPU_VERIFY=$v python bench.py --workload c8 --steps 100 --no-cpu --no-block-config --no-ab --no-host-fed --full-timing 2>> $OUT/err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify'])"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "randomised or exact_stage or small_rates or c8 or hit_list" 2>&1 | tail -3
timeout 600 python scripts/gpu_fuzz_fast.py 160 5 2>&1 | tail -1
