#!/bin/bash
# A/B of the bank kernel's BTGPU_PFB_DBG experiment bits (sync mode: kernel time alone):  bash scripts/dbg_sweep.sh "0 8 16 24"
for d in $1; do
  BTGPU_PFB_DBG=$d python bench.py --no-cpu --sync 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('dbg', sys.argv[1], j['ms_per_step'], j['roofline']['kernel_avg_ms']['ddc_channel'])" $d
done
