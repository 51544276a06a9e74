#!/bin/bash
# A/B: the tail stream(s) confined to n CUs (BTGPU_TAIL_CUS, every (256/n)-th CU) against the whole device -- headline only
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for r in 1 2; do for n in 0 128 64 32; do
  if [ $n = 0 ]; then unset BTGPU_TAIL_CUS; else export BTGPU_TAIL_CUS=$n; fi
  python bench.py --no-cpu --no-block-config --no-c8 --no-exact-all --no-ab --no-host-fed 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tail_cus', sys.argv[1], 'value', j['value'], 'ms', j['ms_per_step'], 'exact', j['roofline']['avg_launch_ms'], 'bank', j['roofline_bank']['avg_launch_ms'], 'records', j['parity']['records_sha256'])" $n
done; done
