#!/bin/bash
# window kernel behind the previous batch's tail (BTGPU_WIN_WAIT_TAIL=1): headline and block configuration, with and without
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_t; mkdir -p $O; cd $R
B="python bench.py --no-cpu --no-host-fed --no-ab --no-block-config"
run() { # name env args
  n=$1; e=$2; shift; shift
  env $e $B "$@" > $O/$n.json 2>> $O/err; python -c "
import json; d=json.loads(open('$O/$n.json').read().strip().split('\n')[-1]); print('$n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
run block_plain X=1 --le --headers
run block_wait BTGPU_WIN_WAIT_TAIL=1 --le --headers
run head_plain X=1
run head_wait BTGPU_WIN_WAIT_TAIL=1
run block_plain2 X=1 --le --headers
run block_wait2 BTGPU_WIN_WAIT_TAIL=1 --le --headers
