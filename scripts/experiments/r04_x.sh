#!/bin/bash
# environment-only A/Bs on the committed build: workgroups of the exact stage's DDC
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_x; mkdir -p $O; cd $R
B="python bench.py --no-cpu --no-host-fed --no-ab --no-block-config --no-parity"
python bench.py --help 2>/dev/null | grep -q -- "--no-parity" || B="python bench.py --no-cpu --no-host-fed --no-ab --no-block-config"
run() { n=$1; e=$2; env $e $B > $O/$n.json 2>> $O/err; python -c "
import json; d=json.loads(open('$O/$n.json').read().strip().split('\n')[-1]); print('$n', '$e', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step; bank', d['roofline']['avg_launch_ms'], 'ms')"; }
run grid2048 X=1
run grid1024 BTGPU_VERIFY_GRID=1024
run grid4096 BTGPU_VERIFY_GRID=4096
[ $SECONDS -lt 60 ] && run grid2048b X=1
[ $SECONDS -lt 75 ] && run grid1024b BTGPU_VERIFY_GRID=1024
