#!/bin/bash
# where the gathered line loses against the ungathered one: per-step host times of the one-rank RCCL gather
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_p; mkdir -p $O; cd $R
B="python bench.py --gpus 1 --no-cpu --no-block-config --no-ab --no-host-fed"
$B > $O/plain.json 2> $O/plain.err
BENCH_DUMP_STEPS=1 $B --force-gather --backend nccl 2> $O/gather.err | head -1 > $O/gather.json
$B > $O/plain2.json 2>> $O/plain.err
$B --force-gather --backend nccl 2> $O/gather2.err | head -1 > $O/gather2.json
python - <<'PY'
import json
for f in ['plain','gather','plain2','gather2']:
    d=json.loads(open('gpurun_out/r04_p/%s.json'%f).read().strip().split("\n")[-1]); print(f, d['value'], d['ms_per_step'], d['fence_ms'], d['step_enqueue_ms'])
PY
grep "gather step\|step_ms\|last step" $O/gather.err | tail -30
