#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_j; mkdir -p $OUT; cd $R
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config > $OUT/gather.json 2> $OUT/gather.err; echo rc $?
tail -5 $OUT/gather.err; wc -c $OUT/gather.json
