#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_k; mkdir -p $OUT; cd $R
python bench.py --no-cpu --no-block-config > $OUT/bench.json 2> $OUT/err; tail -3 $OUT/err
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step']); print(d['host_fed'])"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config --no-host-fed 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('gathered', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_host_block_gpu.py tests/test_host_capi.py -q -m gpu 2>&1 | tail -3
