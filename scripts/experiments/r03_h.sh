#!/bin/bash
set -u
TAG=${1:-r03_h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
for v in run256 run256b run256c run256d run256e run256; do echo -n "$v: "; BTGPU_BANK=$v python scripts/bank_time.py 2304 40 2>&1 | tail -1; done | tee "$OUT/bank_times.txt"
python bench.py --no-cpu --no-block-config 2> /dev/null | grep "^{" | cut -c1-130
python bench.py --no-cpu --no-block-config 2> /dev/null | grep "^{" | cut -c1-130
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config 2> /dev/null | grep "^{" | cut -c1-130
python -m pytest tests -m gpu -x -q -k "hit_list_bit_exact or intermediates or full_size or squelch or le_pass" 2>&1 | tail -2
