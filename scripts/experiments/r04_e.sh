#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_e; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python scripts/gpu_fuzz_fast.py 800 77 > $OUT/fuzz_fast_800_seed77.txt 2>&1; tail -1 $OUT/fuzz_fast_800_seed77.txt
timeout 900 python scripts/gpu_fuzz_fast.py 400 32 > $OUT/fuzz_fast_400_seed32.txt 2>&1; tail -1 $OUT/fuzz_fast_400_seed32.txt
