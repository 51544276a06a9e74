#!/bin/bash
# After the last collection (same library, build be7b054c...): the GPU tests again with the handle-after-handle test in, and more device fuzz
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r06_zz2
mkdir -p "$OUT"; cd $R
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
timeout 700 python scripts/gpu_judge_seamless.py 100000 38001 --mode 100 --seconds 560 > "$OUT/judge_seamless_100M_560s_seed38001.txt" 2>&1; tail -1 "$OUT/judge_seamless_100M_560s_seed38001.txt" | cut -c1-300
timeout 400 python scripts/gpu_judge_seamless.py 100000 38002 --weak --seconds 280 > "$OUT/judge_weak_beside_280s_seed38002.txt" 2>&1; tail -1 "$OUT/judge_weak_beside_280s_seed38002.txt" | cut -c1-300
timeout 500 python scripts/gpu_fuzz_adversarial.py 100000 38003 --seconds 380 > "$OUT/fuzz_adversarial_380s_seed38003.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_380s_seed38003.txt" | cut -c1-300
