#!/bin/bash
# The last call of the round, the committed library (build be7b054c...): bench.py as the driver runs it, then device fuzz to the end of the budget
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r06_zz3
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
/usr/bin/time -v python $R/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_form.json" 2> "$OUT/bench_driver_form.err"; tail -1 "$OUT/bench_driver_form.json" | cut -c1-200; grep -E "Elapsed|Maximum resident" "$OUT/bench_driver_form.err"
cd $R
timeout 500 python scripts/gpu_judge_seamless.py 100000 39001 --seconds 380 > "$OUT/judge_seamless_mix_380s_seed39001.txt" 2>&1; tail -1 "$OUT/judge_seamless_mix_380s_seed39001.txt" | cut -c1-300
timeout 500 python scripts/gpu_fuzz_adversarial.py 100000 39002 --seconds 380 --wide > "$OUT/fuzz_adversarial_wide_380s_seed39002.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_wide_380s_seed39002.txt" | cut -c1-300
timeout 400 python scripts/gpu_fuzz_parity.py 120 > "$OUT/fuzz_parity_120.txt" 2>&1; tail -1 "$OUT/fuzz_parity_120.txt" | cut -c1-300
