set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_n; mkdir -p $OUT; cd $R
timeout 420 python scripts/gpu_fuzz_adversarial.py 500 31001 --rates 8,8,20 > $OUT/gpu_fuzz_adversarial_500_seed31001.txt 2>&1; tail -2 $OUT/gpu_fuzz_adversarial_500_seed31001.txt | cut -c1-600
timeout 420 python scripts/gpu_fuzz_adversarial.py 80 31002 --rates 100 > $OUT/gpu_fuzz_adversarial_80_seed31002_100M.txt 2>&1; tail -2 $OUT/gpu_fuzz_adversarial_80_seed31002_100M.txt | cut -c1-600
