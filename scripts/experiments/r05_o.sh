set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_o; mkdir -p $OUT; cd $R
timeout 560 python scripts/gpu_fuzz_adversarial.py 2200 32001 --rates 8,8,20 > $OUT/gpu_fuzz_adversarial_2200_seed32001.txt 2>&1; tail -1 $OUT/gpu_fuzz_adversarial_2200_seed32001.txt | cut -c1-600
timeout 400 python scripts/gpu_fuzz_adversarial.py 420 32002 --rates 100 > $OUT/gpu_fuzz_adversarial_420_seed32002_100M.txt 2>&1; tail -1 $OUT/gpu_fuzz_adversarial_420_seed32002_100M.txt | cut -c1-600
