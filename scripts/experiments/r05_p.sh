set -u
# the round's last 4 GPU-minutes: the wide generator (stretched companion ranges, carriers and white bursts) and the further rates on the device
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_p; mkdir -p $OUT; cd $R
timeout 215 python scripts/gpu_fuzz_adversarial.py 100000 33001 --rates 8,20,10,16,40,50 --wide --seconds 150 > $OUT/gpu_fuzz_adversarial_wide_seed33001.txt 2>&1; tail -1 $OUT/gpu_fuzz_adversarial_wide_seed33001.txt | cut -c1-700
