#!/bin/bash
# The round's last collection, on the build with the stream pool and the rotating A operand:  bash scripts/collect_r06_zz.sh [tag]
# = collect_r06.sh <tag> tests quick (GPU tests, smoke, ubenches, the bench line, kernel stats + timeline, the PMC passes, the bench line
# with this build's traffic) + handle after handle in one process (pool on / off) + the N > 1 dry runs + C8 with every kernel bracketed
# and its kernel trace + time-boxed device fuzz at 100 Msps (the rate whose exact_rows_kernel changed) and over the committed generators
set -u
TAG=${1:-r06_zz}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
bash $R/scripts/collect_r06.sh $TAG tests quick
cd $R
python scripts/device_code_ids.py gr-bluetooth_amd/libbtgpu.so "$OUT/c79_pmc_hbm.json" > /dev/null 2>> "$OUT/bench.err"     # stamp the per-kernel device-code ids
{ echo "== pool (default)"; python scripts/experiments/second_handle.py c8 c8 c79 c79 c8; echo "== BTGPU_STREAM_POOL=0"; BTGPU_STREAM_POOL=0 python scripts/experiments/second_handle.py c8 c8 c79 c79 c8; } 2>&1 | grep -v amdgpu.ids > "$OUT/second_handle.txt"
cat "$OUT/second_handle.txt"
# the exact rows' kernel alone, the rotating A operand (the library's form) against three 16-byte groups ahead (-DBTGPU_EX_FULLA=0), 4 / 9 / 20 channels per tile
for u in exact_mfma exact_mfma_e0; do [ -x scripts/ubench/$u ] && for pt in 4 9 20; do echo "== $u, $pt channels per tile"; ./scripts/ubench/$u $pt 2304 0 2048 | grep "^mode"; done; done > "$OUT/ubench_exact_full_a_ab.txt" 2>&1
cat "$OUT/ubench_exact_full_a_ab.txt"
python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu --no-c8 --no-exact-all > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config --no-c8 --no-ab --no-host-fed --no-exact-all 2> "$OUT/one_rank_rccl.err" | head -1 > "$OUT/one_rank_rccl_gather_bench.json"
cd /tmp && export TMPDIR=/tmp
BENCH_DUMP_STEPS=1 python $R/bench.py --workload c8 --no-cpu --no-block-config --no-exact-all --no-ab --no-host-fed --full-timing --steps 40 > "$OUT/c8_full_timing.json" 2> "$OUT/c8.err"
rm -rf /tmp/kt_c8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c8 -o kt -- python $R/bench.py --workload c8 --no-cpu --no-block-config --no-exact-all --no-ab --no-host-fed --no-timing --steps 40 > "$OUT/c8_bench_under_rocprof.json" 2>> "$OUT/c8.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_c8 -name '*kernel_stats.csv' | head -1)" "$OUT/c8_kernel_stats.csv"
cd $R
timeout 500 python scripts/gpu_judge_seamless.py 100000 37002 --mode 100 --seconds 360 > "$OUT/judge_seamless_100M_360s_seed37002.txt" 2>&1; tail -1 "$OUT/judge_seamless_100M_360s_seed37002.txt" | cut -c1-300
timeout 400 python scripts/gpu_judge_seamless.py 100000 37001 --seconds 240 > "$OUT/judge_seamless_mix_240s_seed37001.txt" 2>&1; tail -1 "$OUT/judge_seamless_mix_240s_seed37001.txt" | cut -c1-300
timeout 400 python scripts/gpu_fuzz_adversarial.py 100000 37003 --seconds 240 --wide --rates 4,10,16,40,50,100 > "$OUT/fuzz_adversarial_wide_240s_seed37003.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_wide_240s_seed37003.txt" | cut -c1-300
timeout 600 python scripts/gpu_text_parity.py 30 2000 > "$OUT/text_parity_30.txt" 2>&1; tail -2 "$OUT/text_parity_30.txt"
for f in two_rank_on_one_device_bench one_rank_rccl_gather_bench c8_full_timing; do [ -f "$OUT/$f.json" ] && { echo "== $f"; tail -1 "$OUT/$f.json" | cut -c1-330; }; done
