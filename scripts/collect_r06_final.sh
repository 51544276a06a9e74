#!/bin/bash
# The round's last collection on the committed build:  bash scripts/collect_r06_final.sh r06_x
# = collect_r06.sh <tag> tests quick (GPU tests, smoke, ubenches, the bench line, kernel stats + timeline, the PMC passes, the bench line
# with this build's traffic) + the N > 1 dry runs + C8 (configs[1]) with every kernel bracketed and its kernel trace + text parity +
# time-boxed device fuzz (every script prints its TOTAL line when its --seconds are up)
set -u
TAG=${1:-r06_x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
bash $R/scripts/collect_r06.sh $TAG tests quick
cd $R
python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu --no-c8 --no-exact-all > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config --no-c8 --no-ab --no-host-fed --no-exact-all 2> "$OUT/one_rank_rccl.err" | head -1 > "$OUT/one_rank_rccl_gather_bench.json"
cd /tmp && export TMPDIR=/tmp
BENCH_DUMP_STEPS=1 python $R/bench.py --workload c8 --no-cpu --no-block-config --no-exact-all --no-ab --no-host-fed --full-timing --steps 40 > "$OUT/c8_full_timing.json" 2> "$OUT/c8.err"
rm -rf /tmp/kt_c8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c8 -o kt -- python $R/bench.py --workload c8 --no-cpu --no-block-config --no-exact-all --no-ab --no-host-fed --no-timing --steps 40 > "$OUT/c8_bench_under_rocprof.json" 2>> "$OUT/c8.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_c8 -name '*kernel_stats.csv' | head -1)" "$OUT/c8_kernel_stats.csv"
python $R/scripts/timeline.py "$(find /tmp/kt_c8 -name '*kernel_trace.csv' | head -1)" 40 > "$OUT/c8_timeline.txt" 2>&1
cd $R
timeout 900 python scripts/gpu_text_parity.py 60 2000 > "$OUT/text_parity_60.txt" 2>&1; tail -2 "$OUT/text_parity_60.txt"
timeout 600 python scripts/gpu_fuzz_parity.py 200 > "$OUT/fuzz_parity_200.txt" 2>&1; tail -1 "$OUT/fuzz_parity_200.txt"
timeout 800 python scripts/gpu_judge_seamless.py 100000 36001 --seconds 600 > "$OUT/judge_seamless_mix_600s_seed36001.txt" 2>&1; tail -1 "$OUT/judge_seamless_mix_600s_seed36001.txt" | cut -c1-300
timeout 700 python scripts/gpu_fuzz_adversarial.py 100000 36002 --seconds 500 > "$OUT/fuzz_adversarial_500s_seed36002.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_500s_seed36002.txt" | cut -c1-300
grep -h step_ms "$OUT/c8.err" | cut -c1-300
for f in two_rank_on_one_device_bench one_rank_rccl_gather_bench c8_full_timing; do [ -f "$OUT/$f.json" ] && { echo "== $f"; tail -1 "$OUT/$f.json" | cut -c1-330; }; done
