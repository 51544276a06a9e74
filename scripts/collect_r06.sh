#!/bin/bash
# Round-6 evidence in one GPU call, on the build that is committed:  bash scripts/collect_r06.sh r06_m [tests|notests] [full|quick]
# -> gpurun_out/prof_<tag>/: pytest / smoke logs, the bench line (headline = the block's configuration; classic_only, c8, A/B without
#    exact rows, host-fed leg, cpu legs), kernel stats + timeline, HBM traffic (PMC, stamped with the build id of libbtgpu.so), SQ /
#    stall / MFMA counters of the headline, the matrix-pipe ubenches (exact_mfma, mfma_rate, mfma_overlap, mfma_shadow), and (full)
#    the two-rank dry run, the one-rank RCCL gather, the device fuzz slices
set -u
TAG=${1:-r06_x}
MODE=${3:-full}    # full | quick (no fuzz, no N > 1 runs) | pmc (kernel stats + PMC passes + the bench line with traffic only)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd $R
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
fi
if [ "$MODE" != "pmc" ]; then for u in exact_mfma mfma_rate mfma_overlap mfma_shadow; do [ -x scripts/ubench/$u ] && ./scripts/ubench/$u > "$OUT/ubench_$u.txt" 2>&1; done; fi
cd /tmp && export TMPDIR=/tmp
[ "$MODE" = "pmc" ] || python $R/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
KRE='_kernel'   # counters only on the library's kernels (all named *_kernel)
n=c79; slots=2304
rm -rf /tmp/kt_$n /tmp/p1_$n /tmp/p2_$n /tmp/p3_$n /tmp/p4_$n /tmp/p5_$n
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -o kt -- python $R/bench.py --no-cpu --no-block-config --no-c8 --no-exact-all --no-ab --no-host-fed --no-timing > "$OUT/${n}_bench_under_rocprof.json" 2>> "$OUT/bench.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_$n -name '*kernel_stats.csv' | head -1)" "$OUT/${n}_kernel_stats.csv"
python $R/scripts/timeline.py "$(find /tmp/kt_$n -name '*kernel_trace.csv' | head -1)" 30 > "$OUT/${n}_timeline.txt" 2>&1
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --no-c8 --no-exact-all --no-ab --no-host-fed --sync"
rocprofv3 --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d /tmp/p1_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d /tmp/p2_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1_$n -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2_$n -name '*counter_collection.csv' | head -1)" $slots > "$OUT/${n}_pmc_hbm.json"
rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p3_$n -name '*counter_collection.csv' | head -1)" > "$OUT/${n}_pmc_sq.txt" 2>> "$OUT/bench.err"
rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p4_$n -name '*counter_collection.csv' | head -1)" > "$OUT/${n}_pmc_stall.txt" 2>> "$OUT/bench.err"
rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/p5_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p5_$n -name '*counter_collection.csv' | head -1)" > "$OUT/${n}_pmc_mfma.txt" 2>> "$OUT/bench.err"
# the bench line again with the traffic of this very build
python $R/bench.py --no-cpu --no-block-config --no-c8 --no-exact-all --no-ab --no-host-fed --pmc-json "$OUT/c79_pmc_hbm.json" > "$OUT/bench_with_traffic.json" 2>> "$OUT/bench.err"
cd $R
if [ "$MODE" = "full" ]; then
  python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu --no-c8 > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
  python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config --no-c8 --no-ab --no-host-fed 2> "$OUT/one_rank_rccl.err" | head -1 > "$OUT/one_rank_rccl_gather_bench.json"
  timeout 1500 python scripts/gpu_fuzz_adversarial.py 1000 31001 > "$OUT/fuzz_adversarial_1000_seed31001.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_1000_seed31001.txt"
  timeout 900 python scripts/gpu_fuzz_adversarial.py 400 31002 --wide --rates 4,10,16,40,50,100 > "$OUT/fuzz_adversarial_wide_400_seed31002.txt" 2>&1; tail -1 "$OUT/fuzz_adversarial_wide_400_seed31002.txt"
  timeout 1200 python scripts/gpu_judge_seamless.py 1500 32001 > "$OUT/judge_seamless_mix_1500_seed32001.txt" 2>&1; tail -1 "$OUT/judge_seamless_mix_1500_seed32001.txt"
  timeout 900 python scripts/gpu_judge_seamless.py 150 32002 --mode 100 > "$OUT/judge_seamless_100M_150_seed32002.txt" 2>&1; tail -1 "$OUT/judge_seamless_100M_150_seed32002.txt"
  timeout 600 python scripts/gpu_judge_seamless.py 600 32003 --weak > "$OUT/judge_weak_beside_600_seed32003.txt" 2>&1; tail -1 "$OUT/judge_weak_beside_600_seed32003.txt"
  timeout 900 python scripts/gpu_fuzz_parity.py 200 > "$OUT/fuzz_parity_200.txt" 2>&1; tail -1 "$OUT/fuzz_parity_200.txt"
  timeout 1200 python scripts/gpu_text_parity.py 60 2000 > "$OUT/text_parity_60.txt" 2>&1; tail -2 "$OUT/text_parity_60.txt"
fi
echo "== $n"; head -12 "$OUT/${n}_kernel_stats.csv"; cat "$OUT/${n}_pmc_sq.txt" "$OUT/${n}_pmc_stall.txt" "$OUT/${n}_pmc_mfma.txt" | head -40
python -c "import json; d=json.load(open('$OUT/${n}_pmc_hbm.json')); [print(k, {a: round(b/1e6,1) for a,b in v.items()}) for k,v in d['kernels'].items()]"
for f in bench bench_with_traffic two_rank_on_one_device_bench one_rank_rccl_gather_bench; do [ -f "$OUT/$f.json" ] && { echo "== $f"; tail -1 "$OUT/$f.json" | cut -c1-330; }; done
[ -f "$OUT/bench.json" ] && python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
for k in ("roofline", "roofline_bank", "verify", "host_fed", "classic_only", "c8", "cpu_baseline"): print(k, d.get(k))
print("parity", {k:v for k,v in d["parity"].items() if k!="note"})
PY
tail -c 400 "$OUT/bench.err"
