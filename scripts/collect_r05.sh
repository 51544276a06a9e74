#!/bin/bash
# Round-5 evidence in one GPU call, on the build that is committed:  bash scripts/collect_r05.sh r05_a [tests|notests] [fuzz|nofuzz] [full|quick]
# (quick: GPU suite, bench line, kernel stats + PMC of the headline only; c8only: kernel stats + PMC + bench line of the C8 workload)
# -> gpurun_out/prof_<tag>/: pytest / smoke logs, bench line (exact stage on, A/B without it, block configuration, host-fed leg,
#    cpu legs), kernel stats + timeline, HBM traffic (PMC, stamped with the build id of libbtgpu.so), SQ / stall counters --
#    for the headline, the block configuration (LE | HEADERS) and C8 --, two-rank dry run, one-rank RCCL gather, and (fuzz) the
#    randomised differentials of the polyphase path (seeds 32 and 77) and of the direct path
set -u
TAG=${1:-r05_x}
MODE=${4:-full}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd $R
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
  cp gpurun_out/cfo_curve_*.json "$OUT/" 2>/dev/null
fi
cd /tmp && export TMPDIR=/tmp
[ "$MODE" = "c8only" ] || python $R/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
KRE='_kernel'   # counters only on the library's kernels (all named *_kernel): rocprofv3 crashed inside torch's randn / mul_ launches of the C8 capture generator
pmc() { # name slots bench-args...   -> $OUT/<name>_pmc_hbm.json, _pmc_sq.txt, _pmc_stall.txt, _kernel_stats.csv, _timeline.txt
  n=$1; slots=$2; shift; shift
  rm -rf /tmp/kt_$n /tmp/p1_$n /tmp/p2_$n /tmp/p3_$n /tmp/p4_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -o kt -- python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --no-timing "$@" > "$OUT/${n}_bench_under_rocprof.json" 2>> "$OUT/bench.err"
  python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_$n -name '*kernel_stats.csv' | head -1)" "$OUT/${n}_kernel_stats.csv"
  python $R/scripts/timeline.py "$(find /tmp/kt_$n -name '*kernel_trace.csv' | head -1)" 30 > "$OUT/${n}_timeline.txt" 2>&1
  PM="python $R/bench.py $* --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --no-ab --no-host-fed --sync"     # (the one step wins over a --steps of the caller)
  rocprofv3 --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d /tmp/p1_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
  rocprofv3 --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d /tmp/p2_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
  python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1_$n -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2_$n -name '*counter_collection.csv' | head -1)" $slots > "$OUT/${n}_pmc_hbm.json"
  rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
  python $R/scripts/pmc_table.py "$(find /tmp/p3_$n -name '*counter_collection.csv' | head -1)" > "$OUT/${n}_pmc_sq.txt" 2>> "$OUT/bench.err"
  rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4_$n -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
  python $R/scripts/pmc_table.py "$(find /tmp/p4_$n -name '*counter_collection.csv' | head -1)" > "$OUT/${n}_pmc_stall.txt" 2>> "$OUT/bench.err"
}
[ "$MODE" = "c8only" ] || pmc c79 2304
if [ "$MODE" = "c8only" ]; then
pmc c8 16384 --workload c8 --steps 100 --synth-device cpu
python $R/bench.py --workload c8 --steps 100 --no-cpu --no-block-config --no-ab --no-host-fed --pmc-json "$OUT/c8_pmc_hbm.json" > "$OUT/c8_bench_with_traffic.json" 2> "$OUT/c8.err"
cat "$OUT/c8_pmc_hbm.json" | head -c 1500; tail -1 "$OUT/c8_bench_with_traffic.json" | cut -c1-600; exit 0
fi
if [ "$MODE" = "full" ]; then
pmc block 2304 --le --headers --exact-payload
pmc c8 16384 --workload c8 --steps 100 --synth-device cpu      # (the PMC passes crash inside torch's randn launches of the GPU generator)
fi
# the bench line again: picks the PMC summary up when it sits under profiles/ (here: passed explicitly)
python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --pmc-json "$OUT/c79_pmc_hbm.json" > "$OUT/bench_with_traffic.json" 2>> "$OUT/bench.err"
if [ "$MODE" = "full" ]; then
C8J=""; python -c "import json,sys; json.load(open(sys.argv[1]))" "$OUT/c8_pmc_hbm.json" 2>/dev/null && C8J="--pmc-json $OUT/c8_pmc_hbm.json"
python $R/bench.py --workload c8 --steps 100 $C8J > "$OUT/c8_bench.json" 2> "$OUT/c8.err"
# N > 1 path on this one device: two ranks, time-partitioned, gathered (gloo), and the single-rank RCCL group
cd $R
python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config --no-ab --no-host-fed 2> "$OUT/one_rank_rccl.err" | head -1 > "$OUT/one_rank_rccl_gather_bench.json"
fi
cd $R
if [ "$MODE" = "full" ]; then
  timeout 1500 python scripts/gpu_text_parity.py 200 2000 > "$OUT/text_parity_200.txt" 2>&1; tail -2 "$OUT/text_parity_200.txt"
fi
if [ "${3:-fuzz}" = "fuzz" ]; then
  timeout 900 python scripts/gpu_fuzz_fast.py 400 32 > "$OUT/fuzz_fast_400_seed32.txt" 2>&1; tail -1 "$OUT/fuzz_fast_400_seed32.txt"
  timeout 1500 python scripts/gpu_fuzz_fast.py 800 77 > "$OUT/fuzz_fast_800_seed77.txt" 2>&1; tail -1 "$OUT/fuzz_fast_800_seed77.txt"
  timeout 900 python scripts/gpu_fuzz_parity.py 400 > "$OUT/fuzz_parity_400.txt" 2>&1; tail -1 "$OUT/fuzz_parity_400.txt"
fi
for n in c79 $([ "$MODE" = full ] && echo block c8); do echo "== $n"; head -9 "$OUT/${n}_kernel_stats.csv"; cat "$OUT/${n}_pmc_sq.txt" | head -12; python -c "import json; d=json.load(open('$OUT/${n}_pmc_hbm.json')); [print(k, {a: round(b/1e6,1) for a,b in v.items()}) for k,v in d['kernels'].items()]"; done
for f in bench bench_with_traffic c8_bench two_rank_on_one_device_bench one_rank_rccl_gather_bench; do [ -f "$OUT/$f.json" ] && { echo "== $f"; tail -1 "$OUT/$f.json" | cut -c1-330; }; done
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("roofline", d["roofline"]); print("verify", d["verify"]); print("host_fed", d["host_fed"]); print("block_config", d["block_config"]); print("parity", {k:v for k,v in d["parity"].items() if k!="note"})
PY
tail -c 400 "$OUT/bench.err"
