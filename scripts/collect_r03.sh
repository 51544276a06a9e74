#!/bin/bash
# Round-3 evidence in one GPU call, on the build that is committed:  bash scripts/collect_r03.sh r03_x [tests]
# -> gpurun_out/prof_<tag>/: pytest / smoke logs, bench line (with cpu legs), kernel stats, HBM traffic (PMC, stamped with
#    the build id of libbtgpu.so: copy pmc_hbm.json to profiles/<tag>_pmc_hbm.json and bench.py picks it up by itself),
#    SQ / stall counters, kernel timeline without timing events, C8 line + stats, two-rank dry run
set -u
TAG=${1:-r03_x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd $R
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
  cp gpurun_out/cfo_curve_*.json "$OUT/" 2>/dev/null
fi
cd /tmp && export TMPDIR=/tmp
SLOTS=2304
python $R/bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/kt /tmp/kt2 /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/kt8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config > "$OUT/bench_under_rocprof.json" 2>> "$OUT/bench.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# the production path records no timing events: kernel timeline of that configuration (gaps between kernels)
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python $R/bench.py --no-cpu --no-block-config --no-timing > "$OUT/bench_no_timing_under_rocprof.json" 2>> "$OUT/bench.err"
python $R/scripts/timeline.py "$(find /tmp/kt2 -name '*kernel_trace.csv' | head -1)" 30 > "$OUT/timeline_no_timing.txt" 2>&1
# counters: their own passes, no tracing flags
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --sync"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" $SLOTS > "$OUT/pmc_hbm.json"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3 -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p3 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_sq.txt" 2>> "$OUT/bench.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4 -o p -- $PM > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_stall.txt" 2>> "$OUT/bench.err"
# the bench line again: picks the PMC summary up when it sits under profiles/ (here: passed explicitly)
python $R/bench.py --no-cpu --no-block-config --pmc-json "$OUT/pmc_hbm.json" > "$OUT/bench_with_traffic.json" 2>> "$OUT/bench.err"
# C8 (BASELINE configs[1])
python $R/bench.py --workload c8 --steps 100 > "$OUT/c8_bench.json" 2> "$OUT/c8.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o kt -- python $R/bench.py --workload c8 --steps 100 --no-cpu --no-block-config > "$OUT/c8_bench_under_rocprof.json" 2>> "$OUT/c8.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt8 -name '*kernel_stats.csv' | head -1)" "$OUT/c8_kernel_stats.csv"
# N > 1 path on this one device: two ranks, time-partitioned, gathered once per batch (gloo) and the single-rank RCCL group
cd $R
python bench.py --gpus 2 --all-on-device0 --backend gloo --slots 1152 --no-cpu > "$OUT/two_rank_on_one_device_bench.json" 2> "$OUT/two_rank.err"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config > "$OUT/one_rank_rccl_gather_bench.json" 2> "$OUT/one_rank_rccl.err"
cat "$OUT/pmc_sq.txt" "$OUT/pmc_stall.txt"; head -8 "$OUT/kernel_stats.csv"; cat "$OUT/timeline_no_timing.txt" | tail -12
python -c "import json; d=json.load(open('$OUT/pmc_hbm.json')); [print(k, {a: round(b/1e6,1) for a,b in v.items()}) for k,v in d['kernels'].items()]"
for f in bench bench_with_traffic c8_bench two_rank_on_one_device_bench one_rank_rccl_gather_bench; do echo "== $f"; tail -1 "$OUT/$f.json" | cut -c1-330; done
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("roofline", d["roofline"]); print("block_config", d["block_config"]); print("parity", {k:v for k,v in d["parity"].items() if k!="note"})
PY
tail -c 400 "$OUT/bench.err"
