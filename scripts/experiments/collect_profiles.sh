#!/bin/bash
# Collect the per-round evidence on the GPU box:  bash scripts/collect_profiles.sh r01_e
# Writes gpurun_out/prof_<tag>/{bench.json, bench_under_rocprof.json, kernel_stats.csv, pmc_hbm.json}
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SLOTS=${SLOTS:-2304}
python $R/bench.py --slots $SLOTS > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/kt /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --slots $SLOTS --no-cpu > "$OUT/bench_under_rocprof.json" 2>> "$OUT/bench.err"
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# counters: their own passes, no tracing flags
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o p -- python $R/bench.py --slots $SLOTS --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o p -- python $R/bench.py --slots $SLOTS --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" $SLOTS > "$OUT/pmc_hbm.json"
# instruction mix of the kernels (own pass; SQ counters)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3 -o p -- python $R/bench.py --slots $SLOTS --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p3 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_sq.txt" 2>> "$OUT/bench.err"
# where the waves' time goes (own pass): busy / waiting / issuing, LDS share
rm -rf /tmp/p4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4 -o p -- python $R/bench.py --slots $SLOTS --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/bench.err"
python $R/scripts/pmc_table.py "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_stall.txt" 2>> "$OUT/bench.err"
# the bench line again with the measured HBM traffic of this very build embedded
python $R/bench.py --slots $SLOTS --no-cpu --pmc-json "$OUT/pmc_hbm.json" > "$OUT/bench_with_traffic.json" 2>> "$OUT/bench.err"
cat "$OUT/pmc_sq.txt" "$OUT/pmc_stall.txt"
tail -c 600 "$OUT/bench.err"
cat "$OUT/kernel_stats.csv"
python -c "import json; d=json.load(open('$OUT/pmc_hbm.json')); [print(k, {a: round(b/1e6,1) for a,b in v.items()}) for k,v in d['kernels'].items()]"
tail -1 "$OUT/bench.json" | cut -c1-400
