#!/bin/bash
# Round 3: HBM traffic of the bank variants (PMC), bank time, the gather path's cost after the rework, remaining new tests
set -u
TAG=${1:-r03_f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
python -m pytest tests -m gpu -x -q -k "no_nsym or fuzz_case or rccl or two_ranks or small_rates or hopper or btrx" > "$OUT/pytest_gpu_subset.log" 2>&1; tail -3 "$OUT/pytest_gpu_subset.log"
for v in run256 legacy; do echo -n "$v: "; BTGPU_BANK=$v python scripts/bank_time.py 2304 40 2>&1 | tail -1; done | tee "$OUT/bank_times.txt"
python bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config 2> "$OUT/rccl.err" | grep "^{" > "$OUT/one_rank_rccl_gather_bench.json"; cut -c1-120 "$OUT/one_rank_rccl_gather_bench.json"
python bench.py --no-cpu --no-block-config 2> /dev/null | grep "^{" | cut -c1-120
cd /tmp && export TMPDIR=/tmp
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --sync"
for v in run256 legacy; do
rm -rf /tmp/p1 /tmp/p2
BTGPU_BANK=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o p -- $PM > /dev/null 2>> "$OUT/pmc.err"
BTGPU_BANK=$v rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o p -- $PM > /dev/null 2>> "$OUT/pmc.err"
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" 2304 > "$OUT/pmc_hbm_$v.json"
python -c "import json; d=json.load(open('$OUT/pmc_hbm_$v.json')); [print('$v', k, {a: round(b/1e6,1) for a,b in v.items()}) for k,v in d['kernels'].items() if k.startswith('pfb')]"
done
