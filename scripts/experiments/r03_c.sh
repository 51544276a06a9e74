#!/bin/bash
# Round 3: where the run kernel's time goes -- per-wave phase marks, SQ counters (run256 vs legacy)
set -u
TAG=${1:-r03_c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
BTGPU_BANK=run256 python scripts/pfb_phases.py 2304 3 > "$OUT/phases_run256.txt" 2>&1; cat "$OUT/phases_run256.txt" | tail -9
BTGPU_BANK=legacy python scripts/pfb_phases.py 2304 3 > "$OUT/phases_legacy.txt" 2>&1; tail -9 "$OUT/phases_legacy.txt"
cd /tmp && export TMPDIR=/tmp
for v in run256 legacy; do
rm -rf /tmp/p3 /tmp/p4
BTGPU_BANK=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p3 -o p -- python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/pmc.err"
python $R/scripts/pmc_table.py "$(find /tmp/p3 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_sq_$v.txt" 2>> "$OUT/pmc.err"
BTGPU_BANK=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p4 -o p -- python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --sync > /dev/null 2>> "$OUT/pmc.err"
python $R/scripts/pmc_table.py "$(find /tmp/p4 -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_stall_$v.txt" 2>> "$OUT/pmc.err"
echo "== $v"; grep pfb "$OUT/pmc_sq_$v.txt" "$OUT/pmc_stall_$v.txt"
done
