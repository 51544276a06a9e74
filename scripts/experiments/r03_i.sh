#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_i}
mkdir -p "$OUT"
cd $R
for v in run256 run256a legacy run256; do echo -n "$v: "; BTGPU_BANK=$v python scripts/bank_time.py 2304 40 2>&1 | tail -1; done | tee "$OUT/bank_times.txt"
BTGPU_BANK=run256 python scripts/pfb_phases.py 2304 3 > "$OUT/phases_run256.txt" 2>&1; tail -8 "$OUT/phases_run256.txt"
python bench.py --no-cpu --no-block-config 2>/dev/null | grep "^{" | cut -c1-130
python -m pytest tests -m gpu -x -q -k "c79 or fast_path or smoke or full_size" 2>&1 | tail -2
