#!/bin/bash
# occupancy sweep of the bank kernel: 3 / 2 / 1 tiles per CU (extra dynamic LDS), time and per-phase cycles
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_l}
mkdir -p "$OUT"
cd $R
for pad in 0 60000; do echo "== pad $pad"; BTGPU_BANK_LDS_PAD=$pad python scripts/pfb_phases.py 2304 2 2>&1 | tail -11; done | tee "$OUT/occupancy_sweep.txt"
