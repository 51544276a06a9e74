#!/bin/bash
# wave priority per phase of the bank kernel (exploration build: BTGPU_BANK=run256a reads the four priorities from BTGPU_PFB_DBG)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_n}
mkdir -p "$OUT"
cd $R
enc() { echo $(( ($1 << 8) | ($2 << 12) | ($3 << 16) | ($4 << 20) )); }
for combo in "2 0 3 1" "3 0 1 2" "3 0 2 1" "2 0 3 2" "3 0 3 1" "3 0 3 3" "2 1 3 0" "3 0 3 2" "0 0 3 1" "2 0 3 1"; do
  set -- $combo
  echo -n "stage $1 march $2 dft $3 epilogue $4: "
  BTGPU_BANK=run256a BTGPU_PFB_DBG=$(enc $1 $2 $3 $4) python scripts/bank_time.py 2304 30 2>&1 | tail -1
done | tee "$OUT/prio_sweep.txt"
