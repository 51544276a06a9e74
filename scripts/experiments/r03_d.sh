#!/bin/bash
# Round 3: A/B of the run-kernel experiment variants (bank kernel time only, scripts/bank_time.py) + phases of the best
set -u
TAG=${1:-r03_d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
for v in run256 run256b run256c run256d run256e legacy run256c run256d; do
  echo -n "$v: "; BTGPU_BANK=$v python scripts/bank_time.py 2304 40 2>&1 | tail -1
done | tee "$OUT/bank_times.txt"
for v in run256c run256d; do BTGPU_BANK=$v python scripts/pfb_phases.py 2304 3 > "$OUT/phases_$v.txt" 2>&1; tail -8 "$OUT/phases_$v.txt"; done
