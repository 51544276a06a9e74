#!/bin/bash
# build with the opt-in fused C8 stage 1: (1) the headline's PMC summary / bench line re-stamped with this build's id, (2) 8 Msps GPU tests
# with BTGPU_C8_FUSE=1, (3) C8 bench with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_r04_w; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --no-ab --no-host-fed --sync"
rocprofv3 --kernel-include-regex _kernel --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o p -- $PM > /dev/null 2> $O/err
rocprofv3 --kernel-include-regex _kernel --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o p -- $PM > /dev/null 2>> $O/err
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" 2304 > $O/c79_pmc_hbm.json
python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --pmc-json $O/c79_pmc_hbm.json > $O/bench_with_traffic.json 2>> $O/err
echo "t=$SECONDS after restamp"
cd $R
BTGPU_C8_FUSE=1 timeout 70 python -m pytest tests -m gpu -x -q -k "small_rates or staged_squelch or le_pass or hit_symbols or async_pipeline or squelch_threshold or exact_stage or process_device_with_halo or ragged" > $O/pytest_c8_fuse.log 2>&1; tail -2 $O/pytest_c8_fuse.log
echo "t=$SECONDS after tests"
C8="python bench.py --workload c8 --steps 100 --no-cpu --no-block-config --no-ab --no-host-fed --full-timing"
BTGPU_C8_FUSE=1 $C8 > $O/c8_fused.json 2>> $O/err
echo "t=$SECONDS"
if [ $SECONDS -lt 185 ]; then $C8 > $O/c8_apart.json 2>> $O/err; fi
python - <<PY
import json,os
for f in ['bench_with_traffic','c8_fused','c8_apart']:
    p='$O/%s.json'%f
    if not os.path.exists(p) or not os.path.getsize(p): continue
    d=json.loads(open(p).read().strip().split("\n")[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('traffic'), d['roofline'].get('build_id'), d['roofline'].get('kernel_avg_ms'))
PY
