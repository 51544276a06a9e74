#!/bin/bash
# wave-per-tile DDC of the exact stage at the small decimations: GPU tests, C8 kernel stats + bench, and the headline's PMC
# summary / bench line re-stamped with the new build id
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_r04_u; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c8 -o kt -- python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --no-timing --workload c8 --steps 100 > $O/c8_bench_under_rocprof.json 2> $O/err
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_c8 -name '*kernel_stats.csv' | head -1)" $O/c8_kernel_stats.csv; head -10 $O/c8_kernel_stats.csv
python $R/bench.py --workload c8 --steps 100 --no-cpu --no-block-config --no-host-fed > $O/c8_bench.json 2>> $O/err
PM="python $R/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu --no-block-config --no-ab --no-host-fed --sync"
rocprofv3 --kernel-include-regex _kernel --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o p -- $PM > /dev/null 2>> $O/err
rocprofv3 --kernel-include-regex _kernel --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o p -- $PM > /dev/null 2>> $O/err
python $R/scripts/pmc_hbm_json.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" 2304 > $O/c79_pmc_hbm.json
python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --pmc-json $O/c79_pmc_hbm.json > $O/bench_with_traffic.json 2>> $O/err
python - <<PY
import json
for f in ['c8_bench','bench_with_traffic']:
    d=json.loads(open('$O/%s.json'%f).read().strip().split("\n")[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('traffic'), d['roofline'].get('build_id'), d.get('verify'))
PY
tail -3 $O/err
