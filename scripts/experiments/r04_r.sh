#!/bin/bash
# exact stage's DDC with the tile epilogue spread over the next two tiles: exact-stage tests, headline twice, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_r; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x -k "exact_stage or fuzz_case or randomised" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python bench.py --no-cpu --no-host-fed --no-ab --no-block-config"
for i in 1 2; do $B > $O/head$i.json 2>> $O/err; python -c "
import json; d=json.loads(open('$O/head$i.json').read().strip().split('\n')[-1]); print('head', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --no-timing --sync > /dev/null 2>> $O/err
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $O/sync_kernel_stats.csv; head -9 $O/sync_kernel_stats.csv
