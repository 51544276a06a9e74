#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_g; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { # name verify flags...
  n=$1; v=$2; shift; shift
  rm -rf /tmp/kt_$n; BTGPU_VERIFY=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -o kt -- python $R/bench.py --no-cpu --no-block-config "$@" > $OUT/bench_$n.json 2>> $OUT/err
  python $R/scripts/summarize_rocprof.py "$(find /tmp/kt_$n -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$n.csv; echo "== $n (verify $v $@)"; grep -E "window_kernel|finish|verify_ddc|pfb100f" $OUT/kernel_stats_$n.csv
  python -c "import json,sys; d=json.loads(open('$OUT/bench_$n.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
}
run le_v1 1 --le
run hdr_v1 1 --headers
run lehdr_v2 2 --le --headers
run lehdr_v1_sync 1 --le --headers --sync
