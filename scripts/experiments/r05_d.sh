set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_d; mkdir -p $OUT; cd $R
for v in run512 run512r; do
  BTGPU_BANK=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast_path_c79 or full_size_fast_path or time_partition" > $OUT/pytest_$v.log 2>&1; tail -2 $OUT/pytest_$v.log
done
cd /tmp && export TMPDIR=/tmp
for v in run256 run512 run512r run256 run512 run512r; do
  BTGPU_BANK=$v python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed > $OUT/bench_$v.json 2>> $OUT/err.txt
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], "bank ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], d["parity"]["records_sha256"])
PY
done
tail -c 300 $OUT/err.txt
