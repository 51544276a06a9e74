set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_b; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu --no-host-fed > $OUT/bench_prescan1.json 2> $OUT/err1.txt
BTGPU_PRESCAN=0 python $R/bench.py --no-cpu --no-host-fed > $OUT/bench_prescan0.json 2> $OUT/err0.txt
python $R/bench.py --no-cpu --no-host-fed > $OUT/bench_prescan1_again.json 2>> $OUT/err1.txt
for f in bench_prescan1 bench_prescan0 bench_prescan1_again; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "verify", d["verify"]["cost_frac_of_step"], d["verify"]["ab_no_verify"]["value"], "block", d["block_config"]["value"], d["block_config"]["ms_per_step"], d["block_config"]["kernel_avg_ms"], "parity", d["parity"]["differential"]["planted_identical"], d["parity"]["records_sha256"], d["block_config"]["differential"]["planted_identical"])
PY
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --no-timing > $OUT/c79_bench_under_rocprof.json 2>> $OUT/err1.txt
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/c79_kernel_stats.csv; head -12 $OUT/c79_kernel_stats.csv
python $R/scripts/timeline.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" 40 > $OUT/c79_timeline.txt 2>&1; tail -24 $OUT/c79_timeline.txt
tail -c 300 $OUT/err1.txt
