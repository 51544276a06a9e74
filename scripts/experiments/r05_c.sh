set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_c; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python scripts/gpu_text_parity.py 60 1000 > $OUT/text_parity_60.txt 2>&1; tail -12 $OUT/text_parity_60.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu --no-host-fed > $OUT/bench.json 2> $OUT/err.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("headline", d["value"], d["ms_per_step"], "verify", d["verify"])
print("block", json.dumps(d["block_config"])[:1500])
PY
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-ab --no-host-fed --no-timing --le --headers --exact-payload > $OUT/block_bench_under_rocprof.json 2>> $OUT/err.txt
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/block_kernel_stats.csv; head -14 $OUT/block_kernel_stats.csv
python $R/scripts/timeline.py "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" 40 > $OUT/block_timeline.txt 2>&1
tail -c 300 $OUT/err.txt
