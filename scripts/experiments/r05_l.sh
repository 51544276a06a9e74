set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_l; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu --no-host-fed --no-block-config > $OUT/bench.json 2> $OUT/err.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "verify cost", d["verify"]["cost_frac_of_step"], "noverify", d["verify"]["ab_no_verify"]["value"])
PY
python $R/bench.py --workload c8 --steps 100 --no-cpu --no-host-fed --no-block-config > $OUT/c8_bench.json 2>> $OUT/err.txt
python - $OUT/c8_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("c8", d["value"], d["ms_per_step"], "verify cost", d["verify"]["cost_frac_of_step"], "noverify", d["verify"]["ab_no_verify"]["value"])
PY
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --no-cpu --no-block-config --no-ab --no-host-fed --no-timing > $OUT/c79_under_rocprof.json 2>> $OUT/err.txt
python $R/scripts/summarize_rocprof.py "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" $OUT/c79_kernel_stats.csv; head -12 $OUT/c79_kernel_stats.csv
tail -c 200 $OUT/err.txt
