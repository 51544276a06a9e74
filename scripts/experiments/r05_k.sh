set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_k; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu --no-host-fed > $OUT/bench.json 2> $OUT/err.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "verify", d["verify"]); b=d["block_config"]; print("block", b["value"], b["ms_per_step"], b["without_exact_payload"], b["verify"])
PY
python $R/bench.py --workload c8 --steps 100 --no-cpu --no-host-fed > $OUT/c8_bench.json 2>> $OUT/err.txt
python - $OUT/c8_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("c8", d["value"], d["ms_per_step"], "verify", d["verify"]); b=d["block_config"]; print("c8 block", b["value"], b["ms_per_step"], b["without_exact_payload"], b["verify"])
PY
tail -c 300 $OUT/err.txt
