set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_r05_j; mkdir -p $OUT; cd $R
python -m pytest tests/test_host_block_default_path.py tests/test_host_block_gpu.py -m gpu -q > $OUT/pytest_host.log 2>&1; tail -2 $OUT/pytest_host.log
timeout 600 python scripts/gpu_text_parity.py 60 1000 > $OUT/text_parity_60.txt 2>&1; tail -1 $OUT/text_parity_60.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --no-cpu --no-host-fed --no-ab > $OUT/bench.json 2> $OUT/err.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("headline", d["value"], d["ms_per_step"]); print("block", json.dumps(d["block_config"])[:1200])
PY
tail -c 200 $OUT/err.txt
