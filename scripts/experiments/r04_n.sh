#!/bin/bash
# LE rule (every LE-searched window handed to the exact stage): GPU tests, headline + block configuration
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_n; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python bench.py --no-cpu --no-host-fed > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-200
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("verify", d["verify"]); print("block_config", d["block_config"]); print("kernels", d.get("kernels_ms"))
PY
python bench.py --no-cpu --no-host-fed --no-ab --no-block-config --le --headers --steps 30 > $O/block.json 2>> $O/bench.err
python - $O/block.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("block value", d["value"], d["ms_per_step"]); print("verify", d["verify"]); print({k:v for k,v in d.items() if 'kernel' in k or 'timing' in k})
PY
