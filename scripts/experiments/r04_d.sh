#!/bin/bash
# exact stage: GPU tests, default bench line (headline + block configuration), randomised differentials of both seeds
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_d; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
BTGPU_VERIFY=0 timeout 300 python bench.py --no-cpu > $OUT/bench_verify0.json 2>> $OUT/bench.err
timeout 900 python scripts/gpu_fuzz_fast.py 400 32 > $OUT/fuzz_fast_400_seed32.txt 2>&1; tail -1 $OUT/fuzz_fast_400_seed32.txt
timeout 1500 python scripts/gpu_fuzz_fast.py 800 77 > $OUT/fuzz_fast_800_seed77.txt 2>&1; tail -1 $OUT/fuzz_fast_800_seed77.txt
for f in $OUT/bench.json $OUT/bench_verify0.json; do echo "== $f"; python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_ms"), d.get("verify"))
if d.get("block_config"): print("  block", {k: d["block_config"][k] for k in ("value","ms_per_step","kernel_avg_ms","verify") if k in d["block_config"]})
if d.get("parity") and "differential" in d["parity"]: print("  parity", d["parity"]["differential"])
PY
done
