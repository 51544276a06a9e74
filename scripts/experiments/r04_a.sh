#!/bin/bash
# first contact of the exact-confirmation stage on the MI355X: GPU tests, then the bench line with the stage on / hits only / off
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04_a; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py --no-block-config > $OUT/bench_verify1.json 2> $OUT/bench1.err; tail -c 600 $OUT/bench1.err
for v in 0 2 1; do BTGPU_VERIFY=$v timeout 300 python bench.py --no-cpu --no-block-config --full-timing > $OUT/bench_nocpu_verify$v.json 2>> $OUT/bench.err; done
BTGPU_VERIFY=1 timeout 300 python bench.py --no-cpu > $OUT/bench_block_verify1.json 2>> $OUT/bench.err
BTGPU_VERIFY=0 timeout 300 python bench.py --no-cpu > $OUT/bench_block_verify0.json 2>> $OUT/bench.err
timeout 600 python scripts/gpu_fuzz_fast.py 120 32 > $OUT/fuzz_fast_120_seed32.txt 2>&1; tail -1 $OUT/fuzz_fast_120_seed32.txt
for f in $OUT/bench_*.json; do echo "== $f"; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_ms"), d.get("verify"))
    if "block_config" in d: print("  block", {k: d["block_config"][k] for k in ("value","ms_per_step","kernel_avg_ms") if k in d["block_config"]})
    if "parity" in d and "differential" in d["parity"]: print("  parity", d["parity"]["differential"])
except Exception as e: print("ERR", e)
PY
done
