#!/bin/bash
# exact stage's DDC with its list entries fetched an item ahead; the same with the DDC on the tail stream (BTGPU_VERIFY_TAIL=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_o; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x -k "exact_stage or fuzz_case or randomised" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python bench.py --no-cpu --no-host-fed --no-ab --no-block-config"
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in (d.get("kernel_ms") or d.get("kernels_ms") or {}).items()} if isinstance(d.get("kernel_ms") or d.get("kernels_ms"), dict) else [k for k in d if 'ms' in k])
PY
}
$B > $O/head1.json 2>> $O/err; show $O/head1.json
BTGPU_VERIFY_TAIL=1 $B > $O/tail1.json 2>> $O/err; show $O/tail1.json
$B > $O/head2.json 2>> $O/err; show $O/head2.json
BTGPU_VERIFY_TAIL=1 $B > $O/tail2.json 2>> $O/err; show $O/tail2.json
BTGPU_VERIFY_TAIL=1 $B --le --headers > $O/tail_block.json 2>> $O/err; show $O/tail_block.json
$B --le --headers > $O/head_block.json 2>> $O/err; show $O/head_block.json
tail -3 $O/err
