#!/bin/bash
# median of the bank-kernel time and of the throughput over N bench runs (GPU box):  bash scripts/ab_bench.sh [N]
N=${1:-5}
for i in $(seq $N); do python bench.py --no-cpu 2>/dev/null | tail -1; done | python -c "
import json, sys, statistics as st
rows = [json.loads(l) for l in sys.stdin if l.strip()]
k = [r['roofline']['kernel_avg_ms']['ddc_channel'] for r in rows]
v = [r['value'] for r in rows]
print('ddc_channel ms: median %.4f min %.4f max %.4f | value: median %.0f max %.0f' % (st.median(k), min(k), max(k), st.median(v), max(v)))"
