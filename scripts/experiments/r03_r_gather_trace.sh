#!/bin/bash
# kernel + memory-copy trace of the forced single-rank RCCL gather: when do the round's copy-in, collective and copy-out run?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03_r}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gt -o gt -- python $R/bench.py --gpus 1 --force-gather --backend nccl --no-cpu --no-block-config > "$OUT/bench.json" 2> "$OUT/bench.err"
K=$(find /tmp/gt -name '*kernel_trace.csv' | head -1); M=$(find /tmp/gt -name '*memory_copy_trace.csv' | head -1)
python - "$K" "$M" > "$OUT/gather_trace.txt" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "pfb100f" in n: tag = "bank"
    elif "window_kernel" in n: tag = "window"
    elif "noise_stage2" in n: tag = "stage2"
    elif "finish" in n: tag = "finish"
    elif "nccl" in n.lower() or "rccl" in n.lower() or "AllGather" in n: tag = "RCCL:" + n[:40]
    else: continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
for r in csv.DictReader(open(sys.argv[2])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s bytes" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
rows.sort()
t_last_bank = [s for s, e, t in rows if t == "bank"]
t0 = t_last_bank[-22] if len(t_last_bank) > 22 else rows[0][0]        # the timed region: the last 20 + a few
for s, e, t in rows:
    if s >= t0 and (t.startswith("RCCL") or t.startswith("COPY") or t == "bank"):
        print("%10.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, t))
PY
tail -80 "$OUT/gather_trace.txt"
