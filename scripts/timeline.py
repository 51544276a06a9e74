"""Print the btgpu kernel timeline (start offset, duration, gap to previous kernel end) from a
rocprofv3 kernel_trace.csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "btgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("btgpu::", "")[:26]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-26s start %9.1f us  dur %8.1f us  gap_from_prev_end %8.1f us  stream %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Stream_Id", "?")))
    prev_end = e
