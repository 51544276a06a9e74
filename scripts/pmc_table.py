"""Print a per-kernel table of rocprofv3 --pmc counter_collection.csv files (btgpu kernels only)."""
import collections, csv, sys
agg = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "btgpu" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("btgpu::", "")[:34]
            agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in agg.items():
    print(k, " ".join("%s=%.3gM" % (a, b / 1e6) for a, b in sorted(v.items())))
