"""Per-kernel table of rocprofv3 --pmc counter_collection.csv files (btgpu kernels only): counter values
averaged per launch, in millions, with the number of launches seen."""
import collections, csv, sys
agg, disp = collections.defaultdict(dict), collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "btgpu" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("btgpu::", "")[:40]
            agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = max(len(disp[k]), 1)
    print(k, "launches=%d" % n, " ".join("%s=%.4gM" % (a, b / n / 1e6) for a, b in sorted(v.items())))
