"""Per-kernel statistics from a rocprofv3 kernel_trace.csv with the launches of one kernel told apart by where they ran: "main" = the
stream(s) the channel bank (pfb* kernels) ran on, "side" = any other stream (the tail, the squelch's side stream).  What it is for:
exact_rows_kernel runs twice per batch -- in line over presence's marks (the bench line's roofline kernel) and, on the tail, over the
second run's few rows -- and kernel_stats.csv averages the two.
    python scripts/kernel_stats_by_stream.py kernel_trace.csv out.csv"""
import collections, csv, sys


def main(path, out):
    rows = [r for r in csv.DictReader(open(path)) if "btgpu::" in r["Kernel_Name"]]
    short = lambda n: n.split("(")[0].replace("void ", "").replace("btgpu::", "")
    main_streams = {r.get("Stream_Id", "?") for r in rows if short(r["Kernel_Name"]).startswith("pfb")}
    acc = collections.defaultdict(list)
    for r in rows:
        where = "main" if r.get("Stream_Id", "?") in main_streams else "side"
        acc[(short(r["Kernel_Name"]), where)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as f:
        f.write("kernel,stream,calls,total_ms,avg_us,median_us,min_us,max_us\n")
        for (k, w), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            d.sort()
            f.write("%s,%s,%d,%.3f,%.1f,%.1f,%.1f,%.1f\n" % (k, w, len(d), sum(d) / 1e3, sum(d) / len(d), d[len(d) // 2], d[0], d[-1]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
