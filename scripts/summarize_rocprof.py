"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table
(btgpu kernels in full, everything else grouped) for profiles/."""
import csv
import sys


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    keep, other_ns, other_calls = [], 0, 0
    for r in rows:
        name = r["Name"]
        if "btgpu::" in name:
            short = name.split("(")[0].replace("void ", "")
            keep.append((short, int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                         int(r["MinNs"]), int(r["MaxNs"])))
        else:
            other_ns += int(r["TotalDurationNs"]); other_calls += int(r["Calls"])
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us\n")
        for k in sorted(keep, key=lambda x: -x[2]):
            f.write("%s,%d,%.3f,%.1f,%.1f,%.1f\n" % (k[0], k[1], k[2] / 1e6, k[3] / 1e3, k[4] / 1e3, k[5] / 1e3))
        f.write("(non-btgpu kernels: synthetic capture generation by torch),%d,%.3f,,,\n" % (other_calls, other_ns / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
