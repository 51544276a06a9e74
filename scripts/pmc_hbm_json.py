"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE counter_collection.csv (two separate passes)
-> per-kernel HBM bytes per launch as JSON (btgpu kernels only).

    python scripts/pmc_hbm_json.py fetch_counter_collection.csv write_counter_collection.csv SLOTS

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md: the counters are in KiB...
FETCH_SIZE is doubled on gfx950 (it reports half of a wide coalesced streaming read)."""
import collections, csv, hashlib, json, os, sys


def per_launch(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if "btgpu" not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("btgpu::", "")[:34]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    LAUNCHES.update({k: len(n[k]) for k in n})
    return {k: tot[k] / max(len(n[k]), 1) for k in tot}


LAUNCHES = {}


def build_id():
    """sha256 (16 hex) of the libbtgpu.so the counters were collected on (bench.py --pmc-json checks it)"""
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr-bluetooth_amd", "libbtgpu.so")
    return hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]


def main():
    fetch, write, slots = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 1 "
                      "--warmup 0 --prewarm-ms 0 --no-cpu --sync --slots %d" % slots,
           "note": "counter unit KiB, averaged per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 "
                   "of a wide coalesced streaming read); algorithmic bytes per launch of a bank kernel = 8 * slots * 62500",
           "slots": slots, "build_id": build_id(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
        # (exact_rows_kernel runs twice per step -- over presence's marks, then over the second run's: `launches` says how many the average is over)
        out["kernels"][k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "hbm_bytes": 2 * f + w, "launches": LAUNCHES.get(k, 1)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
