"""bench.py's roofline.traffic may come from a PMC summary of ANOTHER build of libbtgpu.so only where the measured kernel's device
code is the same (scripts/device_code_ids.py): the rule, checked on the in-tree library without a GPU."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_for_evidence_test", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_traffic_evidence_follows_the_kernels_device_code():
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import device_code_ids as dci
    ids = dci.kernel_code_ids(os.path.join(ROOT, "gr-bluetooth_amd", "libbtgpu.so"))
    assert len(ids) >= 60 and "pfb100f_kernel<256, true, 10, 255>" in ids, sorted(ids)[:5]
    bank = "pfb100f_kernel<256, true, 10, 255>"
    got = b.traffic_by_device_code("pfb", 2304)
    # every stamped summary of this batch size, newest name first (the order bench.py looks in): the first whose bank kernel has this
    # build's instructions is the evidence; if none has, the bank kernel is un-measured
    import glob
    want = None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm.json")), reverse=True):
        pj = json.load(open(cand))
        if pj.get("slots") == 2304 and isinstance(pj.get("kernel_code_sha"), dict) and pj["kernel_code_sha"].get(bank) == ids[bank]:
            want = (os.path.basename(cand), pj["kernels"][bank]["hbm_bytes"])
            break
    if want:
        assert got and got["traffic_source"] == want[0] and got["traffic"] == want[1], (got, want)
        assert 3.0e9 < got["traffic"] < 4.0e9
    else:
        assert got is None, got                                                         # a changed bank kernel is un-measured
    # a kernel whose instructions differ from every stamped summary's gets nothing; an unknown one neither
    for key in ("window_kernel", "no_such_kernel"):
        g = b.traffic_by_device_code(key, 2304)
        if g is not None:
            src = json.load(open(os.path.join(ROOT, "profiles", g["traffic_source"])))
            k = [k for k in src["kernels"] if k.startswith(key)]
            assert k and src["kernel_code_sha"][k[0]] == dci.lookup(ids, k[0])
    assert b.traffic_by_device_code("pfb", 12345) is None                               # another batch size: other bytes per launch
