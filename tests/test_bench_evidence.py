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
    pj = json.load(open(os.path.join(ROOT, "profiles", "r05_m_c79_pmc_hbm.json")))
    bank = "pfb100f_kernel<256, true, 10, 255>"
    got = b.traffic_by_device_code("pfb", 2304)
    if pj["build_id"] == b.build_id() or pj["kernel_code_sha"][bank] == ids[bank]:
        # the C79 bank kernel is the one the PMC passes ran: its bytes are evidence for this build
        assert got and got["traffic"] == pj["kernels"][bank]["hbm_bytes"] and got["traffic_source"].endswith("pmc_hbm.json"), got
        assert 3.0e9 < got["traffic"] < 4.0e9
    else:
        assert got is None or got["traffic_source"] != "r05_m_c79_pmc_hbm.json", got    # a changed bank kernel is un-measured
    # a kernel whose instructions differ from every stamped summary's gets nothing; an unknown one neither
    for key in ("window_kernel", "no_such_kernel"):
        g = b.traffic_by_device_code(key, 2304)
        if g is not None:
            src = json.load(open(os.path.join(ROOT, "profiles", g["traffic_source"])))
            k = [k for k in src["kernels"] if k.startswith(key)]
            assert k and src["kernel_code_sha"][k[0]] == dci.lookup(ids, k[0])
    assert b.traffic_by_device_code("pfb", 12345) is None                               # another batch size: other bytes per launch
