"""btrx_amd -S on its default path against the oracle pipeline's text over N piconet captures (tests/textparity.py): lines compared,
captures with a differing line.  Run on the GPU box:  python scripts/gpu_text_parity.py 200 [first_seed]"""
import os, sys, tempfile, pathlib, difflib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle as po
from tests.conftest import load_pkg
load_pkg()
import test_host_block_default_path as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tot = dict(captures=0, lines=0, captures_differing=0, lines_differing=0, records=0)
with tempfile.TemporaryDirectory() as td:
    for seed in range(s0, s0 + n):
        big = seed % 10 == 0
        got, want, hits = (T.run_case(po, pathlib.Path(td), 100e6, 2441e6, "2441M", "100M", 20, seed, {0: 37, 24: 38, 78: 39}) if big else
                           T.run_case(po, pathlib.Path(td), 8e6, 2476.5e6, "2476.5M", "8M", 60, seed, {78: 39}))
        gl, wl = got.splitlines(), want.splitlines()
        nd = sum(1 for l in difflib.unified_diff(wl, gl, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---"))
        tot["captures"] += 1; tot["lines"] += len(wl); tot["records"] += len(hits); tot["captures_differing"] += nd > 0; tot["lines_differing"] += nd
        if nd:
            print("seed %d: %d differing lines" % (seed, nd)); print("\n".join(list(difflib.unified_diff(wl, gl, lineterm="", n=0))[:12]), flush=True)
print("TOTAL", tot)
