"""the reference's parsers have out-of-bounds quirks (A.3 Q11); the oracle and the product's
host protocol code must not inherit them.  Both are built with -fsanitize=address,undefined and driven over the
reference's captured symbol stream, random symbols at every boundary length, and a short noise capture through
the float front end (oracle/san_check.c, gr-bluetooth_amd/host/san_check.cc).  No GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san(directory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), "san"], capture_output=True, text=True, timeout=1200)
    if "cannot find -lasan" in r.stderr or "libasan" in r.stderr and "cannot" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout


def test_oracle_clean_under_asan_ubsan():
    out = _san("oracle")
    assert "channel37: 33 hits" in out and "san_check: ok" in out


def test_host_protocol_code_clean_under_asan_ubsan():
    assert "host san_check: ok" in _san(os.path.join("gr-bluetooth_amd", "host"))
