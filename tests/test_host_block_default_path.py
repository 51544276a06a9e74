"""The drop-in multi_sniffer on its DEFAULT path (polyphase banks + exact stage + exact payload; no BTGPU_AUTO=direct): btrx_amd -S
prints, character for character, what the reference's handlers print for the oracle's records.  What makes that
hold: access code and header come from the exact stage, the payload from the long tasks of BTGPU_FLAG_EXACT_PAYLOAD, which
host/blocks.cc sets."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BTRX = os.path.join(ROOT, "gr-bluetooth_amd", "host", "btrx_amd")


def run_case(po, tmp_path, fs, fc, fstr, rstr, n_slots, seed, le_channels):
    import textparity
    iq, truth, lap, uap = textparity.make_piconet_capture(fs, fc, n_slots, seed, le_channels=le_channels)
    path = str(tmp_path / ("cap_%d.cfile" % seed))
    iq.astype(np.complex64).tofile(path)
    env = {k: v for k, v in os.environ.items() if k != "BTGPU_AUTO"}
    out = subprocess.run([BTRX, "-f", fstr, "-r", rstr, "-i", path, "-S"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER, le=True)
    hits, _ = o.run_stream(iq, threads=os.cpu_count() or 1)
    want = textparity.sniffer_text(po, o, iq, hits)
    got = out.stdout.split("\n", 1)[1]
    os.remove(path)
    return got, want, hits


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_btrx_amd_default_path_prints_reference_lines_8msps(po, synth, tmp_path, seed):
    if not os.path.exists(BTRX):
        subprocess.check_call(["make", "-C", os.path.dirname(BTRX)])
    got, want, hits = run_case(po, tmp_path, 8e6, 2476.5e6, "2476.5M", "8M", 60, seed, {78: 39})
    assert len(hits) >= 8
    assert any(t in want for t in ("DH5", "DH3", "DM1", "DH1")), want[:2000]
    assert got == want


def test_btrx_amd_default_path_prints_reference_lines_100msps(po, synth, tmp_path):
    got, want, hits = run_case(po, tmp_path, 100e6, 2441e6, "2441M", "100M", 24, 11, {0: 37, 24: 38, 78: 39})
    assert len(hits) >= 6
    assert got == want
