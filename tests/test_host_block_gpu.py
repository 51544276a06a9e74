"""C++ host side on the GPU box: btrx_amd (GNU Radio block mirror + harness scheduler) prints
the reference's output lines for a capture file; compared with lines built from oracle hits."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BTRX = os.path.join(ROOT, "gr-bluetooth_amd", "host", "btrx_amd")


@pytest.mark.parametrize("sniff", [False, True])
def test_btrx_amd_prints_reference_lines(po, synth, tmp_path, sniff):
    if not os.path.exists(BTRX):
        subprocess.check_call(["make", "-C", os.path.dirname(BTRX)])
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, 16, laps=(0x24D952, 0x4831DD), seed=31, snr_db=24, occupancy=0.5)
    rng = np.random.default_rng(3)
    for k in (1, 4, 6):                                          # ID packets: 68-symbol access code only
        synth.add_burst(iq, synth.access_code_bits(0x9E8B33)[:68], k * 5000 + 300, fs, fc, 73 + k % 3, rng)
    path = str(tmp_path / "cap.cfile")
    iq.astype(np.complex64).tofile(path)                      # .cfile = raw interleaved float32 I/Q
    cmd = [BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-c", "3"] + (["-S"] if sniff else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l]
    if sniff:
        assert any(l.endswith("ID") for l in lines) and any(l.endswith(" ") for l in lines)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER if sniff else po.MODE_LAP, le=sniff)   # multi_sniffer runs the LE pass
    hits, _ = o.run_stream(iq)
    assert len(hits) > 3
    assert lines[0] == "history set to %d samples: channel=%d, noise=%d" % (o.history, o.ntaps_ch + o.decim * 8, o.ntaps_noise)
    if sniff:
        def le_index(ch):
            chan = ch // 2
            return 37 if chan == 0 else 38 if chan == 12 else 39 if chan == 39 else (chan - 1 if chan < 12 else chan - 2)
        def tail(h):
            ch_iq, _ = o.channel_samples(o.window(iq, h.slot), h.channel)
            sym, _ = o.channel_symbols(ch_iq)
            s = sym[h.offset:]
            return "" if po.header_present(s[:200], min(h.nsym, 3125)) else "ID"
        want = [("time %6d, snr=%.1f, channel %2d, LAP %06x " % (h.slot, h.snr, h.channel, h.lap) + tail(h)) if h.kind == 0 else
                ("time %6d, snr=%.1f, BTLE index=%02d, AA=%08x" % (h.slot, h.snr, le_index(h.channel), h.lap)) for h in hits]
    else:
        want = ["GOT PACKET: ch=%d, LAP=%06x, err=%u at time slot %d" % (h.channel, h.lap, h.ac_errors, h.slot) for h in hits]
    assert lines[1:] == want


def test_btrx_amd_int16_input_and_head_limit(po, synth, tmp_path):
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, 12, laps=(0x24D952,), seed=32, snr_db=24, occupancy=0.7, amplitude=1000.0)
    q = np.round(iq.view(np.float32)).astype(np.int16)        # btrx -s: interleaved shorts
    path = str(tmp_path / "cap.sfile")
    q.tofile(path)
    cmd = [BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-s", "-N", str(10 * 5000)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith("GOT PACKET")]
    deq = q.astype(np.float32).view(np.complex64)[:10 * 5000]
    hits, _ = po.Oracle(fs, fc, 10.0, po.MODE_LAP).run_stream(deq)
    want = ["GOT PACKET: ch=%d, LAP=%06x, err=%u at time slot %d" % (h.channel, h.lap, h.ac_errors, h.slot) for h in hits]
    assert len(want) > 2 and got == want
