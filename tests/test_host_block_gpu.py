"""C++ host side on the GPU box: btrx_amd (GNU Radio block mirror + harness scheduler) prints
the reference's output lines for a capture file; compared with lines built from oracle hits."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# These tests compare printed text with the oracle pipeline character for character (they are about the host
# protocol layer): the blocks' AUTO settings resolve to the bit-exact direct forms here.  The default
# (polyphase) path through the same blocks is covered by test_btrx_amd_time_partitioned_... and by
# tests/test_gpu_parity.py.
@pytest.fixture(autouse=True)
def _bit_exact_auto(monkeypatch):
    monkeypatch.setenv("BTGPU_AUTO", "direct")
BTRX = os.path.join(ROOT, "gr-bluetooth_amd", "host", "btrx_amd")


def _sniffer_text(po, o, iq, hits, tap=None):
    """stdout of multi_sniffer for the oracle's hit list: the oracle's packet handlers
    (multi_sniffer_impl::ac and below) fed with the symbols each hit hands over, LE lines as aa() prints."""
    sn = po.Sniffer(tun=tap is not None)
    text = ""
    for h in hits:
        ch_iq, _ = o.channel_samples(o.window(iq, h.slot), h.channel)
        sym, _ = o.channel_symbols(ch_iq)
        if h.kind != 0:              # aa(): prefix + le_packet::print() on the symbols from the preamble on
            text += "time %6d, snr=%.1f, " % (h.slot, h.snr) + po.le_print(sym[h.offset:h.offset + max(h.nsym, 0)], 2402e6 + 1e6 * h.channel)
            continue
        text += sn.ac(sym[h.offset:h.offset + min(h.nsym, 3125)], h.slot, h.channel, h.snr)
    if tap is not None:
        tap.append(sn.tap())
    return text


@pytest.mark.parametrize("sniff", [False, True])
def test_btrx_amd_prints_reference_lines(po, synth, tmp_path, sniff):
    if not os.path.exists(BTRX):
        subprocess.check_call(["make", "-C", os.path.dirname(BTRX)])
    fs, fc = 8e6, 2476.5e6
    iq, _ = synth.make_capture(fs, fc, 16, laps=(0x24D952, 0x4831DD), seed=31, snr_db=24, occupancy=0.5)
    rng = np.random.default_rng(3)
    for k in (1, 4, 6):                                          # ID packets: 68-symbol access code only
        synth.add_burst(iq, synth.access_code_bits(0x9E8B33)[:68], k * 5000 + 300, fs, fc, 73 + k % 3, rng)
    for k, pdu in ((2, 0), (5, 4), (9, 5), (11, 3)):              # LE adverts on index 39 = classic channel 78
        synth.add_burst(iq, synth.le_advert_bits(39, rng, payload_bytes=34 if pdu == 5 else 18, pdu_type=pdu),
                        k * 5000 + 900, fs, fc, 78, rng)
    path = str(tmp_path / "cap.cfile")
    iq.astype(np.complex64).tofile(path)                      # .cfile = raw interleaved float32 I/Q
    cmd = [BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-c", "3"] + (["-S"] if sniff else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l]
    if sniff:
        assert any(l.endswith("ID") for l in lines)
        assert sum("BTLE index=39, AA=8e89bed6, PDUType=" in l for l in lines) >= 3 and any("CRCInit=" in l for l in lines)
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER if sniff else po.MODE_LAP, le=sniff)   # multi_sniffer runs the LE pass
    hits, _ = o.run_stream(iq)
    assert len(hits) > 3
    assert lines[0] == "history set to %d samples: channel=%d, noise=%d" % (o.history, o.ntaps_ch + o.decim * 8, o.ntaps_noise)
    if sniff:
        want = [l for l in _sniffer_text(po, o, iq, hits).splitlines() if l]
        assert out.stdout.split("\n", 1)[1] == _sniffer_text(po, o, iq, hits)      # blank lines of le_packet::print included
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


def test_btrx_amd_uap_discovery_on_captured_symbols(po, synth, tmp_path):
    """The first 1.2 s of samples/channel37.dem (real captured symbols, committed bit-packed) re-modulated
    as GFSK on channel 37 of an 8 Msps capture: btrx_amd -S must print what the reference's multi_sniffer
    prints -- the UAP/CLK1-6 discovery dialogue, the winner (UAP 0xaf, the reference's own answer for this
    capture), the queued packets decoded, POLL / HV3 packets after that -- equal to the oracle pipeline
    (oracle front end + oracle packet handlers) on the same samples."""
    import json
    G = os.path.join(ROOT, "tests", "golden")
    n = json.load(open(os.path.join(G, "channel37_hits.json")))["n_symbols"]
    bits = np.unpackbits(np.load(os.path.join(G, "channel37.bits.npy")))[:n]
    fs, fc = 8e6, 2476.5e6
    nsl = 1200
    rng = np.random.default_rng(9)
    iq = (0.02 * (rng.standard_normal(nsl * 5000) + 1j * rng.standard_normal(nsl * 5000))).astype(np.complex64)
    # every access-code hit of the stretch with its packet (3000 symbols cover the longest type), at
    # the sample its first symbol had in the capture
    for off, lap, errs in po.scan_symbols(bits):
        if (off + 3200) * 8 >= len(iq):
            break
        synth.add_burst(iq, bits[off - 8:off + 3000], (off - 8) * 8, fs, fc, 73, rng, cfo_hz=5e3)
    path = str(tmp_path / "c37.cfile")
    iq.tofile(path)
    tap_path = str(tmp_path / "tap.bin")        # -w: the Wireshark TAP frames (no TUN device here: BTGPU_TAP_FILE)
    out = subprocess.run([BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-S", "-w"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, BTGPU_TAP_FILE=tap_path))
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER, le=True)
    hits, _ = o.run_stream(iq, threads=16)
    frames = []
    want = _sniffer_text(po, o, iq, hits, tap=frames)
    got_frames = open(tap_path, "rb").read()
    # Ethernet-framed packets as lib/tun.cc writes them: ethertype 0xFFF0, destination MAC = 00:00:UAP:LAP
    assert len(frames[0]) > 200 and got_frames == frames[0]
    assert bytes.fromhex("0000af24d952" + "000000000000" + "fff0") in got_frames
    got = out.stdout.split("\n", 1)[1]
    assert "We have a winner! UAP = 0xaf" in want and "Decoding queued packets" in want
    assert got == want


def test_btrx_amd_hopper_follows_a_hopping_piconet(po, synth, tmp_path):
    """btrx_amd -l LAP -p (multi_hopper): GPU front end + GPU header sweep + GPU hop reversal
    (btgpu_hopseq_*) + host piconet logic print exactly what the oracle pipeline prints: UAP / CLK1-6
    discovery, candidate winnowing down to the master clock, then one line per followed packet."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("toh", os.path.join(ROOT, "tests", "test_oracle_hop.py"))
    toh = importlib.util.module_from_spec(spec); spec.loader.exec_module(toh)
    fs, fc = 8e6, 2476.5e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x3A5C7E1, 700
    iq, truth = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=5, dh1_fraction=0.0)
    path = str(tmp_path / "hop.cfile")
    iq.tofile(path)
    out = subprocess.run([BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-l", "%06x" % lap, "-p"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    hits, _ = o.run_stream(iq, threads=16)
    want, hb = toh._hopper_text(po, o, iq, hits, lap, nsl)
    got = out.stdout.split("\n", 1)[1]
    assert "Acquired CLK1-27 offset = 0x%07x" % ((clk0 - 6) & 0x7FFFFFF) in want
    assert got == want


@pytest.mark.parametrize("path_kind", ["direct", "polyphase"])
def test_btrx_amd_hopper_full_band_100msps(po, synth, tmp_path, monkeypatch, path_kind):
    """BASELINE configs[4] on one GPU: the whole 79-channel band at 100 Msps, a master hopping over all
    channels, btrx_amd -l LAP -p (multi_hopper).  Every hop is visible, so UAP / CLK1-6 fall after nine
    packets and CLK1-27 after about twenty; text equal to the oracle pipeline -- on the bit-exact DIRECT
    front end and on the default polyphase front end (the packets' symbols are the same on both)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("toh", os.path.join(ROOT, "tests", "test_oracle_hop.py"))
    toh = importlib.util.module_from_spec(spec); spec.loader.exec_module(toh)
    if path_kind == "polyphase":
        monkeypatch.delenv("BTGPU_AUTO", raising=False)
    fs, fc = 100e6, 2441e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x3A5C7E1, 90
    iq, truth = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=5, dh1_fraction=0.0)
    path = str(tmp_path / "hop100.cfile")
    iq.tofile(path)
    out = subprocess.run([BTRX, "-f", "2441M", "-r", "100M", "-i", path, "-l", "%06x" % lap, "-p", "-c", "16"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    hits, _ = o.run_stream(iq, threads=min(64, os.cpu_count() or 8))
    want, hb = toh._hopper_text(po, o, iq, hits, lap, nsl)
    got = out.stdout.split("\n", 1)[1]
    assert "We have a winner! UAP = 0xaf" in want
    assert "Acquired CLK1-27 offset = 0x%07x" % ((clk0 - 6) & 0x7FFFFFF) in want
    assert got == want


def test_btrx_amd_hopper_with_an_aliasing_receiver(po, synth, tmp_path):
    """btrx_amd -l LAP -p --aliased on a 25 Msps capture that folds all 79 channels into 26..50 (odd
    samples per symbol: the segmented DIRECT path): hop reversal on aliased channel numbers, then one
    line per followed packet on the channel it was observed on -- text equal to the oracle pipeline."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("toh", os.path.join(ROOT, "tests", "test_oracle_hop.py"))
    toh = importlib.util.module_from_spec(spec); spec.loader.exec_module(toh)
    fs, fc = 25e6, 2440e6
    lap, uap, clk0, nsl = 0x24D952, 0xAF, 0x1B3C5D2, 200
    iq, truth = synth.make_hopping_capture(fs, fc, nsl, lap, uap, clk0, seed=8, dh1_fraction=0.0, aliased=True)
    path = str(tmp_path / "alias.cfile")
    iq.tofile(path)
    out = subprocess.run([BTRX, "-f", "2440M", "-r", "25M", "-i", path, "-l", "%06x" % lap, "-p", "--aliased"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER)
    hits, _ = o.run_stream(iq, threads=16)
    want, hb = toh._hopper_text(po, o, iq, hits, lap, nsl, aliased=True)
    got = out.stdout.split("\n", 1)[1]
    assert "Acquired CLK1-27 offset = 0x%07x" % ((clk0 - 6) & 0x7FFFFFF) in want
    assert got == want


def test_btrx_amd_fhs_after_discovery(po, synth, tmp_path):
    """A piconet seen through header-only packets until UAP / CLK1-6 are known, then an FHS and a DM1
    packet: the sniffer block prints the decoded types, the FHS contents (BD_ADDR, CLK) and adopts
    them -- text equal to the oracle pipeline."""
    fs, fc = 8e6, 2476.5e6
    lap, uap, clk0, nsl = 0x4831DD, 0x6B, 0x155AA00, 90
    rng = np.random.default_rng(21)
    iq = (0.05 * (rng.standard_normal(nsl * 5000) + 1j * rng.standard_normal(nsl * 5000))).astype(np.complex64)
    for k in range(4, 60, 4):
        bits = synth.classic_poll_bits(lap, uap, clk0 + k, lt_addr=int(rng.integers(1, 8)), flow=int(rng.integers(0, 2)),
                                       arqn=int(rng.integers(0, 2)), seqn=int(rng.integers(0, 2)), ptype=int(rng.integers(0, 2)))
        synth.add_burst(iq, bits, k * 5000 + 320, fs, fc, 72 + k % 5, rng, cfo_hz=4e3)
    synth.add_burst(iq, synth.classic_fhs_bits(lap, uap, clk0 + 64, 0xABCDEF, 0x47, 0x1234, 0x1F2E3D4 >> 1), 64 * 5000 + 320, fs, fc, 75, rng)
    synth.add_burst(iq, synth.classic_dm1_bits(lap, uap, clk0 + 70, b"hello dm1"), 70 * 5000 + 320, fs, fc, 73, rng)
    path = str(tmp_path / "fhs.cfile")
    iq.tofile(path)
    out = subprocess.run([BTRX, "-f", "2476.5M", "-r", "8M", "-i", path, "-S"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    o = po.Oracle(fs, fc, 10.0, po.MODE_SNIFFER, le=True)
    hits, _ = o.run_stream(iq, threads=16)
    want = _sniffer_text(po, o, iq, hits)
    # packets every fourth slot leave six CLK1-6 candidates; the FHS payload CRC settles it
    assert "Correct CRC! UAP = 0x6b found after 15 total packets." in want
    assert "FHS contents: BD_ADDR 00:34:47:ab:cd:ef, CLK 1f2e3d4" in want and "DM1\n  LLID: 2\n" in want
    assert out.stdout.split("\n", 1)[1] == want


@pytest.mark.parametrize("rate,fc,sniff,nslots", [("8M", 2476.5e6, True, 23), ("8M", 2476.5e6, False, 23), ("100M", 2441e6, True, 11)])
def test_btrx_amd_time_partitioned_over_devices_prints_the_same(synth, tmp_path, rate, fc, sniff, nslots):
    """btrx_amd --gpus N (multi_block::run_partitioned): the capture cut into N contiguous slot ranges, one btgpu
    handle and one host thread per range (every range on device 0 here: one-GPU box), halo of history()-1 plus
    the staged squelch's margin in front of each range, records concatenated in range order -- stdout equals
    the single-device run byte for byte, packet handlers (UAP discovery state, ID suffixes, LE lines) included."""
    fs = 8e6 if rate == "8M" else 100e6
    iq, _ = synth.make_capture(fs, fc, nslots, laps=(0x24D952, 0x4831DD, 0x9E8B33), seed=77, snr_db=24, occupancy=0.6)
    path = str(tmp_path / "cap.cfile")
    iq.astype(np.complex64).tofile(path)
    base = [BTRX, "-f", "%.1fM" % (fc / 1e6), "-r", rate, "-i", path] + (["-S"] if sniff else [])
    env = dict(os.environ)
    env.pop("BTGPU_AUTO", None)                                  # the blocks' default path: polyphase banks, staged squelch
    one = subprocess.run(base, capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stderr
    assert len(one.stdout.splitlines()) > 5
    for n in (2, 3):
        many = subprocess.run(base + ["--gpus", str(n), "--all-on-device0"], capture_output=True, text=True, timeout=300, env=env)
        assert many.returncode == 0, many.stderr
        assert many.stdout == one.stdout, n
