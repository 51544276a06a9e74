"""Randomised differential run of the FAST path (100 Msps polyphase bank + staged squelch, tolerance
contract) vs the oracle.  Records whose symbols lie inside a planted burst must be identical on
(slot, channel, kind, LAP, ac_errors), the symbol offset within +-1 (a symbol more or less emitted in
the noise before the burst), nsym within +-8.  Records born from noise-only symbols
(false access addresses between bursts) are counted separately: the clock-recovery loop quantises mu
to 1/128 sample, so a 1e-7 perturbation of the demodulated stream can move a noise symbol by ~0.02
and flip it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import importlib
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
RATES = [(100e6, 2441e6), (8e6, 2476.5e6), (20e6, 2441e6), (100e6, 2441e6)]     # polyphase banks: 100 bins, 8 bins (FFT), 20 bins
bad = 0
for case in range(cases):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 14)); snr_db = float(rng.uniform(12, 30)); occ = float(rng.uniform(0.2, 0.9))
    sq = float(rng.choice([5.0, 10.0, 14.0])); sniff = bool(rng.integers(0, 2)); le = sniff and bool(rng.integers(0, 2))
    laps = tuple(int(x) for x in rng.integers(0, 1 << 24, 6))
    iq, truth = synth.make_capture(fs, fc, nsl, laps=laps, seed=int(rng.integers(0, 1 << 30)), snr_db=snr_db, occupancy=occ)
    o = po.Oracle(fs, fc, sq, po.MODE_SNIFFER if sniff else po.MODE_LAP, le=le)
    want, _ = o.run_stream(iq, threads=32)
    blk = pkg.multi_sniffer(fs, fc, sq, False, le=le) if sniff else pkg.multi_LAP(fs, fc, sq)
    assert blk.design.channelizer == pkg.CHANNELIZER_POLYPHASE and blk.design.squelch == pkg.SQUELCH_STAGED
    blk_history = blk.design.history
    blk.push(iq); got = blk.poll(); blk.close()
    sps = int(fs / 1e6); slot_len = 625 * sps

    H = blk_history

    def in_burst(key):
        pos = key[0] * slot_len - (H - 1) + key[3] * sps
        return any(t["channel"] == key[1] and t["start"] - H <= pos <= t["start"] + t["nbits"] * sps + H for t in truth)
    gk, wk = [h.key()[:6] for h in got], [h.key()[:6] for h in want]
    gs, ws = set(gk), set(wk)
    diff = gs ^ ws
    noise_born = sorted(d for d in diff if not in_burst(d))
    # packet-born differences: allowed only as the same record one symbol earlier / later
    pg = sorted(d for d in gs - ws if in_burst(d)); pw = sorted(d for d in ws - gs if in_burst(d))
    core = len(pg) == len(pw) and all(a[:3] == b[:3] and a[4:] == b[4:] and abs(a[3] - b[3]) <= 1 for a, b in zip(pg, pw))
    shifted = globals().get("shifted", 0) + len(pg)
    both = {h.key()[:6]: h.nsym for h in want}
    dev = max([abs(h.nsym - both[h.key()[:6]]) for h in got if h.key()[:6] in both], default=0)
    ok = core and dev <= 8
    if diff:
        print('   only GPU   :', sorted(gs - ws))
        print('   only oracle:', sorted(ws - gs))
    bad += not ok
    noise_total = globals().get("noise_total", 0) + len(noise_born)
    print("case %2d fs %3.0fM sniff %d le %d sq %4.1f snr %4.1f occ %.2f slots %2d hits %3d packet-born identical %s, noise-born differing %d, nsym dev %d" %
          (case, fs / 1e6, sniff, le, sq, snr_db, occ, nsl, len(want), core, len(noise_born), dev))
print("mismatches on planted bursts:", bad, " offset +-1:", shifted, " noise-born records differing:", noise_total)
