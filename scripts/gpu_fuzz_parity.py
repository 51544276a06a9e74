"""Randomised differential run: GPU (DIRECT path, bit-exact contract) vs the oracle over random rates,
SNRs, occupancies, squelch thresholds, modes, LE pass on/off, push chunkings.  GPU only.
    python scripts/gpu_fuzz_parity.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import importlib
import numpy as np
import pyoracle as po
from tests.conftest import load_pkg
pkg = load_pkg()
synth = importlib.import_module("gr_bluetooth_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
RATES = [(2e6, 2476e6), (3e6, 2450e6), (4e6, 2476e6), (5e6, 2470e6), (8e6, 2476.5e6), (8e6, 2402e6), (10e6, 2450e6),
         (16e6, 2440e6), (20e6, 2441e6), (25e6, 2441e6)]
bad = 0
for case in range(cases):
    fs, fc = RATES[int(rng.integers(0, len(RATES)))]
    nsl = int(rng.integers(8, 30)) if fs < 16e6 else int(rng.integers(7, 12))
    snr_db = float(rng.uniform(9, 30)); occ = float(rng.uniform(0.1, 0.9)); sq = float(rng.choice([-5.0, 5.0, 10.0, 14.0]))
    sniff = bool(rng.integers(0, 2)); le = sniff and bool(rng.integers(0, 2))
    iq, _ = synth.make_capture(fs, fc, nsl, laps=(0x24D952, 0x4831DD, 0x9E8B33), seed=int(rng.integers(0, 1 << 30)), snr_db=snr_db,
                               occupancy=occ, max_payload_bits=int(rng.choice([0, 240, 2800])))
    o = po.Oracle(fs, fc, sq, po.MODE_SNIFFER if sniff else po.MODE_LAP, le=le)
    want, _ = o.run_stream(iq, threads=16)
    kw = dict(channelizer=pkg.CHANNELIZER_DIRECT, squelch=pkg.SQUELCH_DIRECT, max_batch_slots=int(rng.choice([0, 3, 8])))
    blk = pkg.multi_sniffer(fs, fc, sq, False, le=le, **kw) if sniff else pkg.multi_LAP(fs, fc, sq, **kw)
    pos = 0
    while pos < len(iq):                                   # ragged pushes
        n = int(rng.integers(1, 3 * o.slot))
        blk.push(iq[pos:pos + n]); pos += n
    got = blk.poll()
    blk.close()
    same = [h.key() for h in got] == [h.key() for h in want]
    if not same:
        bad += 1
    print("case %2d fs %4.1f sniff %d le %d sq %5.1f snr %4.1f occ %.2f slots %2d hits %3d %s" %
          (case, fs / 1e6, sniff, le, sq, snr_db, occ, nsl, len(want), "ok" if same else "MISMATCH"))
print("mismatches:", bad)
