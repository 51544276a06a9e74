"""Regenerates the committed fixtures in tests/golden/.  Run in the build container only
(it reads /root/reference/samples/channel37.dem, which does not exist on the GPU box).

  channel37.bits.npy   samples/channel37.dem (one demodulated symbol per byte, values 0/1)
                       bit-packed with numpy.packbits: a reference DATA file, not source.
  channel37_hits.json  access-code hits the reference's classic_packet::sniff_ac yields on
                       it with the sniffer's resume-at-hit+68 policy: (offset, LAP) pairs as
                       recorded in SURVEY.md F3 (33 hits: 24d952 x31, 133bec, f2f57b), here
                       re-derived with oracle/bt_oracle.c and asserted against those counts.
  ac_vectors.json      LAP -> 9-byte access code known answers.
  c8_seed7.json        oracle hit list for a seeded synthetic 8-channel capture (float path;
                       parity unpinned upstream -- this pins the oracle against itself).
"""
import collections
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg  # noqa: E402
import pyoracle as po  # noqa: E402


def main():
    dem = np.fromfile("/root/reference/samples/channel37.dem", dtype=np.uint8)
    assert len(dem) == 3997342 and set(np.unique(dem)) == {0, 1}
    np.save(os.path.join(HERE, "channel37.bits.npy"), np.packbits(dem))
    hits = po.scan_symbols(dem)
    laps = collections.Counter("%06x" % h[1] for h in hits)
    assert len(hits) == 33 and laps == {"24d952": 31, "133bec": 1, "f2f57b": 1}, laps
    assert [h[0] for h in hits[:4]] == [66136, 206587, 225314, 506941]
    json.dump({"n_symbols": int(len(dem)), "hits": [[int(o), "%06x" % l, int(e)] for o, l, e in hits]},
              open(os.path.join(HERE, "channel37_hits.json"), "w"), indent=0)
    vec = {"9e8b33": "5475c58cc73345e72a", "000000": "57e7041e34000000d5", "ffffff": "ae758b5227ffffff2a",
           "123456": "503e461a65a8b120d5", "c6967e": "ab7cc2d999f9a58f2a"}
    for lap, ac in vec.items():
        assert po.acgen(int(lap, 16)).hex() == ac
    json.dump(vec, open(os.path.join(HERE, "ac_vectors.json"), "w"), indent=0)

    load_pkg()
    import importlib
    synth = importlib.import_module("gr_bluetooth_amd.synth")
    out = {}
    # "lap" = multi_LAP with the in-tree correlator, "lap_btbb" = with the libbtbb-style one (the
    # block's default, as in the reference; [EXT] unpinned)
    for mode, name, corr in ((po.MODE_SNIFFER, "sniffer", None), (po.MODE_LAP, "lap", po.CORRELATOR_INTREE),
                             (po.MODE_LAP, "lap_btbb", po.CORRELATOR_BTBB)):
        iq, truth = synth.make_capture(8e6, 2476.5e6, 20, laps=(0x24D952, 0x4831DD, 0x9E8B33), seed=7,
                                       snr_db=22.0, occupancy=0.4)
        hits, done = po.Oracle(8e6, 2476.5e6, 10.0, mode, correlator=corr).run_stream(iq)
        out[name] = [[h.slot, h.channel, h.kind, h.offset, "%06x" % h.lap, h.ac_errors, h.nsym,
                      round(h.snr, 6)] for h in hits]
    out["params"] = dict(sample_rate=8e6, center_freq=2476.5e6, n_slots=20, laps=["24d952", "4831dd", "9e8b33"],
                         seed=7, snr_db=22.0, occupancy=0.4, squelch_db=10.0)
    json.dump(out, open(os.path.join(HERE, "c8_seed7.json"), "w"), indent=0)
    print("golden fixtures written:", {k: len(v) for k, v in out.items() if k != "params"})


if __name__ == "__main__":
    main()
