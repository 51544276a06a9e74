import sys, numpy as np
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import pyoracle as po
from conftest import load_pkg
pkg=load_pkg()
import importlib
synth=importlib.import_module("gr_bluetooth_amd.synth")
for fs, fc, ns in ((10e6, 2450e6, 14), (16e6, 2440e6, 10), (25e6, 2441e6, 8), (50e6, 2441e6, 6), (5e6, 2470e6, 20), (3e6, 2450e6, 30)):
    try:
        iq, truth = synth.make_capture(fs, fc, ns, laps=(0x24D952, 0x4831DD), seed=3, snr_db=24, occupancy=0.6)
        for mode, cls in ((po.MODE_SNIFFER, pkg.multi_sniffer), (po.MODE_LAP, pkg.multi_LAP)):
            want,_ = po.Oracle(fs, fc, 10.0, mode).run_stream(iq, threads=16)
            for ch, sq in ((pkg.CHANNELIZER_DIRECT, pkg.SQUELCH_DIRECT), (0, 0)):
                blk = cls(fs, fc, 10.0, False, channelizer=ch, squelch=sq) if cls is pkg.multi_sniffer else cls(fs, fc, 10.0, channelizer=ch, squelch=sq)
                blk.push(iq); got = blk.poll()
                d = blk.design
                same = [h.key() for h in got] == [h.key() for h in want]
                core = [h.key()[:6] for h in got] == [h.key()[:6] for h in want]
                print(fs/1e6, "mode", mode, "ch/sq", d.channelizer, d.squelch, "hits", len(want), "exact", same, "core", core)
                blk.close()
    except Exception as e:
        print(fs/1e6, "ERR", repr(e)[:200])
