import sys, numpy as np
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import pyoracle as po
from conftest import load_pkg
pkg=load_pkg()
rng=np.random.default_rng(1)
for fs, fc in ((6.4e6, 2470e6), (12.5e6, 2450e6), (2.5e6, 2450e6)):
    try:
        d=pkg.design_query(fs,fc,-100.0,1)
        o=po.Oracle(fs,fc,-100.0,po.MODE_SNIFFER)
        print(fs/1e6, "design", (d.history,d.ddc_out,d.noise_out,d.decimation,d.samples_per_slot)==(o.history,o.ddc_out,o.noise_out,o.decim,o.slot), d.samples_per_slot, d.decimation)
        n = o.slot*12
        iq=(rng.standard_normal(n)+1j*rng.standard_normal(n)).astype(np.complex64)
        want,_=o.run_stream(iq, threads=16)
        blk=pkg.multi_sniffer(fs,fc,-100.0,False,channelizer=1,squelch=1)
        blk.push(iq); got=blk.poll()
        print("   hits", len(want), [h.key() for h in got]==[h.key() for h in want])
        nch = d.high_channel - d.low_channel + 1
        snr = blk.debug_fetch(4, 0, 0, 12 * nch).reshape(12, nch)
        blk.close()
        worst = 0.0
        for k in (0, 5, 11):
            win = o.window(iq, k)
            for ch in (d.low_channel, d.high_channel):
                ch_iq, e_on = o.channel_samples(win, ch)
                ok, s_ref, e_off = o.check_snr(win, ch, e_on)
                worst = max(worst, abs(s_ref - snr[k, ch - d.low_channel]))
        print("   max |snr - oracle| over sampled windows: %.3e dB" % worst)
    except Exception as e:
        print(fs/1e6, "ERR", repr(e)[:300])
