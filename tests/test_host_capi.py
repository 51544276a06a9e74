"""The C-ABI library: loads, exports every symbol include/btgpu.h declares, host-only entry
points work without a GPU, and GPU entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "btgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(btgpu_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), n
    assert set(pkg.EXPORTS) <= set(names)


def test_version_and_strerror(pkg):
    L = pkg.lib()
    assert b"gfx950" in L.btgpu_version()
    assert L.btgpu_strerror(0) == b"ok"
    assert b"gfx950" in L.btgpu_strerror(pkg.ENODEVICE)


def test_design_query_rejects_bad_configs(pkg):
    with pytest.raises(pkg.BtgpuError) as e:
        pkg.design_query(1e6, 2441e6)              # < 2 samples per symbol (apps/btrx:66-78)
    assert e.value.code == pkg.EINVAL
    with pytest.raises(pkg.BtgpuError):
        pkg.design_query(8e6, 2476.5e6, 10.0, mode=7)
    with pytest.raises(pkg.BtgpuError):
        pkg.design_query(8e6, 2300e6)              # no Bluetooth channel in the span
    with pytest.raises(pkg.BtgpuError) as e:       # the libbtbb-style search belongs to multi_LAP only
        pkg.design_query(8e6, 2476.5e6, 10.0, mode=pkg.MODE_SNIFFER, correlator=pkg.CORRELATOR_BTBB)
    assert e.value.code == pkg.EUNSUPPORTED


def test_null_arguments_return_einval(pkg):
    L = pkg.lib()
    assert L.btgpu_design_query(None, None) == pkg.EINVAL
    assert L.btgpu_create(None, None) == pkg.EINVAL
    assert L.btgpu_poll(None, None, 0) == pkg.EINVAL
    assert L.btgpu_work(None, None, 0, None) == pkg.EINVAL
    assert L.btgpu_history(None) == pkg.EINVAL
    L.btgpu_destroy(None)                           # no-op, must not crash


def test_no_cpu_fallback(pkg):
    """Without a GPU the block cannot be constructed: ENODEVICE, never a silent CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.BtgpuError) as e:
        pkg.multi_sniffer(8e6, 2476.5e6, 10.0, False)
    assert e.value.code == pkg.ENODEVICE
    with pytest.raises(pkg.BtgpuError):
        pkg.multi_LAP(8e6, 2476.5e6, 10.0)


def test_staged_squelch_design_fit(pkg):
    """The two-stage squelch filter's composite impulse response must equal the reference's
    low_pass(1, fs, 22.5e3, 10e3, HANN) to ~1e-6 of its l1 norm at every supported rate, and the
    quadrature weights must sum to the 850 outputs (noise_out) they stand for."""
    for fs, fc in ((2e6, 2476e6), (4e6, 2476e6), (8e6, 2476.5e6), (20e6, 2441e6), (100e6, 2441e6)):
        s = pkg.staged_design(fs, fc)
        assert s["fit_l1_error"] < 5e-6, (fs, s)
        assert s["R"] == 5 * int(fs / 1e6) // 2 if fs > 2e6 else s["R"] == 5
        assert s["nw"] == 182 and s["L3"] == 80
        assert abs(s["weight_sum"] - 850.0) < 1e-6


def test_block_mirror_surface(pkg):
    assert pkg.multi_LAP.NAME == "bluetooth multi LAP block"
    assert pkg.multi_sniffer.NAME == "bluetooth multi sniffer block"
    h = pkg.Hit(slot=12, channel=37, offset=5, lap=0x24D952, ac_errors=1, kind=0, nsym=700, snr_db=23.44)
    assert pkg.multi_LAP.format_hit(None, h) == "GOT PACKET: ch=37, LAP=24d952, err=1 at time slot 12"
    assert pkg.multi_sniffer.format_hit(None, h) == "time     12, snr=23.4, channel 37, LAP 24d952 "
    assert ctypes.sizeof(pkg.Hit) == 40 and ctypes.sizeof(pkg.Config) == 56


def test_product_never_imports_oracle():
    """The product tree must not reference oracle/ (checker only)."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "gr-bluetooth_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if "pyoracle" in txt or "bt_oracle.h" in txt or "libbt_oracle" in txt or "bto_" in txt:
                    bad.append(os.path.join(d, f))
    assert bad == []


def test_design_of_odd_samples_per_symbol_rates_matches_oracle(pkg, po):
    """5, 7, 15, 25 Msps: 625 * sps is not a multiple of the decimation (no shared output grid, the
    banks run window by window); the constructor arithmetic is the reference's all the same."""
    for fs, fc in ((5e6, 2470e6), (7e6, 2450e6), (15e6, 2450e6), (25e6, 2441e6)):
        for mode in (pkg.MODE_LAP, pkg.MODE_SNIFFER):
            d = pkg.design_query(fs, fc, 10.0, mode)
            o = po.Oracle(fs, fc, 10.0, mode)
            assert d.samples_per_slot % d.decimation != 0
            assert (d.history, d.ddc_out, d.noise_out, d.decimation, d.low_channel, d.high_channel) == \
                (o.history, o.ddc_out, o.noise_out, o.decim, o.low_ch, o.high_ch)
