"""ctypes binding of the CPU oracle (oracle/libbt_oracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbt_oracle.so")

MODE_LAP, MODE_SNIFFER = 0, 1
CORRELATOR_INTREE, CORRELATOR_BTBB = 1, 2   # multi_LAP default: BTBB (libbtbb, as the reference)
MM_WINDOWED_RESET, MM_REF_FAITHFUL = 0, 1
KIND_AC, KIND_AA = 0, 1


class Hit(ctypes.Structure):
    _fields_ = [("slot", ctypes.c_uint32), ("channel", ctypes.c_int32),
                ("offset", ctypes.c_int32), ("lap", ctypes.c_uint32),
                ("ac_errors", ctypes.c_int32), ("kind", ctypes.c_int32),
                ("nsym", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("snr", ctypes.c_double)]

    def key(self):
        return (self.slot, self.channel, self.kind, self.offset, self.lap, self.ac_errors, self.nsym)


def build(force=False):
    """Compile the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "bt_oracle.c")
    hdr = os.path.join(_HERE, "bt_oracle.h")
    stale = (not os.path.exists(_SO)
             or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = ctypes.CDLL(_SO)
    c_fp = ctypes.POINTER(ctypes.c_float)
    c_dp = ctypes.POINTER(ctypes.c_double)
    vp = ctypes.c_void_p
    L.bto_create.restype = vp
    L.bto_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    L.bto_destroy.argtypes = [vp]
    L.bto_set_mm_policy.argtypes = [vp, ctypes.c_int]
    L.bto_set_le.argtypes = [vp, ctypes.c_int]
    for name in ("history", "samples_per_slot", "decimation", "low_channel", "high_channel",
                 "first_channel_sample", "first_noise_sample", "ntaps_channel", "ntaps_noise",
                 "ddc_out", "noise_out"):
        f = getattr(L, "bto_" + name)
        f.restype = ctypes.c_int
        f.argtypes = [vp]
    for name in ("channel_taps", "noise_taps", "mmse_taps", "atan_table"):
        f = getattr(L, "bto_" + name)
        f.restype = c_fp
        f.argtypes = [vp]
    L.bto_firdes_ntaps.restype = ctypes.c_int
    L.bto_firdes_ntaps.argtypes = [ctypes.c_double, ctypes.c_double]
    L.bto_firdes_low_pass.restype = ctypes.c_int
    L.bto_firdes_low_pass.argtypes = [ctypes.c_double] * 4 + [c_fp, ctypes.c_int]
    L.bto_fast_atan2f.restype = ctypes.c_float
    L.bto_fast_atan2f.argtypes = [vp, ctypes.c_float, ctypes.c_float]
    L.bto_mmse_interpolate.restype = ctypes.c_float
    L.bto_mmse_interpolate.argtypes = [vp, c_fp, ctypes.c_float]
    L.bto_channel_samples.restype = ctypes.c_int
    L.bto_channel_samples.argtypes = [vp, ctypes.c_int, c_fp, c_fp, c_dp]
    L.bto_check_snr.restype = ctypes.c_int
    L.bto_check_snr.argtypes = [vp, ctypes.c_int, ctypes.c_double, c_fp, c_dp, c_dp]
    L.bto_demod.argtypes = [vp, c_fp, c_fp, ctypes.c_int]
    L.bto_mm_cr.restype = ctypes.c_int
    L.bto_mm_cr.argtypes = [vp, c_fp, ctypes.c_int, c_fp, ctypes.c_int]
    L.bto_channel_symbols.restype = ctypes.c_int
    L.bto_channel_symbols.argtypes = [vp, c_fp, ctypes.c_int, ctypes.c_char_p, c_fp]
    L.bto_acgen.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8)]
    L.bto_ac_errors.restype = ctypes.c_int
    L.bto_ac_errors.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    L.bto_check_ac.restype = ctypes.c_int
    L.bto_check_ac.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    L.bto_sniff_ac.restype = ctypes.c_int
    L.bto_sniff_ac.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.bto_sniff_aa.restype = ctypes.c_int
    L.bto_sniff_aa.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_double]
    L.bto_header_present.restype = ctypes.c_int
    L.bto_header_present.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.bto_le_freq2index.restype = ctypes.c_int
    L.bto_le_freq2index.argtypes = [ctypes.c_double]
    L.bto_btbb_find_ac.restype = ctypes.c_int
    L.bto_btbb_find_ac.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32),
                                   ctypes.POINTER(ctypes.c_int)]
    L.bto_set_correlator.restype = None
    L.bto_set_correlator.argtypes = [vp, ctypes.c_int]
    L.bto_try_clock.restype = ctypes.c_int
    L.bto_try_clock.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.bto_crc_check.restype = ctypes.c_int
    L.bto_crc_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.bto_piconet_init.restype = None
    L.bto_piconet_init.argtypes = [ctypes.POINTER(PiconetState), ctypes.c_uint32]
    L.bto_uap_from_header.restype = ctypes.c_int
    L.bto_uap_from_header.argtypes = [ctypes.POINTER(PiconetState), ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32,
                                      ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    L.bto_hopper_block_new.restype = vp
    L.bto_hopper_block_new.argtypes = [ctypes.c_uint32, ctypes.c_int]
    L.bto_hopper_block_free.restype = None
    L.bto_hopper_block_free.argtypes = [vp]
    L.bto_hopper_block_slot.restype = None
    L.bto_hopper_block_slot.argtypes = [vp, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                        ctypes.c_char_p, ctypes.c_size_t]
    L.bto_hopper_block_piconet.restype = ctypes.POINTER(PiconetState)
    L.bto_hopper_block_piconet.argtypes = [vp]
    L.bto_hopper_new.restype = vp
    L.bto_hopper_new.argtypes = [ctypes.c_uint32, ctypes.c_int]
    L.bto_hopper_free.restype = None
    L.bto_hopper_free.argtypes = [vp]
    L.bto_single_hop.restype = ctypes.c_int
    L.bto_single_hop.argtypes = [vp, ctypes.c_uint32]
    L.bto_gen_hops.restype = ctypes.POINTER(ctypes.c_uint8)
    L.bto_gen_hops.argtypes = [vp]
    L.bto_hop_init_candidates.restype = ctypes.c_int
    L.bto_hop_init_candidates.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.bto_hop_winnow.restype = ctypes.c_int
    L.bto_hop_winnow.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.bto_hop_candidates.restype = ctypes.c_int
    L.bto_hop_candidates.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    L.bto_aliased_channel.restype = ctypes.c_int
    L.bto_aliased_channel.argtypes = [ctypes.c_int]
    L.bto_sniffer_new.restype = vp
    L.bto_sniffer_new.argtypes = []
    L.bto_sniffer_free.restype = None
    L.bto_sniffer_free.argtypes = [vp]
    L.bto_sniffer_set_tun.restype = None
    L.bto_sniffer_set_tun.argtypes = [vp, ctypes.c_int]
    L.bto_sniffer_tap.restype = ctypes.c_size_t
    L.bto_sniffer_tap.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.bto_sniffer_ac.restype = None
    L.bto_sniffer_ac.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_double,
                                 ctypes.c_char_p, ctypes.c_size_t]
    L.bto_unfec23.restype = ctypes.c_int
    L.bto_unfec23.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.bto_le_print.restype = ctypes.c_int
    L.bto_le_print.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p, ctypes.c_size_t]
    L.bto_work.restype = ctypes.c_int
    L.bto_work.argtypes = [vp, c_fp, ctypes.c_uint32, ctypes.POINTER(Hit), ctypes.c_int]
    L.bto_run_stream.restype = ctypes.c_int
    L.bto_run_stream.argtypes = [vp, c_fp, ctypes.c_size_t, ctypes.POINTER(Hit), ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_int)]
    L.bto_run_stream_mt.restype = ctypes.c_int
    L.bto_run_stream_mt.argtypes = [vp, c_fp, ctypes.c_size_t, ctypes.POINTER(Hit), ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.bto_scan_symbols.restype = ctypes.c_int
    L.bto_scan_symbols.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(Hit),
                                   ctypes.c_int]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _iq_as_f32(iq):
    """complex64 array or interleaved float32 array -> contiguous float32 view."""
    a = np.ascontiguousarray(iq)
    if a.dtype == np.complex64:
        a = a.view(np.float32)
    assert a.dtype == np.float32
    return a


def acgen(lap):
    ac = (ctypes.c_uint8 * 9)()
    lib().bto_acgen(lap, ac)
    return bytes(ac)


def ac_bits(lap):
    """72 air-order bits (uint8 0/1) of the access code."""
    return np.unpackbits(np.frombuffer(acgen(lap), dtype=np.uint8))


def sniff_ac(symbols, limit=None):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    if limit is None:
        limit = len(s) - 67
    pad = np.concatenate([s, np.zeros(80, np.uint8)])
    return lib().bto_sniff_ac(pad.tobytes(), int(limit))


def sniff_aa(symbols, limit, freq):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    pad = np.concatenate([s, np.zeros(80, np.uint8)])
    return lib().bto_sniff_aa(pad.tobytes(), int(limit), float(freq))


def btbb_find_ac(symbols, search_length=None, max_ac_errors=1):
    """[EXT libbtbb, unpinned] -> (offset of the sync word or -1, lap, ac_errors)."""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    if search_length is None:
        search_length = len(s) - 63
    pad = np.concatenate([s, np.zeros(80, np.uint8)])
    lap, errs = ctypes.c_uint32(0), ctypes.c_int(0)
    off = lib().bto_btbb_find_ac(pad.tobytes(), int(search_length), int(max_ac_errors), ctypes.byref(lap), ctypes.byref(errs))
    return off, int(lap.value), int(errs.value)


class PiconetState(ctypes.Structure):
    _fields_ = [("lap", ctypes.c_uint32), ("got_first_packet", ctypes.c_int), ("packets_observed", ctypes.c_int),
                ("total_packets_observed", ctypes.c_int), ("first_pkt_time", ctypes.c_uint32),
                ("clock6_candidates", ctypes.c_int * 64), ("clk_offset", ctypes.c_uint32), ("uap", ctypes.c_int),
                ("have_uap", ctypes.c_int), ("have_clk6", ctypes.c_int), ("have_clk27", ctypes.c_int),
                ("pattern_indices", ctypes.c_int * 1000), ("pattern_channels", ctypes.c_uint8 * 1000),
                ("winnowed", ctypes.c_int), ("num_candidates", ctypes.c_int), ("hop_reversal_inited", ctypes.c_int),
                ("aliased", ctypes.c_int), ("afh", ctypes.c_int), ("looks_like_afh", ctypes.c_int),
                ("hops", ctypes.c_void_p)]


class Piconet:
    """UAP / CLK1-6 discovery state of one LAP (basic_rate_piconet_impl::UAP_from_header)."""

    def __init__(self, lap):
        self.st = PiconetState()
        lib().bto_piconet_init(ctypes.byref(self.st), lap)

    def uap_from_header(self, symbols, clkn, channel=0):
        """-> (resolved, lines the reference prints)"""
        s = np.ascontiguousarray(symbols, dtype=np.uint8)
        log = ctypes.create_string_buffer(2048)
        r = lib().bto_uap_from_header(ctypes.byref(self.st), s.tobytes() + bytes(64), len(s), int(clkn), int(channel), log, 2048)
        return bool(r), log.value.decode()


class Hopper:
    """Hop reversal of one piconet (lib/piconet_impl.cc:96-338): selection kernel, full table,
    candidate list.  [PARITY UNPINNED: the reference holds no hop vectors.]"""
    LENGTH = 1 << 27

    def __init__(self, address, afh=False):
        self.h = lib().bto_hopper_new(int(address) & 0xFFFFFFF, 1 if afh else 0)

    def __del__(self):
        try:
            if self.h:
                lib().bto_hopper_free(self.h)
                self.h = None
        except Exception:
            pass

    def single_hop(self, clock):
        return lib().bto_single_hop(self.h, int(clock))

    def table(self):
        return np.ctypeslib.as_array(lib().bto_gen_hops(self.h), (self.LENGTH,)).copy()   # outlives the handle

    def init_candidates(self, channel, known_clock_bits, aliased=False):
        return lib().bto_hop_init_candidates(self.h, int(channel), int(known_clock_bits), 1 if aliased else 0)

    def winnow(self, offset, channel, aliased=False):
        return lib().bto_hop_winnow(self.h, int(offset), int(channel), 1 if aliased else 0)

    def candidates(self, cap=1 << 22):
        buf = np.zeros(cap, np.uint32)
        n = lib().bto_hop_candidates(self.h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap)
        return buf[:min(n, cap)]


class HopperBlock:
    """gr::bluetooth::multi_hopper's work() on top of the front end's hit list
    (lib/multi_hopper_impl.cc:76-209): one call per time slot with the first access-code hit of
    every channel that has one."""

    def __init__(self, lap, aliased=False):
        self.h = lib().bto_hopper_block_new(int(lap), 1 if aliased else 0)

    def __del__(self):
        try:
            if self.h:
                lib().bto_hopper_block_free(self.h)
                self.h = None
        except Exception:
            pass

    def slot(self, clkn, hits, low_channel, high_channel):
        """hits: list of (channel, symbols) in ascending channel order -> printed text"""
        n = len(hits)
        ch = (ctypes.c_int * max(n, 1))(*[int(c) for c, _ in hits])
        bufs = [np.ascontiguousarray(s, dtype=np.uint8).tobytes() + bytes(64) for _, s in hits]
        ptrs = (ctypes.c_char_p * max(n, 1))(*bufs)
        lens = (ctypes.c_int * max(n, 1))(*[len(s) for _, s in hits])
        log = ctypes.create_string_buffer(1 << 16)
        lib().bto_hopper_block_slot(self.h, int(clkn), n, ch, ptrs, lens, int(low_channel), int(high_channel), log, 1 << 16)
        return log.value.decode()

    @property
    def piconet(self):
        return lib().bto_hopper_block_piconet(self.h).contents


class Sniffer:
    """The handler half of multi_sniffer_impl (ac / id / discover / recall / decode / fhs,
    lib/multi_sniffer_impl.cc:169-365): feed classic hits in the order work() reports them, get the
    text the reference prints."""

    def __init__(self, tun=False):
        self.h = lib().bto_sniffer_new()
        if tun:
            lib().bto_sniffer_set_tun(self.h, 1)

    def tap(self):
        """the frames written to the TAP device so far (each after its uint32 LE length)"""
        n = lib().bto_sniffer_tap(self.h, None, 0)
        buf = ctypes.create_string_buffer(max(n, 1))
        lib().bto_sniffer_tap(self.h, buf, n)
        return buf.raw[:n]

    def __del__(self):
        try:
            if self.h:
                lib().bto_sniffer_free(self.h)
                self.h = None
        except Exception:
            pass

    def ac(self, symbols, clkn, channel, snr):
        s = np.ascontiguousarray(symbols, dtype=np.uint8)
        log = ctypes.create_string_buffer(1 << 16)
        lib().bto_sniffer_ac(self.h, s.tobytes() + bytes(64), len(s), int(clkn), int(channel), float(snr), log, 1 << 16)
        return log.value.decode()


def try_clock(symbols, clock):
    """classic_packet::try_clock on symbols that start at the access code -> (uap, type, fec13_ok)"""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    t, u = ctypes.c_int(-1), ctypes.c_int(-1)
    r = lib().bto_try_clock(s.tobytes() + bytes(64), int(clock), ctypes.byref(t), ctypes.byref(u))
    return r, t.value, t.value >= 0


def crc_check(symbols, clock, ptype, uap):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    pad = np.concatenate([s[:3125], np.zeros(3200 - min(len(s), 3125), np.uint8)])
    return lib().bto_crc_check(pad.tobytes(), min(len(s), 3125), int(clock), int(ptype), int(uap))


def le_print(symbols, freq):
    """le_packet::print() for symbols that start at the LE preamble (lib/packet_impl.cc:1529-1664)."""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    buf = ctypes.create_string_buffer(4096)
    lib().bto_le_print(s.tobytes() + bytes(8), len(s), float(freq), buf, 4096)
    return buf.value.decode()


def header_present(symbols, length=None):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    n = len(s) if length is None else int(length)
    pad = np.concatenate([s, np.zeros(140, np.uint8)])
    return bool(lib().bto_header_present(pad.tobytes(), n))


def lut(name):
    """A regenerated table of the oracle by the reference's name (bt_oracle.c / bt_uap.c), as bytes."""
    L = lib()
    buf = ctypes.create_string_buffer(1024)
    for f in (L.bto_lut, L.bto_uap_lut):
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    if name == "classic_packet::INDICES":
        n = L.bto_uap_lut(name.encode(), buf, 1024)
    else:
        n = L.bto_lut(name.encode(), buf, 1024)
    if n < 0:
        raise KeyError(name)
    return buf.raw[:n]


def qualifying_offsets(symbols):
    """Every offset of a symbol stream at which classic_packet::sniff_ac would accept (no resume)."""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    raw = s.tobytes()
    L = lib()
    out, pos, n = [], 0, len(s)
    while pos + 68 <= n:
        i = L.bto_sniff_ac(raw[pos:], n - pos - 67)
        if i < 0:
            break
        out.append(pos + i)
        pos += i + 1
    return out


def scan_symbols(symbols, max_hits=4096):
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    hits = (Hit * max_hits)()
    n = lib().bto_scan_symbols(s.tobytes(), len(s), hits, max_hits)
    return [(hits[i].offset, hits[i].lap, hits[i].ac_errors) for i in range(n)]


class Oracle:
    """One reference block instance (multi_LAP or multi_sniffer)."""

    def __init__(self, sample_rate, center_freq, squelch_db=10.0, mode=MODE_SNIFFER,
                 mm_policy=MM_WINDOWED_RESET, le=False, correlator=None):
        self.L = lib()
        self.h = self.L.bto_create(sample_rate, center_freq, squelch_db, mode)
        if not self.h:
            raise MemoryError("bto_create")
        self.L.bto_set_mm_policy(self.h, mm_policy)
        self.L.bto_set_le(self.h, 1 if le else 0)
        if correlator is not None:              # default: libbtbb in LAP mode, in-tree in sniffer mode
            self.L.bto_set_correlator(self.h, int(correlator))
        g = lambda n: getattr(self.L, "bto_" + n)(self.h)
        self.history = g("history")
        self.slot = g("samples_per_slot")
        self.decim = g("decimation")
        self.low_ch = g("low_channel")
        self.high_ch = g("high_channel")
        self.first_ch = g("first_channel_sample")
        self.first_noise = g("first_noise_sample")
        self.ntaps_ch = g("ntaps_channel")
        self.ntaps_noise = g("ntaps_noise")
        self.ddc_out = g("ddc_out")
        self.noise_out = g("noise_out")

    def __del__(self):
        try:
            if self.h:
                self.L.bto_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def channel_taps(self):
        return np.ctypeslib.as_array(self.L.bto_channel_taps(self.h), (self.ntaps_ch,)).copy()

    def noise_taps(self):
        return np.ctypeslib.as_array(self.L.bto_noise_taps(self.h), (self.ntaps_noise,)).copy()

    def mmse_taps(self):
        return np.ctypeslib.as_array(self.L.bto_mmse_taps(self.h), (129, 8)).copy()

    def atan_table(self):
        return np.ctypeslib.as_array(self.L.bto_atan_table(self.h), (257,)).copy()

    def fast_atan2f(self, y, x):
        return self.L.bto_fast_atan2f(self.h, y, x)

    def window(self, iq, k):
        """Window of history() samples the reference sees at work() call k."""
        a = np.ascontiguousarray(iq).astype(np.complex64, copy=False)
        H, slot = self.history, self.slot
        a0 = k * slot - (H - 1)
        win = np.zeros(H, np.complex64)
        lo, hi = max(a0, 0), min(a0 + H, len(a))
        if hi > lo:
            win[lo - a0:hi - a0] = a[lo:hi]
        return win

    def channel_samples(self, win, channel):
        w = _iq_as_f32(win)
        out = np.zeros(2 * (self.ddc_out + 1), np.float32)
        e = ctypes.c_double()
        n = self.L.bto_channel_samples(self.h, channel, _fp(w), _fp(out), ctypes.byref(e))
        return out[:2 * n].view(np.complex64).copy(), e.value

    def check_snr(self, win, channel, on_energy):
        w = _iq_as_f32(win)
        snr, off = ctypes.c_double(), ctypes.c_double()
        ok = self.L.bto_check_snr(self.h, channel, on_energy, _fp(w), ctypes.byref(snr),
                                  ctypes.byref(off))
        return bool(ok), snr.value, off.value

    def demod(self, ch_iq):
        a = _iq_as_f32(ch_iq)
        n = len(a) // 2 - 1
        out = np.zeros(max(n, 1), np.float32)
        self.L.bto_demod(self.h, _fp(a), _fp(out), n)
        return out[:n]

    def channel_symbols(self, ch_iq):
        a = _iq_as_f32(ch_iq)
        n = len(a) // 2
        sym = ctypes.create_string_buffer(n + 64)
        soft = np.zeros(n + 8, np.float32)
        k = self.L.bto_channel_symbols(self.h, _fp(a), n, sym, _fp(soft))
        return np.frombuffer(sym.raw[:k], dtype=np.uint8).copy(), soft[:k].copy()

    def work(self, win, slot, max_hits=256):
        w = _iq_as_f32(win)
        hits = (Hit * max_hits)()
        n = self.L.bto_work(self.h, _fp(w), slot, hits, max_hits)
        return [hits[i] for i in range(n)]

    def run_stream(self, iq, max_hits=65536, threads=0):
        a = _iq_as_f32(iq)
        hits = (Hit * max_hits)()
        done = ctypes.c_int()
        if threads and threads > 0:
            n = self.L.bto_run_stream_mt(self.h, _fp(a), len(a) // 2, hits, max_hits,
                                         ctypes.byref(done), threads)
        else:
            n = self.L.bto_run_stream(self.h, _fp(a), len(a) // 2, hits, max_hits,
                                      ctypes.byref(done))
        return [hits[i] for i in range(n)], done.value
