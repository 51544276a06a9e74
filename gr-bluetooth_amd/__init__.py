"""gr_bluetooth_amd -- Python host mirror of the reference's block surface over libbtgpu.so.

The directory is named ``gr-bluetooth_amd`` (not importable by name); load it with
``tests/conftest.py``'s helper or::

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location(
        "gr_bluetooth_amd", "<repo>/gr-bluetooth_amd/__init__.py",
        submodule_search_locations=["<repo>/gr-bluetooth_amd"])
    mod = importlib.util.module_from_spec(spec); sys.modules["gr_bluetooth_amd"] = mod
    spec.loader.exec_module(mod)

Mirrors ``gr_bluetooth.multi_LAP(sample_rate, center_freq, squelch_threshold)`` and
``gr_bluetooth.multi_sniffer(sample_rate, center_freq, squelch_threshold, tun)``
(reference swig/gr_bluetooth.i:35-45, include/gr_bluetooth/multi_LAP.h:53,
multi_sniffer.h:54).  There is NO CPU fallback: if libbtgpu.so is missing or no gfx950
device is visible, constructing a block raises.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbtgpu.so")

OK, EINVAL, ENOMEM, EDEVICE, ENODEVICE, EOVERFLOW, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
MODE_LAP, MODE_SNIFFER = 0, 1
CHANNELIZER_AUTO, CHANNELIZER_DIRECT, CHANNELIZER_POLYPHASE = 0, 1, 2
SQUELCH_AUTO, SQUELCH_DIRECT, SQUELCH_STAGED = 0, 1, 2
CORRELATOR_AUTO, CORRELATOR_INTREE, CORRELATOR_BTBB = 0, 1, 2   # multi_LAP default: BTBB (libbtbb, as the reference)
FLAG_LE, FLAG_DEBUG_Y, FLAG_ASYNC, FLAG_SYMBOLS, FLAG_HEADERS, FLAG_TIMING, FLAG_NO_NSYM, FLAG_TIMING_BANK = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_NO_VERIFY = 256      # polyphase path without the exact confirmation of its records (A/B)
FLAG_EXACT_PAYLOAD = 512  # exported symbols exact to the end of the packet (include/btgpu.h)
FLAG_EXACT_ALL = 1024     # no selection: every row of every channel exact (include/btgpu.h)
KIND_AC, KIND_AA = 0, 1


class Config(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_double), ("center_freq", ctypes.c_double),
                ("squelch_db", ctypes.c_double), ("mode", ctypes.c_int32),
                ("device", ctypes.c_int32), ("channelizer", ctypes.c_int32),
                ("squelch", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("max_batch_slots", ctypes.c_int32), ("max_hits", ctypes.c_int32),
                ("correlator", ctypes.c_int32)]


class Design(ctypes.Structure):
    _fields_ = [("samples_per_symbol", ctypes.c_double), ("samples_per_slot", ctypes.c_int32),
                ("decimation", ctypes.c_int32), ("ntaps_channel", ctypes.c_int32),
                ("ntaps_noise", ctypes.c_int32), ("low_channel", ctypes.c_int32),
                ("high_channel", ctypes.c_int32), ("first_channel_sample", ctypes.c_int32),
                ("first_noise_sample", ctypes.c_int32), ("history", ctypes.c_int32),
                ("ddc_out", ctypes.c_int32), ("noise_out", ctypes.c_int32),
                ("channelizer", ctypes.c_int32), ("squelch", ctypes.c_int32),
                ("left_margin", ctypes.c_int32), ("correlator", ctypes.c_int32)]


class Hit(ctypes.Structure):
    _fields_ = [("slot", ctypes.c_uint64), ("channel", ctypes.c_int32), ("offset", ctypes.c_int32),
                ("lap", ctypes.c_uint32), ("ac_errors", ctypes.c_int32), ("kind", ctypes.c_int32),
                ("nsym", ctypes.c_int32), ("snr_db", ctypes.c_double)]

    def key(self):
        return (self.slot, self.channel, self.kind, self.offset, self.lap, self.ac_errors, self.nsym)


K_DDC_CHANNEL, K_DEMOD_ENERGY, K_DDC_NOISE, K_NOISE_ENERGY, K_WINDOW = 0, 1, 2, 3, 4
KERNEL_NAMES = ["ddc_channel", "demod_energy", "ddc_noise", "noise_energy", "window", "finish", "verify", "exact"]


class Timing(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_float * 8), ("kernel_launches", ctypes.c_uint32 * 8),
                ("total_ms", ctypes.c_float), ("batches", ctypes.c_uint32),
                ("samples", ctypes.c_uint64), ("slots", ctypes.c_uint64),
                ("verify_windows", ctypes.c_uint64), ("verify_rows", ctypes.c_uint64), ("verify_turned_away", ctypes.c_uint64),
                ("long_tasks", ctypes.c_uint64), ("long_rows", ctypes.c_uint64), ("long_turned_away", ctypes.c_uint64)]


EXPORTS = ["btgpu_design_query", "btgpu_acgen", "btgpu_filter_taps", "btgpu_strerror",
           "btgpu_version", "btgpu_create", "btgpu_destroy", "btgpu_get_design", "btgpu_history",
           "btgpu_last_error", "btgpu_work", "btgpu_push", "btgpu_process_device", "btgpu_poll",
           "btgpu_poll_symbols", "btgpu_poll_headers", "btgpu_hopseq_create", "btgpu_hopseq_destroy",
           "btgpu_hopseq_init_candidates", "btgpu_hopseq_winnow", "btgpu_hopseq_candidates", "btgpu_hopseq_lookup",
           "btgpu_hopseq_fetch", "btgpu_pending", "btgpu_flush", "btgpu_device", "btgpu_device_count", "btgpu_last_timing", "btgpu_debug_fetch",
           "btgpu_debug_scan_symbols", "btgpu_debug_lut", "btgpu_process_host"]


class BtgpuError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = "btgpu error %d" % code
        try:
            msg += ": " + lib().btgpu_strerror(code).decode()
        except Exception:
            pass
        if detail:
            msg += " (" + detail + ")"
        super().__init__(msg)


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libbtgpu.so, in-tree."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("btgpu.hip", "kernels.hip.h", "verify.hip.h", "design.cc", "design.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "btgpu.h"))
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", src_dir] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    """Load libbtgpu.so; raises (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError("libbtgpu.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C gr-bluetooth_amd/csrc` (no CPU fallback exists)")
    try:
        # The PyTorch ROCm wheel bundles its own libamdhip64/libhsa-runtime64.  Two HIP runtimes
        # in one process cannot both own the GPU, so when torch is installed it is imported
        # first and libbtgpu.so then binds to the runtime torch already loaded (same SONAME).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(_SO)
    vp = ctypes.c_void_p
    L.btgpu_design_query.restype = ctypes.c_int
    L.btgpu_design_query.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(Design)]
    L.btgpu_acgen.restype = ctypes.c_int
    L.btgpu_acgen.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8)]
    L.btgpu_filter_taps.restype = ctypes.c_int
    L.btgpu_filter_taps.argtypes = [ctypes.POINTER(Config), ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    L.btgpu_strerror.restype = ctypes.c_char_p
    L.btgpu_strerror.argtypes = [ctypes.c_int]
    L.btgpu_version.restype = ctypes.c_char_p
    L.btgpu_create.restype = ctypes.c_int
    L.btgpu_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.btgpu_destroy.argtypes = [vp]
    L.btgpu_get_design.restype = ctypes.c_int
    L.btgpu_get_design.argtypes = [vp, ctypes.POINTER(Design)]
    L.btgpu_history.restype = ctypes.c_int
    L.btgpu_history.argtypes = [vp]
    L.btgpu_last_error.restype = ctypes.c_char_p
    L.btgpu_last_error.argtypes = [vp]
    L.btgpu_work.restype = ctypes.c_int
    L.btgpu_work.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.btgpu_push.restype = ctypes.c_int
    L.btgpu_push.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
    L.btgpu_process_device.restype = ctypes.c_int
    L.btgpu_process_device.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.btgpu_process_host.restype = ctypes.c_int
    L.btgpu_process_host.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
    L.btgpu_poll.restype = ctypes.c_int
    L.btgpu_poll.argtypes = [vp, ctypes.POINTER(Hit), ctypes.c_int]
    L.btgpu_pending.restype = ctypes.c_int
    L.btgpu_pending.argtypes = [vp]
    L.btgpu_poll_headers.restype = ctypes.c_int
    L.btgpu_poll_headers.argtypes = [vp, ctypes.POINTER(Hit), ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.btgpu_hopseq_create.restype = ctypes.c_int
    L.btgpu_hopseq_create.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.btgpu_hopseq_destroy.restype = None
    L.btgpu_hopseq_destroy.argtypes = [vp]
    L.btgpu_hopseq_init_candidates.restype = ctypes.c_int
    L.btgpu_hopseq_init_candidates.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.btgpu_hopseq_winnow.restype = ctypes.c_int
    L.btgpu_hopseq_winnow.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.btgpu_hopseq_candidates.restype = ctypes.c_int
    L.btgpu_hopseq_candidates.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    L.btgpu_hopseq_lookup.restype = ctypes.c_int
    L.btgpu_hopseq_lookup.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]
    L.btgpu_hopseq_fetch.restype = ctypes.c_long
    L.btgpu_hopseq_fetch.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint8)]
    L.btgpu_poll_symbols.restype = ctypes.c_int
    L.btgpu_poll_symbols.argtypes = [vp, ctypes.POINTER(Hit), ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.btgpu_flush.restype = ctypes.c_int
    L.btgpu_flush.argtypes = [vp]
    L.btgpu_last_timing.restype = ctypes.c_int
    L.btgpu_last_timing.argtypes = [vp, ctypes.POINTER(Timing)]
    L.btgpu_debug_fetch.restype = ctypes.c_long
    L.btgpu_debug_fetch.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.btgpu_debug_tables.restype = ctypes.c_int
    L.btgpu_debug_tables.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(ctypes.c_float),
                                     ctypes.POINTER(ctypes.c_float), ctypes.c_uint32,
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.btgpu_debug_scan_symbols.restype = ctypes.c_long
    L.btgpu_debug_scan_symbols.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(Hit), ctypes.c_long]
    L.btgpu_debug_lut.restype = ctypes.c_int
    L.btgpu_debug_lut.argtypes = [ctypes.c_char_p, vp, ctypes.c_int]
    _lib = L
    return L


def make_config(sample_rate, center_freq, squelch_db=10.0, mode=MODE_SNIFFER, device=-1,
                channelizer=CHANNELIZER_AUTO, squelch=SQUELCH_AUTO, flags=0, max_batch_slots=0,
                max_hits=0, correlator=CORRELATOR_AUTO):
    return Config(float(sample_rate), float(center_freq), float(squelch_db), mode, device,
                  channelizer, squelch, flags, max_batch_slots, max_hits, correlator)


def design_query(sample_rate, center_freq, squelch_db=10.0, mode=MODE_SNIFFER, **kw):
    """multi_block constructor arithmetic on the host (no GPU needed)."""
    cfg = make_config(sample_rate, center_freq, squelch_db, mode, **kw)
    d = Design()
    rc = lib().btgpu_design_query(ctypes.byref(cfg), ctypes.byref(d))
    if rc != OK:
        raise BtgpuError(rc)
    return d


def acgen(lap):
    ac = (ctypes.c_uint8 * 9)()
    rc = lib().btgpu_acgen(lap, ac)
    if rc != OK:
        raise BtgpuError(rc)
    return bytes(ac)


def filter_taps(sample_rate, which):
    cfg = make_config(sample_rate, 2441e6)
    n = lib().btgpu_filter_taps(ctypes.byref(cfg), which, None, 0)
    out = np.zeros(n, np.float32)
    lib().btgpu_filter_taps(ctypes.byref(cfg), which, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n)
    return out


def debug_tables(sample_rate, center_freq, lap=0):
    cfg = make_config(sample_rate, center_freq)
    mmse = np.zeros((129, 8), np.float32)
    atab = np.zeros(257, np.float32)
    lo, hi = ctypes.c_uint64(), ctypes.c_uint32()
    rc = lib().btgpu_debug_tables(ctypes.byref(cfg), mmse.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  atab.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), lap,
                                  ctypes.byref(lo), ctypes.byref(hi))
    if rc != OK:
        raise BtgpuError(rc)
    return mmse, atab, lo.value, hi.value


def debug_lut(name):
    """One of the product's regenerated integer tables as bytes (see btgpu_debug_lut)."""
    buf = ctypes.create_string_buffer(1024)
    n = lib().btgpu_debug_lut(name.encode(), buf, 1024)
    if n < 0:
        raise BtgpuError(n, name)
    return buf.raw[:n]


def scan_symbols(symbols, policy=1, device=-1, cap=1 << 16):
    """The window kernel's access-code search straight on a symbol stream (one 0/1 symbol per byte,
    the format of the reference's samples/channel37.dem): list of (offset, lap, ac_errors).
    policy 1 = stream scan resuming 68 symbols after a hit, 0 = every qualifying offset."""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    out = (Hit * cap)()
    n = lib().btgpu_debug_scan_symbols(s.tobytes(), len(s), device, policy, out, cap)
    if n < 0:
        raise BtgpuError(int(n))
    if n > cap:
        raise BtgpuError(EOVERFLOW, "%d records, cap %d" % (n, cap))
    return [(out[i].offset, out[i].lap, out[i].ac_errors) for i in range(n)]


def staged_design(sample_rate, center_freq):
    """Host-side parameters of the staged squelch filter (composite fit error, sizes)."""
    cfg = make_config(sample_rate, center_freq)
    err, ws = ctypes.c_double(), ctypes.c_double()
    R, L1, L3, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    f = lib().btgpu_debug_staged_design
    f.restype = ctypes.c_int
    rc = f(ctypes.byref(cfg), ctypes.byref(err), ctypes.byref(R), ctypes.byref(L1), ctypes.byref(L3),
           ctypes.byref(nw), ctypes.byref(ws))
    if rc != OK:
        raise BtgpuError(rc)
    return dict(fit_l1_error=err.value, R=R.value, L1=L1.value, L3=L3.value, nw=nw.value, weight_sum=ws.value)


def _as_f32(iq):
    a = np.ascontiguousarray(iq)
    if a.dtype == np.complex64:
        a = a.view(np.float32)
    if a.dtype != np.float32:
        raise TypeError("IQ must be complex64 or interleaved float32")
    return a


class _MultiBlock:
    """Common part of the block mirrors (reference: class multi_block)."""
    MODE = MODE_SNIFFER
    NAME = "bluetooth multi block"

    def __init__(self, sample_rate, center_freq, squelch_threshold, **kw):
        self._L = lib()
        self._h = ctypes.c_void_p()
        cfg = make_config(sample_rate, center_freq, squelch_threshold, self.MODE, **kw)
        rc = self._L.btgpu_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != OK:
            self._h = ctypes.c_void_p()
            raise BtgpuError(rc, "btgpu_create")
        self.design = Design()
        self._L.btgpu_get_design(self._h, ctypes.byref(self.design))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.btgpu_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # gr::block surface
    def name(self):
        return self.NAME

    def history(self):
        return self._L.btgpu_history(self._h)

    def output_multiple(self):
        return self.design.samples_per_slot

    def _check(self, rc, what):
        if rc not in (OK,):
            raise BtgpuError(rc, what + ": " + self._L.btgpu_last_error(self._h).decode())

    def work(self, input_items):
        """work(): input_items = history()-1 old samples followed by the new ones.
        Returns the number of items consumed (a whole number of slots)."""
        a = _as_f32(input_items)
        consumed = ctypes.c_size_t()
        rc = self._L.btgpu_work(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(a) // 2,
                                ctypes.byref(consumed))
        self._check(rc, "btgpu_work")
        return consumed.value

    def push(self, iq):
        a = _as_f32(iq)
        rc = self._L.btgpu_push(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(a) // 2)
        self._check(rc, "btgpu_push")

    def process_device(self, dev_ptr, n_complex, first_slot, n_slots, left_margin=0, stream=None):
        """dev_ptr[left_margin] = absolute sample first_slot*slot-(history()-1)."""
        rc = self._L.btgpu_process_device(self._h, ctypes.c_void_p(dev_ptr), n_complex, left_margin,
                                          first_slot, n_slots, ctypes.c_void_p(stream or 0))
        self._check(rc, "btgpu_process_device")

    def process_host(self, iq, first_slot, n_slots, left_margin=0):
        """btgpu_process_host: the segment lies in HOST memory (a numpy array, or the address of page-locked memory as an int
        together with its length in complex samples: (ptr, n_complex)); iq[left_margin] = absolute sample
        first_slot*slot-(history()-1)."""
        if isinstance(iq, tuple):
            ptr, n = ctypes.cast(ctypes.c_void_p(int(iq[0])), ctypes.POINTER(ctypes.c_float)), int(iq[1])
        else:
            a = _as_f32(iq)
            ptr, n = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(a) // 2
        rc = self._L.btgpu_process_host(self._h, ptr, n, left_margin, first_slot, n_slots)
        self._check(rc, "btgpu_process_host")

    def flush(self):
        rc = self._L.btgpu_flush(self._h)
        if rc not in (OK, EOVERFLOW):
            raise BtgpuError(rc, "btgpu_flush")

    def poll(self, max_hits=1 << 16):
        buf = (Hit * max_hits)()
        n = self._L.btgpu_poll(self._h, buf, max_hits)
        if n < 0:
            raise BtgpuError(n, "btgpu_poll")
        return [buf[i] for i in range(n)]

    HIT_DTYPE = np.dtype([("slot", "<u8"), ("channel", "<i4"), ("offset", "<i4"), ("lap", "<u4"),
                          ("ac_errors", "<i4"), ("kind", "<i4"), ("nsym", "<i4"), ("snr_db", "<f8")])

    def poll_arrays(self, max_hits=1 << 20):
        """Drain the hit queue into a numpy structured array (no per-record Python objects)."""
        n = self._L.btgpu_pending(self._h)
        n = min(max(n, 0), max_hits)
        buf = np.zeros(n, self.HIT_DTYPE)
        if n:
            got = self._L.btgpu_poll(self._h, buf.ctypes.data_as(ctypes.POINTER(Hit)), n)
            if got < 0:
                raise BtgpuError(got, "btgpu_poll")
            buf = buf[:got]
        return buf

    def poll_symbols(self, sym_cap=3125, max_hits=1 << 16):
        """Hits plus the symbols the reference hands to ac()/aa() with each of them (needs
        flags=FLAG_SYMBOLS): returns (structured hit array, uint8 [n, sym_cap], int32 [n])."""
        n = min(max(self._L.btgpu_pending(self._h), 0), max_hits)
        hits = np.zeros(n, self.HIT_DTYPE)
        syms = np.zeros((n, sym_cap), np.uint8)
        lens = np.zeros(n, np.int32)
        if n:
            got = self._L.btgpu_poll_symbols(self._h, hits.ctypes.data_as(ctypes.POINTER(Hit)),
                                             syms.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), sym_cap,
                                             lens.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n)
            if got < 0:
                raise BtgpuError(got, "btgpu_poll_symbols")
            hits, syms, lens = hits[:got], syms[:got], lens[:got]
        return hits, syms, lens

    HEADER_DTYPE = np.dtype([("uap", np.uint8, 64), ("type", np.uint8, 64), ("fec13_ok", "<i4"), ("reserved", "<i4")])

    def poll_headers(self, sym_cap=3125, max_hits=1 << 16):
        """poll_symbols plus, per hit, what classic_packet::try_clock gives for the 64 CLK1-6
        candidates (needs flags=FLAG_HEADERS): (hits, headers [n] of HEADER_DTYPE, symbols, lens)."""
        n = min(max(self._L.btgpu_pending(self._h), 0), max_hits)
        hits = np.zeros(n, self.HIT_DTYPE)
        hdrs = np.zeros(n, self.HEADER_DTYPE)
        syms = np.zeros((n, sym_cap), np.uint8)
        lens = np.zeros(n, np.int32)
        if n:
            got = self._L.btgpu_poll_headers(self._h, hits.ctypes.data_as(ctypes.POINTER(Hit)), hdrs.ctypes.data_as(ctypes.c_void_p),
                                             syms.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), sym_cap,
                                             lens.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n)
            if got < 0:
                raise BtgpuError(got, "btgpu_poll_headers")
            hits, hdrs, syms, lens = hits[:got], hdrs[:got], syms[:got], lens[:got]
        return hits, hdrs, syms, lens

    def timing(self):
        t = Timing()
        self._L.btgpu_last_timing(self._h, ctypes.byref(t))
        return t

    def debug_fetch(self, what, channel=0, first=0, count=0):
        dt = {0: np.complex64, 1: np.float32, 2: np.float64, 3: np.float64, 4: np.float64, 5: np.complex64,
              6: np.float32, 7: np.int32, 9: np.uint64, 11: np.uint32,
              10: np.dtype([('w', '<i4'), ('rows', '<i4'), ('snr', '<f8'), ('emit_from', '<i4'), ('pad', '<i4')]),
              8: np.dtype([('w', '<i4'), ('ii', '<u4'), ('oo', '<i4'), ('mu', '<f4'), ('omega', '<f4'), ('last', '<f4')])}[what]
        out = np.zeros(count, dt)
        n = self._L.btgpu_debug_fetch(self._h, what, channel, first, count, out.ctypes.data_as(ctypes.c_void_p))
        if n < 0:
            raise BtgpuError(int(n), "btgpu_debug_fetch")
        return out[:n]

    def format_hit(self, h):
        raise NotImplementedError


class multi_LAP(_MultiBlock):
    """gr_bluetooth.multi_LAP(sample_rate, center_freq, squelch_threshold)
    (reference lib/multi_LAP_impl.cc:39-56)."""
    MODE = MODE_LAP
    NAME = "bluetooth multi LAP block"

    def __init__(self, sample_rate, center_freq, squelch_threshold, **kw):
        super().__init__(sample_rate, center_freq, squelch_threshold, **kw)

    def format_hit(self, h):
        # lib/multi_LAP_impl.cc:97-100
        return "GOT PACKET: ch=%d, LAP=%06x, err=%u at time slot %d" % (h.channel, h.lap, h.ac_errors, h.slot)


class multi_sniffer(_MultiBlock):
    """gr_bluetooth.multi_sniffer(sample_rate, center_freq, squelch_threshold, tun)
    (reference lib/multi_sniffer_impl.cc:42-72).  `tun` is accepted for signature
    compatibility; the TAP sink is outside the hot path."""
    MODE = MODE_SNIFFER
    NAME = "bluetooth multi sniffer block"

    def __init__(self, sample_rate, center_freq, squelch_threshold, tun=False, le=False, **kw):
        """le=True adds the le_packet::sniff_aa pass the reference's work() always runs after the
        classic one (lib/multi_sniffer_impl.cc:129-149); the C++ block mirror enables it."""
        self.tun = bool(tun)
        if le:
            kw["flags"] = kw.get("flags", 0) | FLAG_LE
        super().__init__(sample_rate, center_freq, squelch_threshold, **kw)

    def format_hit(self, h):
        # lib/multi_sniffer_impl.cc:177-178 (prefix printed by ac() before the handlers)
        return "time %6d, snr=%.1f, channel %2d, LAP %06x " % (h.slot & 0x7ffffff, h.snr_db, h.channel, h.lap)


class HopSequence:
    """The complete hopping sequence of one piconet on the GPU and its CLK1-27 candidate list
    (basic_rate_piconet hop reversal, lib/piconet_impl.cc:96-338)."""
    LENGTH = 1 << 27

    def __init__(self, address, afh=False, device=-1):
        import torch  # noqa: F401  (one HIP runtime per process, see lib())
        self._L = lib()
        h = ctypes.c_void_p()
        rc = self._L.btgpu_hopseq_create(int(address) & 0xFFFFFFF, 1 if afh else 0, device, ctypes.byref(h))
        if rc != OK:
            raise BtgpuError(rc, "btgpu_hopseq_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.btgpu_hopseq_destroy(self._h)
            self._h = None

    __del__ = close

    def init_candidates(self, channel, known_clock_bits, aliased=False):
        n = self._L.btgpu_hopseq_init_candidates(self._h, int(channel), int(known_clock_bits), 1 if aliased else 0)
        if n < 0:
            raise BtgpuError(n, "btgpu_hopseq_init_candidates")
        return n

    def winnow(self, offset, channel, aliased=False):
        n = self._L.btgpu_hopseq_winnow(self._h, int(offset), int(channel), 1 if aliased else 0)
        if n < 0:
            raise BtgpuError(n, "btgpu_hopseq_winnow")
        return n

    def candidates(self, cap=1 << 22):
        buf = np.zeros(cap, np.uint32)
        n = self._L.btgpu_hopseq_candidates(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap)
        if n < 0:
            raise BtgpuError(n, "btgpu_hopseq_candidates")
        return buf[:min(n, cap)]

    def lookup(self, index):
        idx = np.ascontiguousarray(index, dtype=np.uint32)
        out = np.zeros(len(idx), np.uint8)
        rc = self._L.btgpu_hopseq_lookup(self._h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(idx),
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        if rc != OK:
            raise BtgpuError(rc, "btgpu_hopseq_lookup")
        return out

    def fetch(self, first, count):
        out = np.zeros(count, np.uint8)
        n = self._L.btgpu_hopseq_fetch(self._h, first, count, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        if n < 0:
            raise BtgpuError(int(n), "btgpu_hopseq_fetch")
        return out[:n]
