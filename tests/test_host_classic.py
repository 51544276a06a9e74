"""Host packet parsers (gr-bluetooth_amd/host/classic.cc: classic_packet::crc_check / decode / print -- the
product's restatement of lib/packet_impl.cc:612-1202) against the oracle's (oracle/bt_uap.c), differentially:
valid packets of the types the generator can build, and random symbols through every packet type."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gr-bluetooth_amd", "libgnuradio-bluetooth-amd.so")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gr-bluetooth_amd", "host")])
    import torch  # noqa: F401  (libbtgpu's HIP runtime: the one torch bundles)
    L = ctypes.CDLL(LIB)
    L.bt_host_crc_check.restype = ctypes.c_int
    L.bt_host_crc_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.bt_host_decode_print.restype = ctypes.c_int
    L.bt_host_decode_print.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int,
                                       ctypes.c_char_p, ctypes.c_int]
    return L


def _host_crc(L, s, clock, ptype, uap):
    s = np.ascontiguousarray(s, np.uint8)
    return L.bt_host_crc_check(s.tobytes(), len(s), clock, ptype, uap)


def _host_decode(L, s, uap, clock, have27):
    s = np.ascontiguousarray(s, np.uint8)
    buf = ctypes.create_string_buffer(4096)
    got = L.bt_host_decode_print(s.tobytes(), len(s), uap, clock, have27, buf, 4096)
    return got, buf.value.decode()


def _oracle_decode(po, s, uap, clock, have27):
    L = po.lib()
    L.bto_packet_new.restype = ctypes.c_void_p
    L.bto_packet_new.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int]
    L.bto_packet_free.argtypes = [ctypes.c_void_p]
    L.bto_packet_decode_print.restype = ctypes.c_int
    L.bto_packet_decode_print.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p,
                                          ctypes.c_size_t]
    s = np.ascontiguousarray(s, np.uint8)
    p = L.bto_packet_new(s.tobytes() + bytes(64), len(s), 0, 0)
    buf = ctypes.create_string_buffer(4096)
    got = L.bto_packet_decode_print(p, uap, clock, have27, buf, 4096)
    L.bto_packet_free(p)
    return got, buf.value.decode()


def test_valid_packets_decode_like_the_oracle(host, po, synth):
    rng = np.random.default_rng(12)
    lap, uap = 0x4831DD, 0x6B
    seen = set()
    for trial in range(60):
        clk = int(rng.integers(0, 1 << 27))
        kind = trial % 5
        body = bytes(rng.integers(0, 256, int(rng.integers(1, 17)), dtype=np.uint8))
        if kind == 0:
            bits = synth.classic_dh1_bits(lap, uap, clk, body)
        elif kind == 1:
            bits = synth.classic_dm1_bits(lap, uap, clk, body)
        elif kind == 2:
            bits = synth.classic_fhs_bits(lap, uap, clk, 0xABCDEF, 0x47, 0x1234, int(rng.integers(0, 1 << 26)))
        else:
            bits = synth.classic_poll_bits(lap, uap, clk, ptype=kind - 3)
        s = np.concatenate([bits, rng.integers(0, 2, 400, dtype=np.uint8)])       # noise after the packet
        for c in (clk, clk ^ 1, clk + 17):                                        # right and wrong clocks
            u, t, ok = po.try_clock(s, c & 63)
            assert _host_crc(host, s, c & 63, t, u) == po.crc_check(s, c & 63, t, u)
            for have27 in (0, 1):
                assert _host_decode(host, s, uap, c, have27) == _oracle_decode(po, s, uap, c, have27)
        got, text = _host_decode(host, s, uap, clk, 1)
        assert got == 1
        seen.add(text.split("\n")[0])
        if kind == 0:
            assert text == "DH1/2-DH1\n  LLID: 2\n  flow: 1\n  payload length: %d\n" % (len(body) + 3)
        if kind == 1:
            assert text.startswith("DM1\n  LLID: 2\n")
    assert seen == {"DH1/2-DH1", "DM1", "FHS", "NULL", "POLL"}


def test_random_symbols_through_every_packet_type(host, po):
    """Every branch of crc_check (FHS, DM1/3/5, DV, DH1/3/5, EV3/4/5, HV1, and the types it ignores) on
    random symbol streams of random lengths: same verdict from the host parsers and the oracle."""
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.choice([100, 130, 200, 366, 500, 1200, 3125, 3500]))
        s = rng.integers(0, 2, n, dtype=np.uint8)
        if trial % 3 == 0 and n > 130:         # FEC-friendly streams reach deeper into the DM / EV4 / HV parsers
            s[126:] = np.repeat(rng.integers(0, 2, (n - 126 + 2) // 3, dtype=np.uint8), 3)[:n - 126]
        ptype, clock, uap = trial % 16, int(rng.integers(0, 64)), int(rng.integers(0, 256))
        assert _host_crc(host, s, clock, ptype, uap) == po.crc_check(s, clock, ptype, uap), (trial, ptype, n)
