"""Synthetic wideband Bluetooth captures (SURVEY.md section 8(d) "Synthetic inputs").

The reference's IQ captures (samples/headset*.cfile) are not in the checkout
(/root/reference/.MISSING_LARGE_BLOBS), so tests and bench.py drive the hot path
with captures generated here: AWGN floor + N piconets hopping pseudo-randomly
over the visible channels, each transmitting GFSK bursts (1 Msym/s, BT 0.5,
modulation index 0.32) that start with a valid 72-bit access code for the
piconet's LAP.  The generator is independent of both the oracle and the HIP
path: the access code comes from its own BCH(64,30) encoder below.

numpy version: tests (small captures).  torch version: bench.py (built on the
GPU so that a 1e8-sample capture does not cost minutes of host time).
"""
import math

import numpy as np

SYMBOL_RATE = 1e6
BASE_FREQUENCY = 2402e6
_PN = 0x83848D96BBCC54FC
_GEN = 0o260534236651          # BCH(64,30) generator, degree 34


def sync_word(lap):
    """64-bit Bluetooth sync word, bit i = i-th transmitted bit."""
    lap &= 0xFFFFFF
    barker = 0b110010 if (lap >> 23) & 1 else 0b001101   # a24..a29, a24 = MSB of this literal
    # information polynomial: LAP bits a0..a23 then Barker bits, placed at bit 34..63
    info = lap
    for i in range(6):
        info |= ((barker >> (5 - i)) & 1) << (24 + i)
    x = (info ^ (_PN >> 34)) & ((1 << 30) - 1)
    # parity = x(D) * D^34 mod g(D)
    rem = x << 34
    for bit in range(63, 33, -1):
        if (rem >> bit) & 1:
            rem ^= _GEN << (bit - 34)
    cw = (x << 34) | (rem & ((1 << 34) - 1))
    return cw ^ _PN


def access_code_bits(lap):
    """72 air-order bits: 4 preamble, 64 sync word, 4 trailer."""
    sw = sync_word(lap)
    sync = [(sw >> i) & 1 for i in range(64)]
    pre = [1, 0, 1, 0] if sync[0] else [0, 1, 0, 1]
    tr = [0, 1, 0, 1] if sync[63] else [1, 0, 1, 0]
    return np.array(pre + sync + tr, dtype=np.uint8)


def visible_channels(sample_rate, center_freq):
    """Channel range rule of multi_block::set_channels (lib/multi_block.cc:306-324)."""
    center = (center_freq - BASE_FREQUENCY) / 1e6
    bw = sample_rate / 1e6
    lo = int(center - bw / 2 + 0.45 + 1)
    hi = int(center + bw / 2 - 0.45)
    return max(lo, 0), min(hi, 78)


def gaussian_pulse(sps, bt=0.5, span=3):
    """Gaussian frequency pulse, unit area, `span` symbols long."""
    n = int(span * sps) | 1
    t = (np.arange(n) - (n - 1) / 2) / sps
    sigma = math.sqrt(math.log(2)) / (2 * math.pi * bt)
    g = np.exp(-t * t / (2 * sigma * sigma))
    return g / g.sum()


def gfsk_baseband(bits, sps, h=0.32, bt=0.5):
    """Complex baseband GFSK at `sps` samples per symbol (unit amplitude)."""
    nrz = np.repeat(2.0 * np.asarray(bits, dtype=np.float64) - 1.0, sps)
    f = np.convolve(nrz, gaussian_pulse(sps, bt), mode="same")
    phase = np.pi * h * np.cumsum(f) / sps
    return np.exp(1j * phase)


def packet_bits(lap, rng, payload_bits):
    hdr = np.repeat(rng.integers(0, 2, 18, dtype=np.uint8), 3)        # FEC 1/3 shaped header
    pay = rng.integers(0, 2, payload_bits, dtype=np.uint8)
    return np.concatenate([access_code_bits(lap), hdr, pay])


def le_whitening_bits(le_index, n):
    """First n bits of the LE whitening sequence for channel index le_index (position 0 = 1,
    positions 1..6 = index MSB first; x^7 + x^4 + 1)."""
    p = [1] + [(le_index >> (5 - i)) & 1 for i in range(6)]
    out = []
    for _ in range(n):
        o = p[6]
        out.append(o)
        p = [o, p[0], p[1], p[2], p[3] ^ o, p[4], p[5]]
    return np.array(out, dtype=np.uint8)


def le_advert_bits(le_index, rng, payload_bytes=12, pdu_type=0, aa=0x8E89BED6):
    """Air-order bits of an LE advertising-channel packet: preamble, access address, whitened
    (PDU header, payload); CRC bits are random (nothing on the hot path checks them)."""
    aab = [(aa >> i) & 1 for i in range(32)]
    pre = [0, 1, 0, 1, 0, 1, 0, 1] if aab[0] == 0 else [1, 0, 1, 0, 1, 0, 1, 0]
    hdr = [(pdu_type >> i) & 1 for i in range(8)] + [(payload_bytes >> i) & 1 for i in range(8)]
    body = np.array(hdr + list(rng.integers(0, 2, 8 * payload_bytes + 24)), dtype=np.uint8)
    body ^= le_whitening_bits(le_index, len(body))
    return np.concatenate([np.array(pre + aab, dtype=np.uint8), body])


# ---------------------------------------------------------------------------------------------
# A piconet that really hops: Bluetooth basic-rate hop selection (Core Vol 2 Part B 2.6), packet
# header with HEC, data whitening, DH1 payload with CRC.  Written from the specification, independent
# of the oracle and of the HIP path (which are tested against captures built with it).
# ---------------------------------------------------------------------------------------------
def _perm5(z, p_high, p_low):
    i1 = (0, 2, 1, 3, 0, 1, 0, 3, 1, 0, 2, 1, 0, 1)
    i2 = (1, 3, 2, 4, 4, 3, 2, 4, 4, 3, 4, 3, 3, 2)
    p = [(p_low >> i) & 1 for i in range(9)] + [(p_high >> i) & 1 for i in range(5)]
    zb = [(z >> i) & 1 for i in range(5)]
    for i in range(13, -1, -1):
        if p[i]:
            zb[i1[i]], zb[i2[i]] = zb[i2[i]], zb[i1[i]]
    return sum(b << i for i, b in enumerate(zb))


def hop_channel(address, slot_clock):
    """RF channel of the 79-hop system for master clock CLK27..1 = slot_clock (one value per
    625 us slot); address = (UAP << 24 | LAP) & 0xfffffff."""
    address &= 0xFFFFFFF
    clk = (slot_clock << 1) & 0xFFFFFFF                      # CLK27..0
    a1 = (address >> 23) & 0x1F
    b = (address >> 19) & 0x0F
    c1 = sum(((address >> (2 * i)) & 1) << i for i in range(5))
    d1 = (address >> 10) & 0x1FF
    e = sum(((address >> (2 * i + 1)) & 1) << i for i in range(7))
    x = (clk >> 2) & 0x1F
    y1 = (clk >> 1) & 1
    a = (a1 ^ (clk >> 21)) & 0x1F
    c = (c1 ^ (clk >> 16)) & 0x1F
    d = (d1 ^ (clk >> 7)) & 0x1FF
    f = (clk >> 3) & 0x1FFFFF0
    k = (_perm5(((x + a) % 32) ^ b, (y1 * 0x1F) ^ c, d) + e + f + 32 * y1) % 79
    return (2 * k) % 79                                      # register bank: even channels first


def whitening_bits(clk6, n, skip=0):
    """Data whitening sequence (x^7 + x^4 + 1) for CLK6..1 = clk6: register position 6 = 1,
    positions 0..5 = CLK1..CLK6."""
    p = [(clk6 >> i) & 1 for i in range(6)] + [1]
    out = []
    for _ in range(skip + n):
        o = p[6]
        out.append(o)
        p = [o, p[0], p[1], p[2], p[3] ^ o, p[4], p[5]]
    return np.array(out[skip:], dtype=np.uint8)


def _rev8(b):
    return int("{:08b}".format(b & 0xFF)[::-1], 2)


def _uap_of_hec(data, hec):
    for i in range(9, -1, -1):
        if hec & 0x80:
            hec ^= 0x65
        hec = ((hec << 1) | (((hec >> 7) ^ (data >> i)) & 1)) & 0xFF
    return _rev8(hec)


def _crc16(bits, uap):
    reg = (_rev8(uap) << 8) & 0xFF00
    for b in bits:
        reg = ((reg >> 1) | (((reg & 1) ^ int(b)) << 15)) & 0xFFFF
        reg ^= (reg & 0x8000) >> 5
        reg ^= (reg & 0x8000) >> 12
    return reg


def classic_poll_bits(lap, uap, slot_clock, lt_addr=1, flow=1, arqn=0, seqn=0, ptype=1):
    """Air-order bits of a POLL (or NULL, ptype 0) packet: access code + whitened FEC-1/3 header."""
    fields = (lt_addr & 7) | ((ptype & 15) << 3) | ((flow & 1) << 7) | ((arqn & 1) << 8) | ((seqn & 1) << 9)
    hec = next(h for h in range(256) if _uap_of_hec(fields, h) == (uap & 0xFF))
    header = np.array([(fields >> i) & 1 for i in range(10)] + [(hec >> i) & 1 for i in range(8)], np.uint8)
    header ^= whitening_bits(slot_clock & 0x3F, 18)
    return np.concatenate([access_code_bits(lap), np.repeat(header, 3)])


def fec23_encode(bits):
    """(15,10) shortened Hamming code of the basic-rate FEC 2/3: ten data bits, then the five parity
    bits of g(D) = (D + 1)(D^4 + D + 1) (register taps at positions 0, 1 and 3), zero padded to whole blocks."""
    bits = list(int(b) for b in bits)
    bits += [0] * ((-len(bits)) % 10)
    out = []
    for k in range(0, len(bits), 10):
        d = bits[k:k + 10]
        reg = [0] * 5
        for i in range(9, -1, -1):
            fb = d[i] ^ reg[4]
            reg = [fb, reg[0] ^ fb, reg[1], reg[2] ^ fb, reg[3]]
        out += d + reg
    return np.array(out, dtype=np.uint8)


def _classic_header(lap, uap, slot_clock, ptype, lt_addr, flow, arqn, seqn):
    fields = (lt_addr & 7) | ((ptype & 15) << 3) | ((flow & 1) << 7) | ((arqn & 1) << 8) | ((seqn & 1) << 9)
    hec = next(h for h in range(256) if _uap_of_hec(fields, h) == (uap & 0xFF))
    return np.array([(fields >> i) & 1 for i in range(10)] + [(hec >> i) & 1 for i in range(8)], np.uint8)


def classic_dm1_bits(lap, uap, slot_clock, body, lt_addr=1, llid=2, flow=1, arqn=0, seqn=0):
    """DM1: like DH1 (TYPE 3) with the whitened payload FEC-2/3 encoded."""
    body = bytes(body)
    assert len(body) <= 17
    ph = (llid & 3) | ((flow & 1) << 2) | (len(body) << 3)
    pay = [(ph >> i) & 1 for i in range(8)]
    for byte in body:
        pay += [(byte >> i) & 1 for i in range(8)]
    crc = _crc16(pay, uap)
    pay += [(crc >> i) & 1 for i in range(16)]
    wh = whitening_bits(slot_clock & 0x3F, 18 + len(pay))
    header = _classic_header(lap, uap, slot_clock, 3, lt_addr, flow, arqn, seqn) ^ wh[:18]
    return np.concatenate([access_code_bits(lap), np.repeat(header, 3), fec23_encode(np.array(pay, np.uint8) ^ wh[18:])])


def classic_fhs_bits(lap, uap, slot_clock, fhs_lap, fhs_uap, fhs_nap, fhs_clk27_2, lt_addr=0):
    """FHS (TYPE 2): 144 information bits (parity 34, LAP 24, EIR/SR/SP 6, UAP 8, NAP 16, class 24,
    LT_ADDR 3, CLK27-2 26, page scan mode 3) + CRC, whitened, FEC 2/3."""
    info = [0] * 144
    def put(at, value, n):
        for i in range(n):
            info[at + i] = (value >> i) & 1
    put(34, fhs_lap, 24); put(64, fhs_uap, 8); put(72, fhs_nap, 16); put(112, 1, 3); put(115, fhs_clk27_2, 26)
    crc = _crc16(info, uap)
    pay = np.array(info + [(crc >> i) & 1 for i in range(16)], np.uint8)
    wh = whitening_bits(slot_clock & 0x3F, 18 + len(pay))
    header = _classic_header(lap, uap, slot_clock, 2, lt_addr, 1, 0, 0) ^ wh[:18]
    return np.concatenate([access_code_bits(lap), np.repeat(header, 3), fec23_encode(pay ^ wh[18:])])


def classic_dh1_bits(lap, uap, slot_clock, body, lt_addr=1, llid=2, flow=1, arqn=0, seqn=0):
    """Air-order bits of a DH1 packet: 72-bit access code, FEC-1/3 header (LT_ADDR, TYPE 4, FLOW,
    ARQN, SEQN, HEC(UAP)), payload header + body + CRC(UAP); header and payload whitened with CLK6..1."""
    body = bytes(body)
    assert len(body) <= 27
    fields = (lt_addr & 7) | (4 << 3) | ((flow & 1) << 7) | ((arqn & 1) << 8) | ((seqn & 1) << 9)
    hec = next(h for h in range(256) if _uap_of_hec(fields, h) == (uap & 0xFF))
    header = [(fields >> i) & 1 for i in range(10)] + [(hec >> i) & 1 for i in range(8)]
    ph = (llid & 3) | ((flow & 1) << 2) | (len(body) << 3)
    pay = [(ph >> i) & 1 for i in range(8)]
    for byte in body:
        pay += [(byte >> i) & 1 for i in range(8)]
    crc = _crc16(pay, uap)
    pay += [(crc >> i) & 1 for i in range(16)]
    clk6 = slot_clock & 0x3F
    wh = whitening_bits(clk6, 18 + len(pay))
    header = np.array(header, np.uint8) ^ wh[:18]
    pay = np.array(pay, np.uint8) ^ wh[18:]
    return np.concatenate([access_code_bits(lap), np.repeat(header, 3), pay])


def classic_dh_multislot_bits(lap, uap, slot_clock, body, ptype=15, lt_addr=1, llid=2, flow=1, arqn=0, seqn=0):
    """DH3 (TYPE 11, <= 183 bytes) / DH5 (TYPE 15, <= 339 bytes): like DH1 with the two-byte payload header of the multi-slot
    packets (LLID 2, FLOW 1, LENGTH 10 bits, 3 reserved), body, CRC(UAP); header and payload whitened, no FEC."""
    body = bytes(body)
    assert ptype in (11, 15) and len(body) <= (183 if ptype == 11 else 339)
    ph = (llid & 3) | ((flow & 1) << 2) | (len(body) << 3)
    pay = [(ph >> i) & 1 for i in range(16)]
    for byte in body:
        pay += [(byte >> i) & 1 for i in range(8)]
    crc = _crc16(pay, uap)
    pay += [(crc >> i) & 1 for i in range(16)]
    wh = whitening_bits(slot_clock & 0x3F, 18 + len(pay))
    header = _classic_header(lap, uap, slot_clock, ptype, lt_addr, flow, arqn, seqn) ^ wh[:18]
    return np.concatenate([access_code_bits(lap), np.repeat(header, 3), np.array(pay, np.uint8) ^ wh[18:]])


def make_hopping_capture(sample_rate, center_freq, n_slots, lap, uap, clk0, seed=1, snr_db=24.0, occupancy=0.9,
                         cfo_hz=5e3, start_symbol=40, dh1_fraction=1.0, aliased=False):
    """One master hopping over all 79 channels by the real selection kernel, a DH1 packet in each of
    its transmit slots (even CLK1) with probability `occupancy` (a POLL packet instead with
    probability 1 - dh1_fraction); only the channels inside the capture's band are rendered -- or, with
    `aliased`, all of them at their true offsets, which a 25 Msps capture folds into its band exactly as an
    aliasing receiver does.  Slot k of the capture carries master clock clk0 + k.
    Returns (iq, truth) with truth = [(slot, clock, channel)] of the rendered packets."""
    rng = np.random.default_rng(seed)
    sps = int(round(sample_rate / SYMBOL_RATE))
    slot = 625 * sps
    n = n_slots * slot
    sigma2 = (sample_rate / 1e6) / (10.0 ** (snr_db / 10.0))
    s = math.sqrt(sigma2 / 2)
    iq = (rng.standard_normal(n) * s + 1j * rng.standard_normal(n) * s).astype(np.complex64)
    lo, hi = visible_channels(sample_rate, center_freq)
    address = ((uap & 0xFF) << 24) | (lap & 0xFFFFFF)
    truth = []
    for k in range(n_slots - 4):
        clk = (clk0 + k) & 0x7FFFFFF
        # master transmits in even slots, the addressed slave answers in the following odd slot
        if rng.random() >= (occupancy if not clk & 1 else occupancy * 0.5):
            continue
        ch = hop_channel(address, clk)
        body = bytes(rng.integers(0, 256, int(rng.integers(4, 27)), dtype=np.uint8))
        poll = rng.random() >= dh1_fraction
        lt, flow, arqn, seqn = int(rng.integers(1, 8)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        if aliased or lo <= ch <= hi:
            if poll:
                bits = classic_poll_bits(lap, uap, clk, lt_addr=lt, flow=flow, arqn=arqn, seqn=seqn, ptype=int(rng.integers(0, 2)))
            else:
                bits = classic_dh1_bits(lap, uap, clk, body, lt_addr=lt, flow=flow, arqn=arqn, seqn=seqn)
            add_burst(iq, bits, k * slot + start_symbol * sps, sample_rate, center_freq, ch, rng, cfo_hz=cfo_hz)
            truth.append((k, clk, ch))
    return iq, truth


def add_burst(iq, bits, start, sample_rate, center_freq, channel, rng, cfo_hz=10e3, amplitude=1.0):
    sps = int(round(sample_rate / SYMBOL_RATE))
    bb = gfsk_baseband(bits, sps) * amplitude
    f = (BASE_FREQUENCY + channel * 1e6 - center_freq) + float(rng.uniform(-cfo_hz, cfo_hz))
    m = np.arange(len(bb))
    bb = bb * np.exp(1j * (2 * np.pi * f / sample_rate * m + rng.uniform(0, 2 * np.pi)))
    end = min(start + len(bb), len(iq))
    if end > start:
        iq[start:end] += bb[:end - start].astype(np.complex64)


def make_capture(sample_rate, center_freq, n_slots, laps=(0x24D952,), seed=1, snr_db=25.0,
                 occupancy=0.3, cfo_hz=10e3, max_payload_bits=240, extra_slots=0.0,
                 channels=None, noise=True, amplitude=1.0, cfo_offset_hz=0.0):
    """Return (iq complex64 [n_slots*slot], truth list of dicts).

    snr_db is signal power over the noise power in 1 MHz.  Each piconet (LAP)
    transmits in a slot with probability `occupancy` on a pseudo-random visible
    channel; bursts start 5-15 us into the slot."""
    rng = np.random.default_rng(seed)
    sps = int(round(sample_rate / SYMBOL_RATE))
    assert abs(sps * SYMBOL_RATE - sample_rate) < 1e-6, "integer samples per symbol only"
    slot = 625 * sps
    n = int(n_slots * slot + extra_slots * slot)
    lo, hi = visible_channels(sample_rate, center_freq)
    chans = list(range(lo, hi + 1)) if channels is None else list(channels)
    if noise:
        sigma2 = (sample_rate / 1e6) / (10.0 ** (snr_db / 10.0)) * amplitude ** 2
        s = math.sqrt(sigma2 / 2)
        iq = (rng.standard_normal(n) * s + 1j * rng.standard_normal(n) * s).astype(np.complex64)
    else:
        iq = np.zeros(n, np.complex64)
    truth = []
    for k in range(n_slots):
        used = set()
        for lap in laps:
            if rng.random() >= occupancy:
                continue
            ch = int(rng.choice(chans))
            if ch in used:
                continue
            used.add(ch)
            nb = int(rng.integers(0, max_payload_bits + 1))
            bits = packet_bits(lap, rng, nb)
            start = k * slot + int(rng.integers(5 * sps, 15 * sps))
            bb = gfsk_baseband(bits, sps) * amplitude
            # carrier offset of the burst: cfo_offset_hz (the same for every burst: CFO sweeps) + uniform +-cfo_hz
            f = (BASE_FREQUENCY + ch * 1e6 - center_freq) + float(rng.uniform(-cfo_hz, cfo_hz)) + float(cfo_offset_hz)
            ph0 = rng.uniform(0, 2 * np.pi)
            m = np.arange(len(bb))
            bb = bb * np.exp(1j * (2 * np.pi * f / sample_rate * m + ph0))
            end = min(start + len(bb), n)
            if end > start:
                iq[start:end] += bb[:end - start].astype(np.complex64)
            truth.append(dict(slot=k, channel=ch, lap=lap, start=start, nbits=len(bits)))
    return iq, truth


def burst_schedule(sample_rate, center_freq, slot_begin, slot_end, laps, seed, occupancy, cfo_hz,
                   max_payload_bits):
    """Position-deterministic burst schedule: slot k's bursts depend only on (seed, k), so any
    rank can regenerate any part of one long stream (time-partitioned multi-GPU runs)."""
    sps = int(round(sample_rate / SYMBOL_RATE))
    slot = 625 * sps
    lo, hi = visible_channels(sample_rate, center_freq)
    chans = list(range(lo, hi + 1))
    truth, sched = [], []
    for k in range(max(slot_begin, 0), slot_end):
        rng = np.random.default_rng([seed, k])
        used = set()
        for lap in laps:
            if rng.random() >= occupancy:
                continue
            ch = int(rng.choice(chans))
            if ch in used:
                continue
            used.add(ch)
            nb = int(rng.integers(0, max_payload_bits + 1))
            bits = packet_bits(lap, rng, nb)
            start = k * slot + int(rng.integers(5 * sps, 15 * sps))
            f = (BASE_FREQUENCY + ch * 1e6 - center_freq) + float(rng.uniform(-cfo_hz, cfo_hz))
            ph0 = float(rng.uniform(0, 2 * np.pi))
            truth.append(dict(slot=k, channel=ch, lap=lap, start=start, nbits=len(bits)))
            sched.append((bits, start, f, ph0))
    return truth, sched


def make_segment_torch(sample_rate, center_freq, slot_begin, slot_end, device, laps=(0x24D952,),
                       seed=1, snr_db=25.0, occupancy=0.3, cfo_hz=10e3, max_payload_bits=240,
                       burst_batch=128, left_pad=0):
    """Samples [slot_begin*slot - left_pad, slot_end*slot) of the synthetic stream `seed`, built
    with torch on `device`.  Samples at negative absolute index are zero (GNU Radio history
    pre-fill).  Noise of slot k is seeded by (seed, k); bursts by burst_schedule().

    Returns (iq float32 tensor [n, 2] interleaved I/Q on device, truth list)."""
    import torch

    sps = int(round(sample_rate / SYMBOL_RATE))
    slot = 625 * sps
    pad_slots = (left_pad + slot - 1) // slot
    k0 = slot_begin - pad_slots
    a0 = slot_begin * slot - left_pad                 # absolute index of iq[0]
    n = slot_end * slot - a0
    iq = torch.zeros((n, 2), device=device, dtype=torch.float32)
    sigma2 = (sample_rate / 1e6) / (10.0 ** (snr_db / 10.0))
    gen = torch.Generator(device=device)
    for k in range(max(k0, 0), slot_end):
        gen.manual_seed((seed * 1000003 + k) & 0x7FFFFFFFFFFFFFFF)
        blk = torch.randn((slot, 2), generator=gen, device=device, dtype=torch.float32)
        blk.mul_(math.sqrt(sigma2 / 2))
        lo_a, hi_a = max(k * slot, a0), (k + 1) * slot
        if hi_a > lo_a:
            iq[lo_a - a0:hi_a - a0] = blk[lo_a - k * slot:]
    # bursts that can overlap the segment: those starting up to one max-burst earlier
    max_bits = 72 + 54 + max_payload_bits
    L = max_bits * sps
    back = (L + slot - 1) // slot
    truth, sched = burst_schedule(sample_rate, center_freq, k0 - back, slot_end, laps, seed, occupancy,
                                  cfo_hz, max_payload_bits)
    g = torch.tensor(gaussian_pulse(sps), device=device, dtype=torch.float64)
    pad = (len(g) - 1) // 2
    flat = iq.view(-1)
    for b0 in range(0, len(sched), burst_batch):
        batch = sched[b0:b0 + burst_batch]
        B = len(batch)
        nrz = torch.zeros((B, max_bits), dtype=torch.float64)
        lens = []
        for i, (bits, _, _, _) in enumerate(batch):
            nrz[i, :len(bits)] = torch.from_numpy(2.0 * bits.astype(np.float64) - 1.0)
            lens.append(len(bits) * sps)
        nrz = nrz.to(device).repeat_interleave(sps, dim=1)                       # [B, L]
        f = torch.nn.functional.conv1d(nrz[:, None, :], g[None, None, :], padding=pad)[:, 0, :]
        phase = math.pi * 0.32 * torch.cumsum(f, dim=1) / sps
        m = torch.arange(L, device=device, dtype=torch.float64)
        fo = torch.tensor([b[2] for b in batch], device=device, dtype=torch.float64)
        p0 = torch.tensor([b[3] for b in batch], device=device, dtype=torch.float64)
        phase = phase + (2 * math.pi / sample_rate) * fo[:, None] * m[None, :] + p0[:, None]
        re = torch.cos(phase).to(torch.float32)
        im = torch.sin(phase).to(torch.float32)
        for i, (_, start, _, _) in enumerate(batch):
            lo_a = max(start, a0)
            hi_a = min(start + lens[i], a0 + n)
            if hi_a <= lo_a:
                continue
            seg = flat[2 * (lo_a - a0): 2 * (hi_a - a0)].view(hi_a - lo_a, 2)
            seg[:, 0] += re[i, lo_a - start:hi_a - start]
            seg[:, 1] += im[i, lo_a - start:hi_a - start]
    truth = [t for t in truth if t["slot"] >= 0]
    return iq, truth
