// classic.cc -- see classic.h.  Reference behaviour is kept down to its quirks:
//  * unfec23 never corrects a data bit: the byte that collects the five syndrome bits still holds the
//    mismatch count and is shifted once too often, so its switch cannot match (lib/packet_impl.cc:433-464);
//  * EV3/EV5 unwhiten the first payload byte again and again (:905, :990) and their positive CRC
//    results are discarded by crc_check (:665-667);
//  * decode_payload's EV5 case falls through into the DM5 case (:1147-1153);
//  * nap_from_fhs reads 16 bits through an 8-bit accumulator (:1259-1263).
#include "classic.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include <fcntl.h>
#include <linux/if.h>
#include <linux/if_tun.h>
#include <sys/ioctl.h>
#include <unistd.h>

namespace gr {
namespace bluetooth {
namespace host {

namespace {

void appendf(std::string &out, const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    out += buf;
}

// whitening sequence of x^7 + x^4 + 1 and, per CLK1-6 value, where in it the register that starts
// with position 6 = 1, positions 0..5 = CLK1..CLK6 begins (lib/packet_impl.cc:84-90, :182-186)
struct whitening_tables {
    uint8_t seq[127];
    uint8_t start[64];
    whitening_tables()
    {
        const uint8_t seed[7] = {1, 1, 1, 0, 0, 0, 1};
        for (int i = 0; i < 7; i++) seq[i] = seed[i];
        for (int i = 7; i < 127; i++) seq[i] = seq[i - 7] ^ seq[i - 3];
        for (int clk = 0; clk < 64; clk++) {
            uint8_t reg[7], first[7];
            for (int i = 0; i < 6; i++) reg[i] = (clk >> i) & 1;
            reg[6] = 1;
            for (int k = 0; k < 7; k++) {
                const uint8_t o = reg[6];
                first[k] = o;
                const uint8_t nxt[7] = {o, reg[0], reg[1], reg[2], (uint8_t)(reg[3] ^ o), reg[4], reg[5]};
                std::memcpy(reg, nxt, 7);
            }
            start[clk] = 0;
            for (int i = 0; i < 127; i++) {
                bool same = true;
                for (int k = 0; k < 7 && same; k++) same = seq[(i + k) % 127] == first[k];
                if (same) { start[clk] = (uint8_t)i; break; }
            }
        }
    }
};
const whitening_tables &wt()
{
    static const whitening_tables t;
    return t;
}

uint8_t reverse8(uint8_t b)
{
    uint8_t r = 0;
    for (int i = 0; i < 8; i++) r |= (uint8_t)(((b >> i) & 1) << (7 - i));
    return r;
}

const char *const TYPE_NAMES[16] = {"NULL", "POLL", "FHS", "DM1", "DH1/2-DH1", "HV1", "HV2/2-EV3", "HV3/EV3/3-EV3",
                                    "DV/3-DH1", "AUX1", "DM3/2-DH3", "DH3/3-DH3", "EV4/2-EV5", "EV5/3-EV5",
                                    "DM5/2-DH5", "DH5/3-DH5"};
constexpr uint32_t GIAC = 0x9e8b33, LIAC = 0x9e8b00;

}  // namespace

// ------------------------------------------------------------------------ tap_sink
tap_sink::~tap_sink()
{
    if (d_fd >= 0) ::close(d_fd);
    if (d_file) fclose(d_file);
}

bool tap_sink::open(const char *name)
{
    int fd = ::open("/dev/net/tun", O_RDWR);
    if (fd >= 0) {
        struct ifreq ifr;
        std::memset(&ifr, 0, sizeof ifr);
        ifr.ifr_flags = IFF_TAP | IFF_NO_PI;
        snprintf(ifr.ifr_name, IFNAMSIZ, "%s", name);
        int one = 1;
        if (ioctl(fd, TUNSETIFF, (void *)&ifr) == -1 || ioctl(fd, TUNSETPERSIST, (void *)&one) == -1) { ::close(fd); fd = -1; }
    }
    d_fd = fd;
    if (d_fd < 0) {
        const char *path = getenv("BTGPU_TAP_FILE");
        if (path && *path) d_file = fopen(path, "wb");
    }
    return is_open();
}

std::vector<uint8_t> tap_sink::frame(const uint8_t *data, unsigned len, uint64_t src_addr, uint64_t dst_addr, uint16_t ether_type)
{
    std::vector<uint8_t> f(14 + len, 0);
    for (int i = 0; i < 6; i++) {
        f[i] = (uint8_t)(dst_addr >> (8 * (5 - i)));
        f[6 + i] = (uint8_t)(src_addr >> (8 * (5 - i)));
    }
    f[12] = (uint8_t)(ether_type >> 8);
    f[13] = (uint8_t)ether_type;
    if (data && len) std::memcpy(&f[14], data, len);
    return f;
}

void tap_sink::write(const uint8_t *data, unsigned len, uint64_t src_addr, uint64_t dst_addr, uint16_t ether_type)
{
    if (!is_open()) return;
    const std::vector<uint8_t> f = frame(data, len, src_addr, dst_addr, ether_type);
    if (d_fd >= 0) { if (::write(d_fd, f.data(), f.size()) == -1) perror("write"); return; }
    const uint32_t n = (uint32_t)f.size();
    const uint8_t hdr[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    fwrite(hdr, 1, 4, d_file);
    fwrite(f.data(), 1, f.size(), d_file);
    fflush(d_file);
}

// ------------------------------------------------------------------ classic_packet
uint32_t classic_packet::bits(const uint8_t *air, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint32_t)(air[i] & 1) << i;
    return v;
}

classic_packet::classic_packet(const uint8_t *symbols, int length, uint32_t clkn, int channel, const btgpu_header &sweep)
    : d_symbols(MAX_SYMBOLS + 64, 0), d_clkn(clkn), d_channel(channel), d_sweep(sweep), d_payload(3000, 0)
{
    if (length > MAX_SYMBOLS) length = MAX_SYMBOLS;
    if (length < 0) length = 0;
    if (length > 0) std::memcpy(d_symbols.data(), symbols, (size_t)length);   // (a hit may hand over no symbols at all)
    d_length = length;
    d_lap = bits(&d_symbols[38], 24);
}

bool classic_packet::header_present() const
{
    // trailer (4 bits alternating out of the sync word's last bit) + the 54 header symbols must
    // look like a 1/3-rate code: fewer than ID_THRESHOLD (5) disagreements
    if (d_length < 126) return false;
    const uint8_t *st = &d_symbols[67];
    int be = 0;
    const uint8_t msb = st[0];
    be += st[1] ^ !msb; be += st[2] ^ msb; be += st[3] ^ !msb; be += st[4] ^ msb;
    st += 5;
    for (int a = 0; a < 54; a += 3) be += (st[a] ^ st[a + 1]) | (st[a + 1] ^ st[a + 2]) | (st[a + 2] ^ st[a]);
    return be < 5;
}

bool classic_packet::unfec13(const uint8_t *in, uint8_t *out, int length)
{
    int be = 0;
    for (int i = 0; i < length; i++) {
        const uint8_t a = in[3 * i] & 1, b = in[3 * i + 1] & 1, c = in[3 * i + 2] & 1;
        out[i] = (uint8_t)((a & b) | (b & c) | (c & a));
        be += (a ^ b) | (b ^ c) | (c ^ a);
    }
    return be < length / 4;
}

bool classic_packet::unfec23(const uint8_t *in, int length, std::vector<uint8_t> &out)
{
    if (length % 10) length += 10 - length % 10;
    out.assign((size_t)length, 0);
    for (int blk = 0; blk < length / 10; blk++) {
        const uint8_t *cw = in + 15 * blk;
        for (int k = 0; k < 10; k++) out[(size_t)10 * blk + k] = cw[k];
        // parity of the ten data bits: the register of lfsr(data, 15, 10, {1,1,0,1,0,1})
        // (lib/packet_impl.cc:278-307), feedback taps at positions 0, 1 and 3
        uint8_t par[5] = {0, 0, 0, 0, 0};
        for (int i = 9; i >= 0; i--) {
            const uint8_t fb = (uint8_t)((cw[i] & 1) ^ par[4]);
            par[4] = par[3];
            par[3] = (uint8_t)(par[2] ^ fb);
            par[2] = par[1];
            par[1] = (uint8_t)(par[0] ^ fb);
            par[0] = fb;
        }
        int mismatches = 0;
        for (int k = 0; k < 5; k++) mismatches += par[k] != (cw[10 + k] & 1);
        if (mismatches >= 2) return false;        // the reference's correction table is unreachable (see top)
    }
    return true;
}

void classic_packet::unwhiten(const uint8_t *in, uint8_t *out, int clock, int length, int skip)
{
    const whitening_tables &t = wt();
    int idx = (t.start[clock & 0x3f] + skip) % 127;
    for (int i = 0; i < length; i++) {
        out[i] = (uint8_t)((in[i] & 1) ^ t.seq[idx]);
        idx = idx + 1 == 127 ? 0 : idx + 1;
    }
}

uint16_t classic_packet::crcgen(const uint8_t *payload, int length, int uap)
{
    uint16_t reg = (uint16_t)(reverse8((uint8_t)uap) << 8);
    for (int i = 0; i < length; i++) {
        reg = (uint16_t)((reg >> 1) | (((reg & 1) ^ (payload[i] & 1)) << 15));
        reg ^= (uint16_t)((reg & 0x8000) >> 5);
        reg ^= (uint16_t)((reg & 0x8000) >> 12);
    }
    return reg;
}

int classic_packet::uap_from_hec(uint16_t data, uint8_t hec)
{
    for (int i = 9; i >= 0; i--) {
        if (hec & 0x80) hec ^= 0x65;
        hec = (uint8_t)((hec << 1) | (((hec >> 7) ^ (data >> i)) & 1));
    }
    return reverse8(hec);
}

uint8_t classic_packet::try_clock(int clock)
{
    if (!d_sweep.fec13_ok) return 0;               // type and UAP stay as they were
    d_uap = d_sweep.uap[clock & 63];
    d_type = d_sweep.type[clock & 63];
    return d_uap;
}

void classic_packet::set_clock(uint32_t clock, bool have27)
{
    d_clock = have27 ? (clock & 0x7ffffff) : (clock & 0x3f);
    d_have_clk6 = true;
    d_have_clk27 = have27;
}

bool classic_packet::payload_crc() const
{
    // EV4 starts with a payload length of 1: the reference then takes the 16 check bits from eight bytes in
    // front of d_payload (lib/packet_impl.cc:675-686 called from :946-1001; undefined behaviour).  Policy, as
    // in the oracle: bits in front of the payload read as 0, a negative length is an empty CRC input.
    const int start = (d_payload_length - 2) * 8;
    const uint16_t crc = crcgen(d_payload.data(), start, d_uap);
    uint16_t chk = 0;
    for (int i = 0; i < 16; i++)
        if (start + i >= 0) chk |= (uint16_t)((d_payload[(size_t)(start + i)] & 1) << i);
    return crc == chk;
}

int classic_packet::fhs(int clock)
{
    const uint8_t *stream = &d_symbols[126];
    const int size = d_length - 126;
    d_payload_length = 20;
    if (size < d_payload_length * 12) return 1;
    std::vector<uint8_t> corrected;
    if (!unfec23(stream, d_payload_length * 8, corrected)) return 0;
    unwhiten(corrected.data(), d_payload.data(), clock, d_payload_length * 8, 18);
    if (payload_crc()) return 1000;
    for (int c = 32; c < 64; c++) {                // every X-input value
        unwhiten(corrected.data(), d_payload.data(), c, d_payload_length * 8, 18);
        if (payload_crc()) return 1000;
    }
    return 0;
}

bool classic_packet::decode_payload_header(const uint8_t *stream, int clock, int header_bytes, int size, bool fec)
{
    uint8_t ph[16];
    const int nbits = header_bytes * 8;
    if (size < nbits) return false;
    if (fec) {
        if (size < nbits * 15 / 8) return false;   // 30 resp. 15 symbols
        std::vector<uint8_t> corrected;
        if (!unfec23(stream, nbits, corrected)) return false;
        unwhiten(corrected.data(), ph, clock, nbits, 18);
    } else {
        unwhiten(stream, ph, clock, nbits, 18);
    }
    // body length + payload header + 2 bytes CRC
    d_payload_length = header_bytes == 2 ? (int)bits(&ph[3], 10) + 4 : (int)bits(&ph[3], 5) + 3;
    d_llid = (int)bits(&ph[0], 2);
    d_flow = (int)bits(&ph[2], 1);
    d_payload_header_length = header_bytes;
    return true;
}

int classic_packet::DM(int clock)
{
    const uint8_t *stream = &d_symbols[126];
    int size = d_length - 126, header_bytes = 2, max_length;
    switch (d_type) {
        case 8: stream += 80; size -= 80; header_bytes = 1; max_length = 12; break;     // DV: skip the voice field
        case 3: header_bytes = 1; max_length = 20; break;
        case 10: max_length = 125; break;
        case 14: max_length = 228; break;
        default: return 0;
    }
    if (!decode_payload_header(stream, clock, header_bytes, size, true)) return 0;
    if (d_payload_length > max_length) return 1;   // could be encrypted
    const int bitlength = d_payload_length * 8;
    if (bitlength > size) return 1;
    std::vector<uint8_t> corrected;
    if (!unfec23(stream, bitlength, corrected)) return 0;
    unwhiten(corrected.data(), d_payload.data(), clock, bitlength, 18);
    return payload_crc() ? 10 : 1;
}

int classic_packet::DH(int clock)
{
    const uint8_t *stream = &d_symbols[126];
    const int size = d_length - 126;
    int header_bytes = 2, max_length;
    switch (d_type) {
        case 9: case 4: header_bytes = 1; max_length = 30; break;
        case 11: max_length = 187; break;
        case 15: max_length = 343; break;
        default: return 0;
    }
    if (!decode_payload_header(stream, clock, header_bytes, size, false)) return 0;
    if (d_payload_length > max_length) return 1;
    const int bitlength = d_payload_length * 8;
    if (bitlength > size) return 1;
    unwhiten(stream, d_payload.data(), clock, bitlength, 18);
    if (d_type == 9) return 1;                     // AUX1 has no CRC
    return payload_crc() ? 10 : 1;
}

int classic_packet::EV(int clock, int maxlength)
{
    const uint8_t *stream = &d_symbols[126];
    const int size = d_length - 126;
    for (d_payload_length = 0; d_payload_length < maxlength; d_payload_length++) {
        const int nb = d_payload_length * 8;
        if (nb + 8 > size) return 1;
        unwhiten(stream, &d_payload[(size_t)nb], clock, 8, 18 + nb);     // always the first byte of the stream
        if (d_payload_length > 2 && payload_crc()) return 10;
    }
    return 1;
}

int classic_packet::EV4(int clock)
{
    const uint8_t *stream = &d_symbols[126];
    const int size = d_length - 126;
    int syms = 0, nb = 0;
    d_payload_length = 1;
    while (syms < 1470) {
        if (syms + 15 > size) return 1;
        std::vector<uint8_t> corrected;
        if (!unfec23(stream + syms, 10, corrected)) return syms < 45 ? 0 : 1;
        unwhiten(corrected.data(), &d_payload[(size_t)nb], clock, 10, 18 + nb);
        while (d_payload_length * 8 <= nb) {
            if (payload_crc()) return 10;
            d_payload_length++;
        }
        syms += 15;
        nb += 10;
    }
    return 1;
}

int classic_packet::HV(int clock)
{
    const uint8_t *stream = &d_symbols[126];
    const int size = d_length - 126;
    if (size < 240) { d_payload_length = 0; return 1; }
    if (d_type == 5) {
        uint8_t corrected[80];
        if (!unfec13(stream, corrected, 80)) return 0;
        d_payload_length = 10;
        unwhiten(corrected, d_payload.data(), clock, 80, 18);
    } else if (d_type == 6) {
        std::vector<uint8_t> corrected;
        if (!unfec23(stream, 160, corrected)) return 0;
        d_payload_length = 20;
        unwhiten(corrected.data(), d_payload.data(), clock, 160, 18);
    } else if (d_type == 7) {
        d_payload_length = 30;
        unwhiten(stream, d_payload.data(), clock, 240, 18);
    }
    return 1;
}

int classic_packet::crc_check(int clock)
{
    int r = 1;                                     // 1 inconclusive, > 1 positive, 0 negative
    switch (d_type) {
        case 2: r = fhs(clock); break;
        case 8: case 3: case 10: case 14: r = DM(clock); break;
        case 4: case 11: case 15: r = DH(clock); break;
        case 7: r = EV(clock, 32); break;
        case 12: r = EV4(clock); break;
        case 13: r = EV(clock, 182); break;
        case 5: r = HV(clock); break;
        default: break;
    }
    if (r == 0 && d_type != 2 && d_type != 3 && d_type != 5) return 1;      // may be another logical transport
    if (r > 1 && (d_type == 7 || d_type == 13)) return 1;                   // EV3 / EV5: too many false positives
    return r;
}

bool classic_packet::decode_header(std::string &out)
{
    uint8_t header[18];
    if (d_have_clk6 && unfec13(&d_symbols[72], header, 18)) {
        unwhiten(header, d_header, (int)d_clock, 18, 0);
        const int uap = uap_from_hec((uint16_t)bits(d_header, 10), (uint8_t)bits(d_header + 10, 8));
        if (uap == d_uap) { d_type = (int)bits(&d_header[3], 4); return true; }
        appendf(out, "bad HEC! %02x %02x %i ", uap, d_uap, (int)bits(&d_header[3], 4));
    }
    out += "failed to decode header\n";
    return false;
}

void classic_packet::decode_payload()
{
    const int clk = (int)d_clock;
    d_payload_header_length = 0;
    switch (d_type) {
        case 0: case 1: d_payload_length = 0; break;
        case 2: fhs(clk); break;
        case 3: case 8: case 10: case 14: DM(clk); break;
        case 4: case 9: case 11: case 15: DH(clk); break;
        case 5: case 6: HV(clk); break;
        case 7: if (EV(clk, 32) <= 1) HV(clk); break;
        case 12: EV4(clk); break;
        case 13: EV(clk, 182); DM(clk); break;     // falls through into the DM5 case
    }
    d_have_payload = true;
}

void classic_packet::decode(std::string &out)
{
    d_have_payload = false;
    if (decode_header(out)) decode_payload();
}

void classic_packet::print(std::string &out) const
{
    if (!d_have_payload) return;
    out += TYPE_NAMES[d_type & 15];
    out += "\n";
    if (d_payload_header_length > 0)
        appendf(out, "  LLID: %d\n  flow: %d\n  payload length: %d\n", d_llid, d_flow, d_payload_length);
}

// ------------------------------------------------------------------------ le_packet
std::string le_packet_text(const uint8_t *symbols, int avail, int channel)
{
    std::string out;
    // le_packet::freq2index (lib/packet_impl.cc:1285-1314): LE channels sit on the even classic channels
    if (channel < 0 || channel > 78 || (channel & 1)) return out;
    const int chan = channel / 2;
    const int index = chan == 0 ? 37 : chan == 12 ? 38 : chan == 39 ? 39 : (chan < 12 ? chan - 1 : chan - 2);
    // whitening: the x^7 + x^4 + 1 register started with position 0 = 1, positions 1..6 = the channel
    // index MSB first (:1446-1450), from link symbol 40 (the PDU header) on
    uint8_t reg[7];
    reg[0] = 1;
    for (int i = 0; i < 6; i++) reg[1 + i] = (uint8_t)((index >> (5 - i)) & 1);
    constexpr int N = 376;                         // LE_MAX_SYMBOLS; symbols past `avail` count as 0
    uint8_t link[N];
    for (int i = 0; i < N; i++) link[i] = (uint8_t)(i < avail ? (symbols[i] & 1) : 0);
    for (int i = 40; i < N; i++) {
        const uint8_t o = reg[6];
        link[i] ^= o;
        const uint8_t nxt[7] = {o, reg[0], reg[1], reg[2], (uint8_t)(reg[3] ^ o), reg[4], reg[5]};
        std::memcpy(reg, nxt, 7);
    }
    const uint32_t aa = classic_packet::bits(&link[8], 32);
    const uint32_t header = classic_packet::bits(&link[40], 16);
    uint8_t pdu[64] = {0};                         // d_pdu[39]; bytes beyond it (read out of bounds by the reference) = 0
    for (int pi = 0, i = 56; i + 8 < N; pi++, i += 8) pdu[pi] = (uint8_t)classic_packet::bits(&link[i], 8);
    auto six = [&](const char *name, int at) {
        appendf(out, "  %s=%02x%02x%02x%02x%02x%02x\n", name, pdu[at], pdu[at + 1], pdu[at + 2], pdu[at + 3], pdu[at + 4], pdu[at + 5]);
    };
    if (index >= 37) {
        const unsigned type = header & 0xf, length = (header >> 8) & 0x3f;
        appendf(out, "BTLE index=%02d, AA=%08x, PDUType=%d, TxAdd=%d, RxAdd=%d, Length=%d\n", index, aa, type, (header >> 6) & 1,
                (header >> 7) & 1, length);
        switch (type) {
            case 0: case 2: case 4: case 6: {
                const char *what = type == 4 ? "ScanRspData" : "AdvData";
                six("AdvA", 0);
                appendf(out, "\n  (char) %s=", what);
                for (unsigned i = 6; i < length; i++) {
                    char c = (char)pdu[i];
                    if (c < ' ' || c > '~') c = '.';
                    appendf(out, " %c", c);
                }
                appendf(out, "\n  (byte) %s=", what);
                for (unsigned i = 6; i < length; i++) appendf(out, "%02x", pdu[i]);
                out += "\n";
                break;
            }
            case 1: six("AdvA", 0); six("InitA", 6); break;
            case 3: six("ScanA", 0); six("AdvA", 6); break;
            case 5: {
                six("InitA", 0); six("AdvA", 6);
                auto le = [&](int at, int nbytes) { uint64_t v = 0; for (int k = 0; k < nbytes; k++) v |= (uint64_t)pdu[at + k] << (8 * k); return v; };
                appendf(out, "  AA=%08x, CRCInit=%06x, WinSize=%d, WinOffset=%d\n", (unsigned)le(12, 4), (unsigned)le(16, 3), pdu[19],
                        (int)le(20, 2));
                appendf(out, "  Interval=%d, Latency=%d, Timeout=%d, ChM=%010lx, Hop=%d, SCA=%d\n", (int)le(22, 2), (int)le(24, 2),
                        (int)le(26, 2), (unsigned long)le(28, 5), pdu[33] & 0x1f, (pdu[33] >> 5) & 7);
                break;
            }
            default: break;
        }
    } else {
        appendf(out, "BTLE index=%02d, AA=%08x, LLID=%d, NESN=%d, SN=%d, MD=%d, Length=%d\n", index, aa, header & 3, (header >> 2) & 1,
                (header >> 3) & 1, (header >> 4) & 1, (header >> 8) & 0x1f);
    }
    return out;
}

std::vector<uint8_t> classic_packet::tun_format() const
{
    std::vector<uint8_t> t((size_t)9 + (size_t)d_payload_length, 0);
    t[0] = (uint8_t)d_clock; t[1] = (uint8_t)(d_clock >> 8); t[2] = (uint8_t)(d_clock >> 16); t[3] = (uint8_t)(d_clock >> 24);
    t[4] = (uint8_t)d_channel;
    t[5] = (uint8_t)((d_have_clk27 ? 1 : 0) | ((d_have_nap ? 1 : 0) << 1));
    t[6] = (uint8_t)bits(&d_header[0], 7);         // LT_ADDR and type
    t[7] = (uint8_t)bits(&d_header[7], 3);         // flags
    t[8] = (uint8_t)bits(&d_header[10], 8);        // HEC
    for (int i = 0; i < d_payload_length; i++) t[(size_t)9 + i] = (uint8_t)bits(&d_payload[(size_t)i * 8], 8);
    return t;
}

// --------------------------------------------------------------- basic_rate_piconet
basic_rate_piconet::~basic_rate_piconet()
{
    if (d_hops) btgpu_hopseq_destroy(d_hops);
}

void basic_rate_piconet::reset(std::string &out)
{
    out += "no candidates remaining! starting over . . .\n";
    if (d_hop_reversal_inited && d_hops) { btgpu_hopseq_destroy(d_hops); d_hops = nullptr; }
    d_got_first_packet = false;
    d_packets_observed = 0;
    d_hop_reversal_inited = false;
    d_have_uap = false;
    d_have_clk6 = false;
    d_have_clk27 = false;
    // two packets in a row on one channel were seen: assume AFH next time
    d_afh = d_looks_like_afh;
    d_looks_like_afh = false;
}

int basic_rate_piconet::init_hop_reversal(bool aliased, std::string &out)
{
    out += "\nCalculating complete hopping sequence.\n";
    if (d_hops) { btgpu_hopseq_destroy(d_hops); d_hops = nullptr; }
    const uint32_t address = (((uint32_t)d_uap << 24) | d_lap) & 0xfffffff;
    if (btgpu_hopseq_create(address, d_afh ? 1 : 0, -1, &d_hops) != BTGPU_OK) {
        fprintf(stderr, "Error: btgpu_hopseq_create failed\n");      // no CPU fallback
        abort();
    }
    const int clock = (int)((d_clk_offset + d_first_pkt_time) & 0x3f);
    d_num_candidates = btgpu_hopseq_init_candidates(d_hops, d_pattern_channels[0], clock, aliased ? 1 : 0);
    d_winnowed = 0;
    d_hop_reversal_inited = true;
    d_have_clk27 = false;
    d_aliased = aliased;
    appendf(out, "%d initial CLK1-27 candidates\n", d_num_candidates);
    return d_num_candidates;
}

int basic_rate_piconet::winnow(int offset, int channel, std::string &out)
{
    const int n = btgpu_hopseq_winnow(d_hops, offset, channel, d_aliased ? 1 : 0);
    d_num_candidates = n;
    if (n == 1) {
        uint32_t survivor = 0;
        btgpu_hopseq_candidates(d_hops, &survivor, 1);
        d_clk_offset = (survivor - d_first_pkt_time) & 0x7ffffff;
        d_have_clk27 = true;
        appendf(out, "\nAcquired CLK1-27 offset = 0x%07x\n", d_clk_offset);
    } else if (n == 0) {
        reset(out);
    } else {
        appendf(out, "%d CLK1-27 candidates remaining\n", n);
    }
    return n;
}

int basic_rate_piconet::winnow(std::string &out)
{
    int n = d_num_candidates;
    for (; d_winnowed < d_packets_observed; d_winnowed++) {
        const int index = d_pattern_indices[d_winnowed], channel = d_pattern_channels[d_winnowed];
        n = winnow(index, channel, out);
        if (!d_hop_reversal_inited) break;         // reset() inside: the observed pattern is gone
        if (d_winnowed > 0) {                      // (the reference also looks at entry -1 when d_winnowed == 0)
            const int last_index = d_pattern_indices[d_winnowed - 1], last_channel = d_pattern_channels[d_winnowed - 1];
            if (!d_looks_like_afh && index == last_index + 1 && channel == last_channel) d_looks_like_afh = true;
        }
    }
    return n;
}

int basic_rate_piconet::hop(uint32_t clock)
{
    uint32_t idx = clock;
    uint8_t ch = 0;
    if (!d_hops || btgpu_hopseq_lookup(d_hops, &idx, 1, &ch) != BTGPU_OK) return -1;
    return ch;
}

// UAP / CLK1-6 discovery (what lib/piconet_impl.cc:433-517 does, on this class's own state).
// State: bit k of d_clk6_alive = "if the FIRST packet was sent at CLK1-6 = k, every header seen so far checks out";
// d_clk6_uap[k] = the UAP that hypothesis implies.  A packet seen `elapsed` slots after the first one is tested
// under hypothesis k with clock (k + elapsed) mod 64: try_clock() gives the UAP that makes the HEC come out, and
// the hypothesis survives if that is the UAP it already implied and crc_check() does not refute it.
bool basic_rate_piconet::remember_hop(uint32_t clkn, int channel, std::string &out)
{
    if (d_packets_observed >= 1000) {              // MAX_PATTERN_LENGTH
        out += "Oops. More hops than we can remember.\n";
        reset(out);
        return false;
    }
    d_pattern_indices[d_packets_observed] = (int)(clkn - d_first_pkt_time);
    d_pattern_channels[d_packets_observed] = (uint8_t)channel;
    d_packets_observed++;
    d_total_packets_observed++;
    return true;
}

void basic_rate_piconet::lock_clk6(int k, uint8_t uap)
{
    d_clk_offset = (uint32_t)((k - (int)(d_first_pkt_time & 0x3f)) & 0x3f);
    d_uap = uap;
    d_have_clk6 = d_have_uap = true;
}

bool basic_rate_piconet::uap_from_header(classic_packet &pkt, std::string &out)
{
    const uint32_t clkn = pkt.clkn();
    const bool first = !d_got_first_packet;
    if (first) d_first_pkt_time = clkn;
    if (!remember_hop(clkn, pkt.channel(), out)) return false;
    const uint32_t elapsed = clkn - d_first_pkt_time;
    const uint64_t tried = first ? ~0ull : d_clk6_alive;      // the first packet opens all 64 hypotheses
    if (first) d_clk6_alive = ~0ull;
    for (uint64_t todo = tried; todo; todo &= todo - 1) {
        const int k = __builtin_ctzll(todo);
        const int clock = (int)(((uint32_t)k + elapsed) & 63u);
        const uint8_t uap = pkt.try_clock(clock);
        // crc_check: 0 refuted, 1 possible, > 1 payload CRC correct (proof); an inconsistent UAP refutes without looking
        const int verdict = (first || uap == d_clk6_uap[k]) ? pkt.crc_check(clock) : 0;
        if (verdict >= 2) {
            // (the reference returns here before it marks the first packet as seen: kept, see
            // tests/test_host_block_gpu.py::test_hopper_block_crc_success_quirk)
            appendf(out, "Correct CRC! UAP = 0x%x found after %d total packets.\n", uap, d_total_packets_observed);
            lock_clk6(k, uap);
            d_total_packets_observed = 0;
            return true;
        }
        if (verdict == 1) d_clk6_uap[k] = uap;
        else d_clk6_alive &= ~(1ull << k);
    }
    d_got_first_packet = true;
    const int before = __builtin_popcountll(tried), left = __builtin_popcountll(d_clk6_alive);
    appendf(out, "reduced from %d to %d CLK1-6 candidates\n", before, left);
    if (left == 1) {
        const int k = __builtin_ctzll(d_clk6_alive);
        lock_clk6(k, d_clk6_uap[k]);
        appendf(out, "We have a winner! UAP = 0x%x found after %d total packets.\n", d_uap, d_total_packets_observed);
        d_total_packets_observed = 0;
        return true;
    }
    if (left == 0) reset(out);
    return false;
}

// ------------------------------------------------------------------ hopper_handlers
std::string hopper_handlers::hit(const btgpu_hit &hit, const btgpu_header &sweep, const uint8_t *symbols, int nsymbols)
{
    std::string out;
    if (hit.slot != d_slot) {                      // a new work() call of the reference
        d_slot = hit.slot;
        d_last_channel = -1;
        d_slot_done = false;
        d_locked = d_piconet.have_clk27();
    }
    if (hit.kind != BTGPU_KIND_AC || hit.channel == d_last_channel) return out;      // sniff_ac: first hit of a channel
    d_last_channel = hit.channel;
    if (d_slot_done) return out;
    const uint32_t clkn = (uint32_t)(hit.slot & 0x7ffffff);
    if (d_locked) {
        // hopalong: only the channel the sequence predicts for this slot
        const uint32_t clock27 = (clkn + d_piconet.offset()) & 0x7ffffff;
        const int hop = d_piconet.hop(clock27);
        const int obs = d_aliased ? basic_rate_piconet::aliased_channel(hop) : hop;
        // the reference tunes to the true hop frequency (lib/multi_hopper_impl.cc:166); in an aliasing capture
        // that offset equals, modulo the sample rate, the observed channel's -- the hit to take is the one on `obs`
        if (obs < d_low || obs > d_high || hit.channel != obs) return out;
        d_slot_done = true;
        classic_packet pkt(symbols, nsymbols, 0, obs, sweep);
        if (pkt.lap() != d_lap) return out;
        appendf(out, "clock 0x%07x, channel %2d: ", clock27, obs);
        if (pkt.header_present()) {
            pkt.set_uap(d_piconet.uap());
            pkt.set_clock(clock27, true);
            pkt.decode(out);
            if (pkt.got_payload()) {
                pkt.print(out);
                if (d_tap) {
                    const std::vector<uint8_t> data = pkt.tun_format();
                    d_tap->write(data.data(), (unsigned)data.size(), 0, ((uint32_t)pkt.uap() << 24) | pkt.lap(), tap_sink::ETHER_TYPE);
                }
            }
        } else {
            out += "ID\n";
            if (d_tap) d_tap->write(nullptr, 0, 0, ((uint32_t)d_piconet.uap() << 24) | pkt.lap(), tap_sink::ETHER_TYPE);
        }
        return out;
    }
    classic_packet pkt(symbols, nsymbols, clkn, hit.channel, sweep);
    if (pkt.lap() != d_lap || !pkt.header_present()) return out;
    d_slot_done = true;                            // the reference breaks out of its channel loop here
    if (!d_piconet.have_clk6()) {
        d_piconet.uap_from_header(pkt, out);       // CLK1-6 / UAP discovery
        if (d_piconet.have_clk6()) {
            d_piconet.init_hop_reversal(d_aliased, out);
            d_piconet.winnow(out);                 // with the packets seen so far
        }
    } else {
        d_piconet.uap_from_header(pkt, out);       // timing of one more packet
        if (d_piconet.have_clk6()) d_piconet.winnow(out);
    }
    return out;
}

// ----------------------------------------------------------------- sniffer_handlers
std::string sniffer_handlers::ac(const btgpu_hit &hit, const btgpu_header &sweep, const uint8_t *symbols, int nsymbols)
{
    std::string out;
    const uint32_t clkn = (uint32_t)(hit.slot & 0x7ffffff);
    auto pkt = std::make_shared<classic_packet>(symbols, nsymbols, clkn, hit.channel, sweep);
    const uint32_t lap = pkt->lap();
    appendf(out, "time %6d, snr=%.1f, channel %2d, LAP %06x ", (int)clkn, hit.snr_db, hit.channel, lap);
    if (!pkt->header_present()) { id(lap, out); return out; }
    auto &slot = d_piconets[lap];
    if (!slot) slot = std::make_shared<basic_rate_piconet>(lap);
    auto pn = slot;
    if (pn->have_clk6() && pn->have_uap()) decode(pkt, pn, true, out);
    else discover(pkt, pn, out);
    if (lap == GIAC || lap == LIAC) d_piconets.erase(lap);      // inquiry responses: keep no state
    return out;
}

void sniffer_handlers::id(uint32_t lap, std::string &out)
{
    out += "ID\n";
    if (d_tap) d_tap->write(nullptr, 0, 0, lap, tap_sink::ETHER_TYPE);
}

void sniffer_handlers::decode(std::shared_ptr<classic_packet> pkt, std::shared_ptr<basic_rate_piconet> pn, bool first_run,
                              std::string &out)
{
    pkt->set_clock(pkt->clkn() + pn->offset(), pn->have_clk27());
    pkt->set_uap(pn->uap());
    pkt->decode(out);
    if (pkt->got_payload()) {
        pkt->print(out);
        if (d_tap) {
            uint64_t addr = ((uint64_t)pkt->uap() << 24) | pkt->lap();
            if (pn->have_nap()) { addr |= (uint64_t)pn->nap() << 32; pkt->set_nap(); }
            const std::vector<uint8_t> data = pkt->tun_format();      // 9 bytes of meta data and header + payload
            d_tap->write(data.data(), (unsigned)data.size(), 0, addr, tap_sink::ETHER_TYPE);
        }
        if (pkt->type() == 2) fhs(*pkt, out);
    } else if (first_run) {
        out += "lost clock!\n";
        pn->reset(out);
        discover(pkt, pn, out);                    // start rediscovery with this packet
    } else {
        out += "Giving up on queued packet!\n";
    }
}

void sniffer_handlers::discover(std::shared_ptr<classic_packet> pkt, std::shared_ptr<basic_rate_piconet> pn, std::string &out)
{
    out += "working on UAP/CLK1-6\n";
    pn->queue.push_back(pkt);                      // decoded once discovery completes
    if (pn->uap_from_header(*pkt, out)) recall(pn, out);
}

void sniffer_handlers::recall(std::shared_ptr<basic_rate_piconet> pn, std::string &out)
{
    out += "Decoding queued packets\n";
    while (!pn->queue.empty()) {
        auto pkt = pn->queue.front();
        pn->queue.pop_front();
        appendf(out, "time %6d, channel %2d, LAP %06x ", (int)pkt->clkn(), pkt->channel(), pkt->lap());
        decode(pkt, pn, false, out);
    }
    out += "Finished decoding queued packets\n";
}

void sniffer_handlers::fhs(classic_packet &pkt, std::string &out)
{
    const uint32_t lap = pkt.lap_from_fhs();
    const uint8_t uap = pkt.uap_from_fhs();
    const uint16_t nap = pkt.nap_from_fhs();
    const uint32_t clk = pkt.clock_from_fhs() << 1;            // units of 625 us
    const uint32_t offset = (clk - pkt.clkn()) & 0x7ffffff;
    appendf(out, "FHS contents: BD_ADDR %2.2x:%2.2x:%2.2x:%2.2x:%2.2x:%2.2x, CLK %07x\n", (nap >> 8) & 0xff, nap & 0xff, uap,
            (lap >> 16) & 0xff, (lap >> 8) & 0xff, lap & 0xff, clk);
    auto &slot = d_piconets[lap];
    if (!slot) slot = std::make_shared<basic_rate_piconet>(lap);
    slot->set_uap(uap);
    slot->set_nap(nap);
    slot->set_offset(offset);
}

}  // namespace host
}  // namespace bluetooth
}  // namespace gr

// the regenerated whitening tables by the reference's names (digest test against the reference's literals)
extern "C" int bt_host_lut(const char *name, uint8_t *out, int cap)
{
    const gr::bluetooth::host::whitening_tables &t = gr::bluetooth::host::wt();
    if (std::strcmp(name, "packet::WHITENING_DATA") == 0 && cap >= 127) { std::memcpy(out, t.seq, 127); return 127; }
    if (std::strcmp(name, "classic_packet::INDICES") == 0 && cap >= 64) { std::memcpy(out, t.start, 64); return 64; }
    return -1;
}

extern "C" int bt_host_crc_check(const uint8_t *symbols, int length, int clock, int type, int uap)
{
    btgpu_header none;
    std::memset(&none, 0, sizeof none);
    gr::bluetooth::host::classic_packet pkt(symbols, length, 0, 0, none);
    pkt.force_header(type, (uint8_t)uap);
    return pkt.crc_check(clock);
}

extern "C" int bt_host_decode_print(const uint8_t *symbols, int length, int uap, uint32_t clock, int have27, char *out, int cap)
{
    btgpu_header none;
    std::memset(&none, 0, sizeof none);
    gr::bluetooth::host::classic_packet pkt(symbols, length, 0, 0, none);
    pkt.set_uap((uint8_t)uap);
    pkt.set_clock(clock, have27 != 0);
    std::string text;
    pkt.decode(text);
    pkt.print(text);
    if (out && cap > 0) { std::strncpy(out, text.c_str(), (size_t)cap - 1); out[cap - 1] = 0; }
    return pkt.got_payload() ? 1 : 0;
}
