/*
 * bt_uap.c -- CPU ORACLE, classic header path (SURVEY.md section 8(f) rank 1 + the part of rank 2
 * that UAP discovery needs).  TEST INFRASTRUCTURE ONLY, like bt_oracle.c.
 *
 * Restates, from /root/reference (paths relative to it):
 *   lib/packet_impl.cc:367-383    classic_packet::unfec13
 *   lib/packet_impl.cc:386-468    classic_packet::unfec23   (with quirk Q12 below)
 *   lib/packet_impl.cc:513-526    classic_packet_impl::unwhiten (tables :84-90, :182-186 regenerated
 *                                 from their rule: x^7 + x^4 + 1, register position 6 = 1,
 *                                 positions 0..5 = CLK1..CLK6)
 *   lib/packet_impl.cc:529-548    classic_packet::crcgen
 *   lib/packet_impl.cc:597-609    classic_packet::UAP_from_hec
 *   lib/packet_impl.cc:612-671    classic_packet_impl::crc_check
 *   lib/packet_impl.cc:675-1043   payload_crc, fhs, decode_payload_header, DM, DH, EV3, EV4, EV5, HV
 *   lib/packet_impl.cc:1046-1063  classic_packet_impl::try_clock
 *   lib/piconet_impl.cc:433-517   basic_rate_piconet_impl::UAP_from_header (+ reset :526-547)
 *
 * Pinned by the known answer the compiled reference gave on samples/channel37.dem:
 * UAP 0xaf, CLK1-6 offset 38 after 3 packets -- tests/test_oracle_uap.py.
 *
 * Quirk Q12 (reproduced): in unfec23 the variable that collects the 5 syndrome bits still holds
 * the mismatch count (2..5) when the bits are shifted in, and it is shifted once more after the
 * last bit; as a uint8_t it can then never equal one of the ten single-error patterns, so a block
 * with two or more parity mismatches always fails and no data bit is ever corrected.
 */
#include "bt_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static uint8_t WH[127];
static uint8_t WH_INDEX[64];
static int wh_ready = 0;

static void wh_build(void)
{
    if (wh_ready) return;
    static const uint8_t seed[7] = {1, 1, 1, 0, 0, 0, 1};
    for (int i = 0; i < 7; i++) WH[i] = seed[i];
    for (int i = 7; i < 127; i++) WH[i] = WH[i - 7] ^ WH[i - 3];
    for (int clk = 0; clk < 64; clk++) {
        uint8_t p[7], s[7];
        for (int i = 0; i < 6; i++) p[i] = (clk >> i) & 1;
        p[6] = 1;
        for (int k = 0; k < 7; k++) {
            uint8_t o = p[6];
            s[k] = o;
            uint8_t q[7] = {o, p[0], p[1], p[2], (uint8_t)(p[3] ^ o), p[4], p[5]};
            memcpy(p, q, 7);
        }
        for (int i = 0; i < 127; i++) {
            int ok = 1;
            for (int k = 0; k < 7 && ok; k++) ok = WH[(i + k) % 127] == s[k];
            if (ok) { WH_INDEX[clk] = (uint8_t)i; break; }
        }
    }
    wh_ready = 1;
}

/* the regenerated tables by the reference's names (digest test, tests/golden/lut_digests.json) */
int bto_uap_lut(const char *name, uint8_t *out, int cap)
{
    wh_build();
    if (strcmp(name, "packet::WHITENING_DATA") == 0 && cap >= 127) { memcpy(out, WH, 127); return 127; }
    if (strcmp(name, "classic_packet::INDICES") == 0 && cap >= 64) { memcpy(out, WH_INDEX, 64); return 64; }
    return -1;
}

static unsigned air_bits(const char *air, int n)
{
    unsigned v = 0;
    for (int i = 0; i < n; i++) v |= ((unsigned)(air[i] & 1)) << i;
    return v;
}

static unsigned rev8(unsigned b)
{
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r |= ((b >> i) & 1) << (7 - i);
    return r;
}

/* :367-383 -- majority vote of three; ok iff fewer than length/4 triples disagreed */
int bto_unfec13(const char *in, char *out, int length)
{
    int be = 0;
    for (int i = 0; i < length; i++) {
        int a = in[3 * i] & 1, b = in[3 * i + 1] & 1, c = in[3 * i + 2] & 1;
        out[i] = (char)((a & b) | (b & c) | (c & a));
        be += ((a ^ b) | (b ^ c) | (c ^ a));
    }
    return be < (length / 4);
}

/* :386-468 -- (15,10) shortened Hamming, fecgen = {1,1,0,1,0,1}; `out` holds
 * ceil(length/10)*10 bits.  Returns 1, or 0 where the reference returns NULL. */
int bto_unfec23(const char *in, int length, char *out)
{
    static const uint8_t g[6] = {1, 1, 0, 1, 0, 1};
    if (length % 10) length += 10 - length % 10;
    int blocks = length / 10;
    for (int b = 0; b < blocks; b++) {
        const char *cw = in + 15 * b;
        char *o = out + 10 * b;
        for (int k = 0; k < 10; k++) o[k] = cw[k];
        /* parity of the ten data bits: the register of lfsr(data, 15, 10, g) (:278-307) */
        uint8_t reg[5] = {0, 0, 0, 0, 0};
        for (int i = 9; i >= 0; i--) {
            uint8_t fb = (uint8_t)((cw[i] & 1) ^ reg[4]);
            for (int j = 4; j > 0; j--) reg[j] = (uint8_t)(reg[j - 1] ^ (g[j] ? fb : 0));
            reg[0] = (uint8_t)(g[0] && fb);
        }
        int diff = 0;
        for (int k = 0; k < 5; k++) if (reg[k] != (uint8_t)(cw[10 + k] & 1)) diff++;
        if (diff <= 1) continue;
        /* Q12: the "correction" switch can never match -> failure */
        uint8_t d = (uint8_t)diff;
        for (int k = 0; k < 5; k++) { d |= (uint8_t)(reg[k] ^ (cw[10 + k] & 1)); d = (uint8_t)(d << 1); }
        int pos = -1;
        switch (d) {
            case 26: pos = 0; break; case 13: pos = 1; break; case 28: pos = 2; break; case 14: pos = 3; break;
            case 7: pos = 4; break;  case 25: pos = 5; break; case 22: pos = 6; break; case 11: pos = 7; break;
            case 31: pos = 8; break; case 21: pos = 9; break;
            default: return 0;
        }
        o[pos] ^= 1;
    }
    return 1;
}

/* :513-526 (d_whitened is always true for a classic packet, :238) */
void bto_unwhiten(const char *in, char *out, int clock, int length, int skip)
{
    wh_build();
    int index = (WH_INDEX[clock & 0x3f] + skip) % 127;
    for (int i = 0; i < length; i++) {
        out[i] = (char)((in[i] & 1) ^ WH[index]);
        index = (index + 1) % 127;
    }
}

/* :529-548 */
unsigned bto_crcgen(const char *payload, int length, int uap)
{
    unsigned reg = (rev8((unsigned)uap & 0xff) << 8) & 0xff00;
    for (int i = 0; i < length; i++) {
        reg = ((reg >> 1) | (((reg & 1) ^ ((unsigned)payload[i] & 1)) << 15)) & 0xffff;
        reg ^= (reg & 0x8000) >> 5;
        reg ^= (reg & 0x8000) >> 12;
    }
    return reg & 0xffff;
}

/* :597-609 */
int bto_uap_from_hec(unsigned data, unsigned hec)
{
    hec &= 0xff;
    for (int i = 9; i >= 0; i--) {
        if (hec & 0x80) hec ^= 0x65;
        hec = ((hec << 1) | (((hec >> 7) ^ (data >> i)) & 1)) & 0xff;
    }
    return (int)rev8(hec);
}

/* :1046-1063.  `symbols` starts at the access code (preamble).  Returns the UAP for this clock and
 * stores the packet type; when the 1/3 FEC of the header fails it returns 0 and leaves *type and
 * *uap as they were (the reference leaves d_packet_type / d_UAP untouched). */
int bto_try_clock(const char *symbols, int clock, int *type, int *uap)
{
    char header[18], plain[18];
    if (!bto_unfec13(symbols + 72, header, 18)) return 0;
    bto_unwhiten(header, plain, clock, 18, 0);
    unsigned data = air_bits(plain, 10), hec = air_bits(plain + 10, 8);
    *uap = bto_uap_from_hec(data, hec);
    *type = (int)air_bits(plain + 3, 4);
    return *uap;
}

/* ---- the packet object: what classic_packet_impl keeps between calls (lib/packet_impl.h) ---- */
struct bto_packet {
    char sym[3125 + 64];          /* d_symbols: zero-initialised, clipped copy (packet::packet :41-59) */
    int length;                   /* d_length */
    uint32_t clkn; int channel; uint32_t lap;
    int type, uap;                /* d_packet_type (0 at construction), d_UAP */
    uint32_t clock; int have_clk6, have_clk27, have_nap;
    int have_payload, payload_length, payload_header_length, llid, flow;
    char header[18];              /* d_packet_header */
    char payload[3000];           /* d_payload, one bit per byte */
};

bto_packet *bto_packet_new(const char *symbols, int length, uint32_t clkn, int channel)
{
    bto_packet *p = (bto_packet *)calloc(1, sizeof *p);
    if (length > 3125) length = 3125;
    memcpy(p->sym, symbols, (size_t)(length > 0 ? length : 0));
    p->length = length; p->clkn = clkn; p->channel = channel;
    p->lap = air_bits(&p->sym[38], 24);
    return p;
}
void bto_packet_free(bto_packet *p) { free(p); }
int  bto_packet_type(const bto_packet *p) { return p->type; }
int  bto_packet_uap(const bto_packet *p) { return p->uap; }

static int payload_crc_ok(const bto_packet *p)                                  /* :675-686 */
{
    /* EV4 (:946-1001) starts with payload_length = 1: the reference then reads the 16 check bits from
     * d_payload[-8 .. 7], eight bytes in front of the array (UB).  Policy: bits in front of the
     * payload read as 0 (and a negative length is an empty CRC input, as the reference's loop makes it). */
    int start = (p->payload_length - 2) * 8;
    unsigned crc = bto_crcgen(p->payload, start, p->uap);
    unsigned chk = 0;
    for (int i = 0; i < 16; i++)
        if (start + i >= 0) chk |= ((unsigned)(p->payload[start + i] & 1)) << i;
    return crc == chk;
}

static int pk_fhs(bto_packet *p, int clock)                                     /* :688-722 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126;
    p->payload_length = 20;
    if (size < p->payload_length * 12) return 1;
    char corrected[160];
    if (!bto_unfec23(stream, p->payload_length * 8, corrected)) return 0;
    bto_unwhiten(corrected, p->payload, clock, p->payload_length * 8, 18);
    if (payload_crc_ok(p)) return 1000;
    for (int c = 32; c < 64; c++) {
        bto_unwhiten(corrected, p->payload, c, p->payload_length * 8, 18);
        if (payload_crc_ok(p)) return 1000;
    }
    return 0;
}

/* :725-767 */
static int payload_header(bto_packet *p, const char *stream, int clock, int header_bytes, int size, int fec)
{
    char ph[16], corrected[20];
    if (header_bytes == 2) {
        if (size < 16) return 0;
        if (fec) {
            if (size < 30) return 0;
            if (!bto_unfec23(stream, 16, corrected)) return 0;
            bto_unwhiten(corrected, ph, clock, 16, 18);
        } else bto_unwhiten(stream, ph, clock, 16, 18);
        p->payload_length = (int)air_bits(&ph[3], 10) + 4;
    } else {
        if (size < 8) return 0;
        if (fec) {
            if (size < 15) return 0;
            if (!bto_unfec23(stream, 8, corrected)) return 0;
            bto_unwhiten(corrected, ph, clock, 8, 18);
        } else bto_unwhiten(stream, ph, clock, 8, 18);
        p->payload_length = (int)air_bits(&ph[3], 5) + 3;
    }
    p->llid = (int)air_bits(&ph[0], 2);
    p->flow = (int)air_bits(&ph[2], 1);
    p->payload_header_length = header_bytes;
    return 1;
}

static int pk_dm(bto_packet *p, int clock)                                      /* :770-830 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126, header_bytes = 2, max_length;
    switch (p->type) {
        case 8: stream += 80; size -= 80; header_bytes = 1; max_length = 12; break;
        case 3: header_bytes = 1; max_length = 20; break;
        case 10: max_length = 125; break;
        case 14: max_length = 228; break;
        default: return 0;
    }
    if (!payload_header(p, stream, clock, header_bytes, size, 1)) return 0;
    if (p->payload_length > max_length) return 1;
    int bitlength = p->payload_length * 8;
    if (bitlength > size) return 1;
    char *corrected = (char *)malloc((size_t)bitlength + 16);
    int ok = bto_unfec23(stream, bitlength, corrected);
    if (!ok) { free(corrected); return 0; }
    bto_unwhiten(corrected, p->payload, clock, bitlength, 18);
    free(corrected);
    return payload_crc_ok(p) ? 10 : 1;
}

static int pk_dh(bto_packet *p, int clock)                                      /* :834-884 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126, header_bytes = 2, max_length;
    switch (p->type) {
        case 9: case 4: header_bytes = 1; max_length = 30; break;
        case 11: max_length = 187; break;
        case 15: max_length = 343; break;
        default: return 0;
    }
    if (!payload_header(p, stream, clock, header_bytes, size, 0)) return 0;
    if (p->payload_length > max_length) return 1;
    int bitlength = p->payload_length * 8;
    if (bitlength > size) return 1;
    bto_unwhiten(stream, p->payload, clock, bitlength, 18);
    if (p->type == 9) return 1;
    return payload_crc_ok(p) ? 10 : 1;
}

static int pk_ev35(bto_packet *p, int clock, int maxlength)                     /* :886-915, :971-1000 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126;
    for (p->payload_length = 0; p->payload_length < maxlength; p->payload_length++) {
        int bits = p->payload_length * 8;
        if (bits + 8 > size) return 1;
        /* the reference unwhitens `stream` (not stream + bits) into d_payload + bits */
        bto_unwhiten(stream, p->payload + bits, clock, 8, 18 + bits);
        if (p->payload_length > 2 && payload_crc_ok(p)) return 10;
    }
    return 1;
}

static int pk_ev4(bto_packet *p, int clock)                                     /* :917-969 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126, syms = 0, bits = 0;
    p->payload_length = 1;
    while (syms < 1470) {
        char corrected[10];
        if (syms + 15 > size) return 1;
        if (!bto_unfec23(stream + syms, 10, corrected)) return syms < 45 ? 0 : 1;
        bto_unwhiten(corrected, p->payload + bits, clock, 10, 18 + bits);
        while (p->payload_length * 8 <= bits) {
            if (payload_crc_ok(p)) return 10;
            p->payload_length++;
        }
        syms += 15; bits += 10;
    }
    return 1;
}

static int pk_hv(bto_packet *p, int clock)                                      /* :1003-1043 */
{
    const char *stream = p->sym + 126;
    int size = p->length - 126;
    if (size < 240) { p->payload_length = 0; return 1; }
    switch (p->type) {
        case 5: {
            char corrected[80];
            if (!bto_unfec13(stream, corrected, 80)) return 0;
            p->payload_length = 10;
            bto_unwhiten(corrected, p->payload, clock, 80, 18);
            break;
        }
        case 6: {
            char corrected[160];
            if (!bto_unfec23(stream, 160, corrected)) return 0;
            p->payload_length = 20;
            bto_unwhiten(corrected, p->payload, clock, 160, 18);
            break;
        }
        case 7:
            p->payload_length = 30;
            bto_unwhiten(stream, p->payload, clock, 240, 18);
            break;
        default: break;
    }
    return 1;
}

/* :1046-1063 on the packet object */
int bto_packet_try_clock(bto_packet *p, int clock)
{
    return bto_try_clock(p->sym, clock, &p->type, &p->uap);
}

/* :612-671 -- 1 inconclusive, > 1 positive, 0 negative */
int bto_packet_crc_check(bto_packet *p, int clock)
{
    int r = 1;
    switch (p->type) {
        case 2: r = pk_fhs(p, clock); break;
        case 8: case 3: case 10: case 14: r = pk_dm(p, clock); break;
        case 4: case 11: case 15: r = pk_dh(p, clock); break;
        case 7: r = pk_ev35(p, clock, 32); break;
        case 12: r = pk_ev4(p, clock); break;
        case 13: r = pk_ev35(p, clock, 182); break;
        case 5: r = pk_hv(p, clock); break;
        default: break;
    }
    if (r == 0 && p->type != 2 && p->type != 3 && p->type != 5) return 1;
    if (r > 1 && (p->type == 7 || p->type == 13)) return 1;
    return r;
}

int bto_crc_check(const char *symbols, int length, int clock, int type, int uap)
{
    bto_packet *p = bto_packet_new(symbols, length, 0, 0);
    p->type = type; p->uap = uap;
    int r = bto_packet_crc_check(p, clock);
    bto_packet_free(p);
    return r;
}

/* ---- basic_rate_piconet_impl::UAP_from_header (lib/piconet_impl.cc:433-517) ---- */
void bto_piconet_init(bto_piconet *pn, uint32_t lap)
{
    memset(pn, 0, sizeof *pn);
    pn->lap = lap;
}

#define LOGF(...) do { if (log) { size_t n__ = strlen(log); if (n__ < cap) snprintf(log + n__, cap - n__, __VA_ARGS__); } } while (0)

static void pn_reset(bto_piconet *pn, char *log, size_t cap)                     /* :526-547 */
{
    LOGF("no candidates remaining! starting over . . .\n");
    if (pn->hop_reversal_inited) { bto_hopper_free(pn->hops); pn->hops = NULL; }
    pn->got_first_packet = 0;
    pn->packets_observed = 0;
    pn->hop_reversal_inited = 0;
    pn->have_uap = 0;
    pn->have_clk6 = 0;
    pn->have_clk27 = 0;
    /* two packets in a row on one channel were seen: try AFH next time (:541-546) */
    pn->afh = pn->looks_like_afh;
    pn->looks_like_afh = 0;
}
void bto_piconet_reset(bto_piconet *pn, char *log, size_t cap) { pn_reset(pn, log, cap); }
void bto_piconet_release(bto_piconet *pn) { if (pn->hops) { bto_hopper_free(pn->hops); pn->hops = NULL; } }
uint32_t bto_packet_lap(const bto_packet *p) { return p->lap; }
int bto_packet_header_present(const bto_packet *p) { return bto_header_present(p->sym, p->length); }

static int uap_from_header(bto_piconet *pn, bto_packet *pkt, char *log, size_t cap)
{
    int starting = 0, remaining = 0, first_clock = 0;
    uint32_t clkn = pkt->clkn;
    if (!pn->got_first_packet) pn->first_pkt_time = clkn;
    if (pn->packets_observed >= 1000) {                                          /* MAX_PATTERN_LENGTH */
        LOGF("Oops. More hops than we can remember.\n");
        pn_reset(pn, log, cap);
        return 0;
    }
    pn->pattern_indices[pn->packets_observed] = (int)(clkn - pn->first_pkt_time);
    pn->pattern_channels[pn->packets_observed] = (uint8_t)pkt->channel;
    pn->packets_observed++;
    pn->total_packets_observed++;
    for (int count = 0; count < 64; count++) {
        if (pn->clock6_candidates[count] > -1 || !pn->got_first_packet) {
            int clock = (int)((count + clkn - pn->first_pkt_time) % 64);
            starting++;
            int uap = bto_packet_try_clock(pkt, clock);
            int retval = -1;
            if (!pn->got_first_packet || uap == pn->clock6_candidates[count])
                retval = bto_packet_crc_check(pkt, clock);
            if (retval == -1 || retval == 0) pn->clock6_candidates[count] = -1;
            else if (retval == 1) { pn->clock6_candidates[count] = uap; first_clock = count; remaining++; }
            else {
                LOGF("Correct CRC! UAP = 0x%x found after %d total packets.\n", uap, pn->total_packets_observed);
                pn->clk_offset = (uint32_t)((count - (int)(pn->first_pkt_time & 0x3f)) & 0x3f);
                pn->uap = uap; pn->have_clk6 = 1; pn->have_uap = 1;
                pn->total_packets_observed = 0;
                return 1;
            }
        }
    }
    pn->got_first_packet = 1;
    LOGF("reduced from %d to %d CLK1-6 candidates\n", starting, remaining);
    if (remaining == 1) {
        pn->clk_offset = (uint32_t)((first_clock - (int)(pn->first_pkt_time & 0x3f)) & 0x3f);
        pn->uap = pn->clock6_candidates[first_clock];
        pn->have_clk6 = 1; pn->have_uap = 1;
        LOGF("We have a winner! UAP = 0x%x found after %d total packets.\n", pn->uap, pn->total_packets_observed);
        pn->total_packets_observed = 0;
        return 1;
    }
    if (remaining == 0) pn_reset(pn, log, cap);
    return 0;
}

int bto_piconet_uap_from_header_pkt(bto_piconet *pn, bto_packet *pkt, char *log, size_t cap)
{
    return uap_from_header(pn, pkt, log, cap);
}

int bto_uap_from_header(bto_piconet *pn, const char *symbols, int length, uint32_t clkn, int channel,
                        char *log, size_t cap)
{
    if (log && cap) log[0] = 0;
    bto_packet *pkt = bto_packet_new(symbols, length, clkn, channel);
    int r = uap_from_header(pn, pkt, log, cap);
    bto_packet_free(pkt);
    return r;
}

/* ---- packet::decode (:169-175), decode_header (:1066-1090), decode_payload (:1092-1165),
 *      print (:1168-1179) ---- */
static int decode_header(bto_packet *p, char *log, size_t cap)
{
    char header[18];
    if (p->have_clk6 && bto_unfec13(p->sym + 72, header, 18)) {
        bto_unwhiten(header, p->header, (int)p->clock, 18, 0);
        unsigned data = air_bits(p->header, 10), hec = air_bits(p->header + 10, 8);
        int uap = bto_uap_from_hec(data, hec);
        if (uap == p->uap) { p->type = (int)air_bits(&p->header[3], 4); return 1; }
        LOGF("bad HEC! %02x %02x %i ", uap, p->uap, (int)air_bits(&p->header[3], 4));
    }
    LOGF("failed to decode header\n");
    return 0;
}

static void decode_payload(bto_packet *p)
{
    int clk = (int)p->clock;
    p->payload_header_length = 0;
    switch (p->type) {
        case 0: case 1: p->payload_length = 0; break;
        case 2: pk_fhs(p, clk); break;
        case 3: pk_dm(p, clk); break;
        case 4: pk_dh(p, clk); break;
        case 5: case 6: pk_hv(p, clk); break;
        case 7: if (pk_ev35(p, clk, 32) <= 1) pk_hv(p, clk); break;
        case 8: pk_dm(p, clk); break;
        case 9: pk_dh(p, clk); break;
        case 10: pk_dm(p, clk); break;
        case 11: pk_dh(p, clk); break;
        case 12: pk_ev4(p, clk); break;
        case 13: pk_ev35(p, clk, 182); pk_dm(p, clk); break;      /* EV5 falls through into the DM5 case (Q11) */
        case 14: pk_dm(p, clk); break;
        case 15: pk_dh(p, clk); break;
    }
    p->have_payload = 1;
}

static const char *TYPE_NAME[16] = {"NULL", "POLL", "FHS", "DM1", "DH1/2-DH1", "HV1", "HV2/2-EV3", "HV3/EV3/3-EV3",
                                    "DV/3-DH1", "AUX1", "DM3/2-DH3", "DH3/3-DH3", "EV4/2-EV5", "EV5/3-EV5",
                                    "DM5/2-DH5", "DH5/3-DH5"};

/* ---- multi_sniffer_impl::ac / id / decode / discover / recall / fhs (lib/multi_sniffer_impl.cc:169-365),
 *      tun = false ---- */
#define MAX_PN 64
#define MAX_Q 1024
typedef struct { int used; bto_piconet pn; bto_packet *queue[MAX_Q]; int qn; uint32_t nap; int have_nap; } pn_slot;
struct bto_sniffer { pn_slot pn[MAX_PN]; int tun; uint8_t *tap; size_t tap_len, tap_cap; };

bto_sniffer *bto_sniffer_new(void) { return (bto_sniffer *)calloc(1, sizeof(bto_sniffer)); }

/* tun_format (lib/packet_impl.cc:1181-1210) and write_interface (lib/tun.cc:92-123): with `tun` on, every
 * frame the reference writes to its TAP device is appended here, preceded by its length (uint32 LE) */
void bto_sniffer_set_tun(bto_sniffer *s, int on) { s->tun = on; }
size_t bto_sniffer_tap(const bto_sniffer *s, uint8_t *out, size_t cap)
{
    size_t n = s->tap_len < cap ? s->tap_len : cap;
    if (out && n) memcpy(out, s->tap, n);
    return s->tap_len;
}
static void tap_write(bto_sniffer *s, const uint8_t *data, unsigned len, uint64_t src, uint64_t dst, unsigned ether_type)
{
    if (!s->tun) return;
    size_t need = s->tap_len + 4 + 14 + len;
    if (need > s->tap_cap) { s->tap_cap = need * 2 + 4096; s->tap = (uint8_t *)realloc(s->tap, s->tap_cap); }
    uint8_t *f = s->tap + s->tap_len;
    uint32_t n = 14 + len;
    f[0] = (uint8_t)n; f[1] = (uint8_t)(n >> 8); f[2] = (uint8_t)(n >> 16); f[3] = (uint8_t)(n >> 24);
    f += 4;
    for (int i = 0; i < 6; i++) { f[i] = (uint8_t)(dst >> (8 * (5 - i))); f[6 + i] = (uint8_t)(src >> (8 * (5 - i))); }
    f[12] = (uint8_t)(ether_type >> 8); f[13] = (uint8_t)ether_type;
    if (len) memcpy(f + 14, data, len);
    s->tap_len = need;
}
static unsigned packet_tun_format(const bto_packet *p, uint8_t *t)
{
    t[0] = (uint8_t)p->clock; t[1] = (uint8_t)(p->clock >> 8); t[2] = (uint8_t)(p->clock >> 16); t[3] = (uint8_t)(p->clock >> 24);
    t[4] = (uint8_t)p->channel;
    t[5] = (uint8_t)((p->have_clk27 ? 1 : 0) | ((p->have_nap ? 1 : 0) << 1));
    t[6] = (uint8_t)air_bits(&p->header[0], 7);
    t[7] = (uint8_t)air_bits(&p->header[7], 3);
    t[8] = (uint8_t)air_bits(&p->header[10], 8);
    for (int i = 0; i < p->payload_length; i++) t[9 + i] = (uint8_t)air_bits(&p->payload[i * 8], 8);
    return 9u + (unsigned)p->payload_length;
}
void bto_sniffer_free(bto_sniffer *s)
{
    if (!s) return;
    for (int i = 0; i < MAX_PN; i++) for (int k = 0; k < s->pn[i].qn; k++) bto_packet_free(s->pn[i].queue[k]);
    free(s->tap);
    free(s);
}

static pn_slot *pn_get(bto_sniffer *s, uint32_t lap, int create)
{
    for (int i = 0; i < MAX_PN; i++) if (s->pn[i].used && s->pn[i].pn.lap == lap) return &s->pn[i];
    if (!create) return NULL;
    for (int i = 0; i < MAX_PN; i++) if (!s->pn[i].used) {
        memset(&s->pn[i], 0, sizeof s->pn[i]);
        s->pn[i].used = 1; bto_piconet_init(&s->pn[i].pn, lap);
        return &s->pn[i];
    }
    return NULL;
}
static void pn_erase(bto_sniffer *s, uint32_t lap)
{
    pn_slot *q = pn_get(s, lap, 0);
    if (!q) return;
    for (int k = 0; k < q->qn; k++) bto_packet_free(q->queue[k]);
    q->used = 0; q->qn = 0;
}

static void sn_discover(bto_sniffer *s, pn_slot *q, bto_packet *pkt, char *log, size_t cap);

static void sn_fhs(bto_sniffer *s, bto_packet *pkt, char *log, size_t cap)         /* :324-365 */
{
    uint32_t lap = air_bits(&pkt->payload[34], 24);
    unsigned uap = air_bits(&pkt->payload[64], 8);
    unsigned nap = air_bits(&pkt->payload[72], 16) & 0xff;    /* nap_from_fhs uses air_to_host8(..., 16) (Q11) */
    uint32_t clk = air_bits(&pkt->payload[115], 26) << 1;
    uint32_t offset = (clk - pkt->clkn) & 0x7ffffff;
    LOGF("FHS contents: BD_ADDR %2.2x:%2.2x:%2.2x:%2.2x:%2.2x:%2.2x, CLK %07x\n", (nap >> 8) & 0xff, nap & 0xff, uap,
         (lap >> 16) & 0xff, (lap >> 8) & 0xff, lap & 0xff, clk);
    pn_slot *q = pn_get(s, lap, 1);
    if (!q) return;
    q->pn.uap = (int)uap; q->pn.have_uap = 1;
    q->nap = nap; q->have_nap = 1;
    q->pn.clk_offset = offset; q->pn.have_clk6 = 1; q->pn.have_clk27 = 1;
}

static void sn_decode(bto_sniffer *s, pn_slot *q, bto_packet *pkt, int first_run, char *log, size_t cap)   /* :235-280 */
{
    uint32_t clock = pkt->clkn + q->pn.clk_offset;
    pkt->clock = q->pn.have_clk27 ? (clock & 0x7ffffff) : (clock & 0x3f);
    pkt->have_clk6 = 1; pkt->have_clk27 = q->pn.have_clk27;
    pkt->uap = q->pn.uap;
    pkt->have_payload = 0;
    if (decode_header(pkt, log, cap)) decode_payload(pkt);
    if (pkt->have_payload) {
        LOGF("%s\n", TYPE_NAME[pkt->type & 15]);
        if (pkt->payload_header_length > 0)
            LOGF("  LLID: %d\n  flow: %d\n  payload length: %d\n", pkt->llid, pkt->flow, pkt->payload_length);
        if (s->tun) {                                                            /* :248-263 */
            uint64_t addr = ((uint64_t)(unsigned)pkt->uap << 24) | pkt->lap;
            if (q->have_nap) { addr |= (uint64_t)q->nap << 32; pkt->have_nap = 1; }
            uint8_t data[9 + 3000 / 8 + 8];
            unsigned n = packet_tun_format(pkt, data);
            tap_write(s, data, n, 0, addr, 0xFFF0);
        }
        if (pkt->type == 2) sn_fhs(s, pkt, log, cap);
        bto_packet_free(pkt);
    } else if (first_run) {
        LOGF("lost clock!\n");
        pn_reset(&q->pn, log, cap);
        sn_discover(s, q, pkt, log, cap);
    } else {
        LOGF("Giving up on queued packet!\n");
        bto_packet_free(pkt);
    }
}

static void sn_recall(bto_sniffer *s, pn_slot *q, char *log, size_t cap)           /* :303-319 */
{
    LOGF("Decoding queued packets\n");
    while (q->qn > 0) {
        bto_packet *pkt = q->queue[0];
        memmove(&q->queue[0], &q->queue[1], sizeof(q->queue[0]) * (size_t)(q->qn - 1));
        q->qn--;
        LOGF("time %6d, channel %2d, LAP %06x ", (int)pkt->clkn, pkt->channel, pkt->lap);
        sn_decode(s, q, pkt, 0, log, cap);
    }
    LOGF("Finished decoding queued packets\n");
}

static void sn_discover(bto_sniffer *s, pn_slot *q, bto_packet *pkt, char *log, size_t cap)   /* :286-297 */
{
    LOGF("working on UAP/CLK1-6\n");
    if (q->qn < MAX_Q) q->queue[q->qn++] = pkt;
    if (uap_from_header(&q->pn, pkt, log, cap)) sn_recall(s, q, log, cap);
}

/* one classic hit: `symbols` from the access code on, `len` symbols (lib/multi_sniffer_impl.cc:169-205).
 * Appends everything the reference prints for it to `log`. */
void bto_sniffer_ac(bto_sniffer *s, const char *symbols, int len, uint32_t clkn, int channel, double snr,
                    char *log, size_t cap)
{
    clkn &= 0x7ffffff;
    bto_packet *pkt = bto_packet_new(symbols, len, clkn, channel);
    uint32_t lap = pkt->lap;
    LOGF("time %6d, snr=%.1f, channel %2d, LAP %06x ", (int)clkn, snr, channel, lap);
    if (bto_header_present(pkt->sym, pkt->length)) {
        pn_slot *q = pn_get(s, lap, 1);
        if (!q) { bto_packet_free(pkt); return; }
        if (q->pn.have_clk6 && q->pn.have_uap) sn_decode(s, q, pkt, 1, log, cap);
        else sn_discover(s, q, pkt, log, cap);
        if (lap == 0x9e8b33 || lap == 0x9e8b00) pn_erase(s, lap);
    } else {
        LOGF("ID\n");
        tap_write(s, NULL, 0, 0, lap, 0xFFF0);                                   /* id(): :229-235 */
        bto_packet_free(pkt);
    }
}

/* decode() + print() of a packet with known UAP / clock (hopalong, lib/multi_hopper_impl.cc:181-186) */
int bto_packet_decode_print(bto_packet *p, int uap, uint32_t clock, int have27, char *log, size_t cap)
{
    p->uap = uap;
    p->clock = have27 ? (clock & 0x7ffffff) : (clock & 0x3f);
    p->have_clk6 = 1; p->have_clk27 = have27;
    p->have_payload = 0;
    if (decode_header(p, log, cap)) decode_payload(p);
    if (p->have_payload) {
        LOGF("%s\n", TYPE_NAME[p->type & 15]);
        if (p->payload_header_length > 0)
            LOGF("  LLID: %d\n  flow: %d\n  payload length: %d\n", p->llid, p->flow, p->payload_length);
    }
    return p->have_payload;
}

