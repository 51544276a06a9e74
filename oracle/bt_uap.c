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

/* :386-468 -- (15,10) shortened Hamming, g(D) = D^5 + D^4 + D^2 + 1; `out` holds
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

/* ---- payload parsers; `s` = symbols from the access code on, `len` symbols in total ---- */
typedef struct { char bits[3000]; int length; /* bytes */ } payload_t;

static int payload_crc_ok(const payload_t *p, int uap)
{
    unsigned crc = bto_crcgen(p->bits, (p->length - 2) * 8, uap);
    unsigned chk = air_bits(&p->bits[(p->length - 2) * 8], 16);
    return crc == chk;
}

static int pk_fhs(const char *s, int len, int clock, int uap)                  /* :688-722 */
{
    const char *stream = s + 126;
    int size = len - 126;
    payload_t p; p.length = 20;
    if (size < p.length * 12) return 1;
    char corrected[160];
    if (!bto_unfec23(stream, p.length * 8, corrected)) return 0;
    bto_unwhiten(corrected, p.bits, clock, p.length * 8, 18);
    if (payload_crc_ok(&p, uap)) return 1000;
    for (int c = 32; c < 64; c++) {
        bto_unwhiten(corrected, p.bits, c, p.length * 8, 18);
        if (payload_crc_ok(&p, uap)) return 1000;
    }
    return 0;
}

/* :725-767; returns 0 on failure, else sets *plen (bytes incl. payload header and CRC) */
static int payload_header(const char *stream, int clock, int header_bytes, int size, int fec, int *plen)
{
    char ph[16], corrected[20];
    if (header_bytes == 2) {
        if (size < 16) return 0;
        if (fec) {
            if (size < 30) return 0;
            if (!bto_unfec23(stream, 16, corrected)) return 0;
            bto_unwhiten(corrected, ph, clock, 16, 18);
        } else bto_unwhiten(stream, ph, clock, 16, 18);
        *plen = (int)air_bits(&ph[3], 10) + 4;
    } else {
        if (size < 8) return 0;
        if (fec) {
            if (size < 15) return 0;
            if (!bto_unfec23(stream, 8, corrected)) return 0;
            bto_unwhiten(corrected, ph, clock, 8, 18);
        } else bto_unwhiten(stream, ph, clock, 8, 18);
        *plen = (int)air_bits(&ph[3], 5) + 3;
    }
    return 1;
}

static int pk_dm(const char *s, int len, int clock, int type, int uap)          /* :770-830 */
{
    const char *stream = s + 126;
    int size = len - 126, header_bytes = 2, max_length;
    switch (type) {
        case 8: stream += 80; size -= 80; header_bytes = 1; max_length = 12; break;
        case 3: header_bytes = 1; max_length = 20; break;
        case 10: max_length = 125; break;
        case 14: max_length = 228; break;
        default: return 0;
    }
    payload_t p;
    if (!payload_header(stream, clock, header_bytes, size, 1, &p.length)) return 0;
    if (p.length > max_length) return 1;
    int bitlength = p.length * 8;
    if (bitlength > size) return 1;
    char *corrected = (char *)malloc((size_t)bitlength + 16);
    int ok = bto_unfec23(stream, bitlength, corrected);
    if (!ok) { free(corrected); return 0; }
    bto_unwhiten(corrected, p.bits, clock, bitlength, 18);
    free(corrected);
    return payload_crc_ok(&p, uap) ? 10 : 1;
}

static int pk_dh(const char *s, int len, int clock, int type, int uap)          /* :834-884 */
{
    const char *stream = s + 126;
    int size = len - 126, header_bytes = 2, max_length;
    switch (type) {
        case 9: case 4: header_bytes = 1; max_length = 30; break;
        case 11: max_length = 187; break;
        case 15: max_length = 343; break;
        default: return 0;
    }
    payload_t p;
    if (!payload_header(stream, clock, header_bytes, size, 0, &p.length)) return 0;
    if (p.length > max_length) return 1;
    int bitlength = p.length * 8;
    if (bitlength > size) return 1;
    bto_unwhiten(stream, p.bits, clock, bitlength, 18);
    if (type == 9) return 1;
    return payload_crc_ok(&p, uap) ? 10 : 1;
}

static int pk_ev35(const char *s, int len, int clock, int uap, int maxlength)   /* :886-915, :971-1000 */
{
    const char *stream = s + 126;
    int size = len - 126;
    payload_t p;
    for (p.length = 0; p.length < maxlength; p.length++) {
        int bits = p.length * 8;
        if (bits + 8 > size) return 1;
        /* the reference unwhitens `stream` (not stream + bits) into d_payload + bits */
        bto_unwhiten(stream, p.bits + bits, clock, 8, 18 + bits);
        if (p.length > 2 && payload_crc_ok(&p, uap)) return 10;
    }
    return 1;
}

static int pk_ev4(const char *s, int len, int clock, int uap)                   /* :917-969 */
{
    const char *stream = s + 126;
    int size = len - 126, syms = 0, bits = 0;
    payload_t p; p.length = 1;
    while (syms < 1470) {
        char corrected[10];
        if (syms + 15 > size) return 1;
        if (!bto_unfec23(stream + syms, 10, corrected)) return syms < 45 ? 0 : 1;
        bto_unwhiten(corrected, p.bits + bits, clock, 10, 18 + bits);
        while (p.length * 8 <= bits) {
            if (payload_crc_ok(&p, uap)) return 10;
            p.length++;
        }
        syms += 15; bits += 10;
    }
    return 1;
}

static int pk_hv(const char *s, int len, int type)                              /* :1003-1043 */
{
    int size = len - 126;
    if (size < 240) return 1;
    if (type == 5) {                       /* crc_check only routes HV1 here */
        char corrected[80];
        if (!bto_unfec13(s + 126, corrected, 80)) return 0;
    }
    return 1;
}

/* :612-671 -- 1 inconclusive, > 1 positive, 0 negative */
int bto_crc_check(const char *symbols, int length, int clock, int type, int uap)
{
    int r = 1;
    if (length > 3125) length = 3125;                      /* packet::packet clips to MAX_SYMBOLS (:52-54) */
    switch (type) {
        case 2: r = pk_fhs(symbols, length, clock, uap); break;
        case 8: case 3: case 10: case 14: r = pk_dm(symbols, length, clock, type, uap); break;
        case 4: case 11: case 15: r = pk_dh(symbols, length, clock, type, uap); break;
        case 7: r = pk_ev35(symbols, length, clock, uap, 32); break;
        case 12: r = pk_ev4(symbols, length, clock, uap); break;
        case 13: r = pk_ev35(symbols, length, clock, uap, 182); break;
        case 5: r = pk_hv(symbols, length, type); break;
        default: break;
    }
    if (r == 0 && type != 2 && type != 3 && type != 5) return 1;
    if (r > 1 && (type == 7 || type == 13)) return 1;
    return r;
}

/* ---- basic_rate_piconet_impl::UAP_from_header (lib/piconet_impl.cc:433-517) ---- */
void bto_piconet_init(bto_piconet *pn, uint32_t lap)
{
    memset(pn, 0, sizeof *pn);
    pn->lap = lap;
}

static void pn_reset(bto_piconet *pn, char *log, size_t cap)                     /* :526-547 */
{
    if (log) snprintf(log + strlen(log), cap - strlen(log), "no candidates remaining! starting over . . .\n");
    pn->got_first_packet = 0;
    pn->packets_observed = 0;
    pn->have_uap = 0;
    pn->have_clk6 = 0;
}

int bto_uap_from_header(bto_piconet *pn, const char *symbols, int length, uint32_t clkn, int channel,
                        char *log, size_t cap)
{
    (void)channel;
    int starting = 0, remaining = 0, first_clock = 0;
    if (log && cap) log[0] = 0;
    /* packet::packet (:41-59): a zero-initialised 3125-symbol copy, clipped */
    char pkt[3125 + 64];
    memset(pkt, 0, sizeof pkt);
    if (length > 3125) length = 3125;
    memcpy(pkt, symbols, (size_t)length);
    symbols = pkt;
    if (!pn->got_first_packet) pn->first_pkt_time = clkn;
    if (pn->packets_observed >= 1000) {                                          /* MAX_PATTERN_LENGTH */
        if (log) snprintf(log + strlen(log), cap - strlen(log), "Oops. More hops than we can remember.\n");
        pn_reset(pn, log, cap);
        return 0;
    }
    pn->packets_observed++;
    pn->total_packets_observed++;
    int type = 0, uap_state = 0;                          /* d_packet_type = 0 (:46); d_UAP as left by try_clock */
    for (int count = 0; count < 64; count++) {
        if (pn->clock6_candidates[count] > -1 || !pn->got_first_packet) {
            int clock = (int)((count + clkn - pn->first_pkt_time) % 64);
            starting++;
            int uap = bto_try_clock(symbols, clock, &type, &uap_state);
            int retval = -1;
            if (!pn->got_first_packet || uap == pn->clock6_candidates[count])
                retval = bto_crc_check(symbols, length, clock, type, uap_state);
            if (retval == -1 || retval == 0) pn->clock6_candidates[count] = -1;
            else if (retval == 1) { pn->clock6_candidates[count] = uap; first_clock = count; remaining++; }
            else {
                if (log) snprintf(log + strlen(log), cap - strlen(log),
                                  "Correct CRC! UAP = 0x%x found after %d total packets.\n", uap, pn->total_packets_observed);
                pn->clk_offset = (count - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
                pn->uap = uap; pn->have_clk6 = 1; pn->have_uap = 1;
                pn->total_packets_observed = 0;
                return 1;
            }
        }
    }
    pn->got_first_packet = 1;
    if (log) snprintf(log + strlen(log), cap - strlen(log), "reduced from %d to %d CLK1-6 candidates\n", starting, remaining);
    if (remaining == 1) {
        pn->clk_offset = (first_clock - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
        pn->uap = pn->clock6_candidates[first_clock];
        pn->have_clk6 = 1; pn->have_uap = 1;
        if (log) snprintf(log + strlen(log), cap - strlen(log),
                          "We have a winner! UAP = 0x%x found after %d total packets.\n", pn->uap, pn->total_packets_observed);
        pn->total_packets_observed = 0;
        return 1;
    }
    if (remaining == 0) pn_reset(pn, log, cap);
    return 0;
}
