/*
 * bt_hop.c -- CPU ORACLE, hop reversal.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, from /root/reference:
 *   lib/piconet_impl.cc:131-146   precalc (frequency register bank; the perm5 lookup table is an
 *                                 optimisation of the reference and not needed here)
 *   lib/piconet_impl.cc:149-167   address_precalc
 *   lib/piconet_impl.cc:179-211   perm5
 *   lib/piconet_impl.cc:214-255   gen_hops   (sequence index = CLK27..1, one entry per 625 us slot)
 *   lib/piconet_impl.cc:259-276   single_hop (clock = CLK27..0)
 *   lib/piconet_impl.cc:285-302   init_candidates
 *   lib/piconet_impl.cc:305-338   winnow(offset, channel)
 *   lib/piconet_impl.cc:520-523   aliased_channel
 *
 * PARITY UNPINNED: the reference holds no hop-sequence vectors (its tests are empty);
 * what is checked: gen_hops == single_hop on every index, structural properties of the
 * selection kernel (Bluetooth Core, Vol 2 Part B 2.6), and the GPU table against this file.
 */
#include "bt_oracle.h"
#include <stdlib.h>
#include <string.h>

#define HOP_CHANNELS 79
#define HOP_ALIASED  25

struct bto_hopper {
    int a1, b, c1, d1, e;
    int afh;
    int bank[HOP_CHANNELS];
    uint8_t *sequence;           /* BTO_SEQUENCE_LENGTH entries, built on demand */
    uint32_t *cand; int ncand;
};

static int perm5(int z, int p_high, int p_low)
{
    static const int i1[14] = {0, 2, 1, 3, 0, 1, 0, 3, 1, 0, 2, 1, 0, 1};
    static const int i2[14] = {1, 3, 2, 4, 4, 3, 2, 4, 4, 3, 4, 3, 3, 2};
    int zb[5], p[14];
    for (int i = 0; i < 9; i++) p[i] = (p_low >> i) & 1;
    for (int i = 0; i < 5; i++) p[i + 9] = (p_high >> i) & 1;
    for (int i = 0; i < 5; i++) zb[i] = (z >> i) & 1;
    for (int i = 13; i >= 0; i--)
        if (p[i]) { int t = zb[i1[i]]; zb[i1[i]] = zb[i2[i]]; zb[i2[i]] = t; }
    int out = 0;
    for (int i = 0; i < 5; i++) out += zb[i] << i;
    return out;
}

bto_hopper *bto_hopper_new(uint32_t address /* UAP<<24 | LAP, 28 bits used */, int afh)
{
    bto_hopper *h = (bto_hopper *)calloc(1, sizeof *h);
    address &= 0xfffffff;
    for (int i = 0; i < HOP_CHANNELS; i++) h->bank[i] = (i * 2) % HOP_CHANNELS;
    h->a1 = (address >> 23) & 0x1f;
    h->b = (address >> 19) & 0x0f;
    h->c1 = ((address >> 4) & 0x10) + ((address >> 3) & 0x08) + ((address >> 2) & 0x04) + ((address >> 1) & 0x02) +
            (address & 0x01);
    h->d1 = (address >> 10) & 0x1ff;
    h->e = ((address >> 7) & 0x40) + ((address >> 6) & 0x20) + ((address >> 5) & 0x10) + ((address >> 4) & 0x08) +
           ((address >> 3) & 0x04) + ((address >> 2) & 0x02) + ((address >> 1) & 0x01);
    h->afh = afh;
    return h;
}

void bto_hopper_free(bto_hopper *h)
{
    if (!h) return;
    free(h->sequence); free(h->cand); free(h);
}

/* single_hop (:259-276): clock = CLK27..0 */
int bto_single_hop(const bto_hopper *h, uint32_t clock)
{
    int x = (clock >> 2) & 0x1f, y1 = (clock >> 1) & 1, y2 = y1 << 5;
    int a = (h->a1 ^ (int)(clock >> 21)) & 0x1f;
    int c = (h->c1 ^ (int)(clock >> 16)) & 0x1f;
    int d = (h->d1 ^ (int)(clock >> 7)) & 0x1ff;
    int f = (int)((clock >> 3) & 0x1fffff0);
    return h->bank[(perm5(((x + a) % 32) ^ h->b, (y1 * 0x1f) ^ c, d) + h->e + f + y2) % HOP_CHANNELS];
}

/* gen_hops (:214-255), nested exactly like the reference so that it is an independent path from
 * single_hop; with AFH the odd entries repeat the even ones */
const uint8_t *bto_gen_hops(bto_hopper *hp)
{
    if (hp->sequence) return hp->sequence;
    hp->sequence = (uint8_t *)malloc(BTO_SEQUENCE_LENGTH);
    size_t index = 0;
    int f = 0;
    for (int h = 0; h < 4; h++)
        for (int i = 0; i < 0x20; i++) {
            int a = hp->a1 ^ i;
            for (int j = 0; j < 0x20; j++) {
                int c = hp->c1 ^ j, c_flipped = c ^ 0x1f;
                for (int k = 0; k < 0x200; k++) {
                    int d = hp->d1 ^ k;
                    for (int x = 0; x < 0x20; x++) {
                        int perm_in = ((x + a) % 32) ^ hp->b;
                        hp->sequence[index] = (uint8_t)hp->bank[(perm5(perm_in, c, d) + hp->e + f) % HOP_CHANNELS];
                        if (hp->afh) hp->sequence[index + 1] = hp->sequence[index];
                        else hp->sequence[index + 1] =
                                 (uint8_t)hp->bank[(perm5(perm_in, c_flipped, d) + hp->e + f + 32) % HOP_CHANNELS];
                        index += 2;
                    }
                    f += 16;
                }
            }
        }
    return hp->sequence;
}

int bto_aliased_channel(int channel) { return ((channel + 24) % HOP_ALIASED) + 26; }      /* :520-523 */

/* init_candidates (:285-302) */
int bto_hop_init_candidates(bto_hopper *h, int channel, int known_clock_bits, int aliased)
{
    const uint8_t *seq = bto_gen_hops(h);
    free(h->cand);
    h->cand = (uint32_t *)malloc(sizeof(uint32_t) * (BTO_SEQUENCE_LENGTH / 64 + 1));
    int count = 0;
    for (uint32_t i = (uint32_t)known_clock_bits; i < BTO_SEQUENCE_LENGTH; i += 0x40) {
        int obs = aliased ? bto_aliased_channel(seq[i]) : seq[i];
        if (obs == channel) h->cand[count++] = i;
    }
    h->ncand = count;
    return count;
}

/* winnow(offset, channel) (:305-321, the list part) */
int bto_hop_winnow(bto_hopper *h, int offset, int channel, int aliased)
{
    const uint8_t *seq = bto_gen_hops(h);
    int n = 0;
    for (int i = 0; i < h->ncand; i++) {
        uint32_t at = (uint32_t)(((uint64_t)h->cand[i] + (uint64_t)(uint32_t)offset) % BTO_SEQUENCE_LENGTH);
        int obs = aliased ? bto_aliased_channel(seq[at]) : seq[at];
        if (obs == channel) h->cand[n++] = h->cand[i];
    }
    h->ncand = n;
    return n;
}

int bto_hop_candidates(const bto_hopper *h, uint32_t *out, int cap)
{
    int n = h->ncand < cap ? h->ncand : cap;
    if (out && n > 0) memcpy(out, h->cand, sizeof(uint32_t) * (size_t)n);
    return h->ncand;
}

/* ---- basic_rate_piconet hop reversal on a bto_piconet ---- */
#include <stdio.h>
#define HLOG(...) do { if (log) { size_t n__ = strlen(log); if (n__ < cap) snprintf(log + n__, cap - n__, __VA_ARGS__); } } while (0)

int bto_piconet_init_hop_reversal(bto_piconet *pn, int aliased, char *log, size_t cap)       /* :96-129 */
{
    HLOG("\nCalculating complete hopping sequence.\n");
    if (pn->hops) bto_hopper_free(pn->hops);
    pn->hops = bto_hopper_new((((uint32_t)pn->uap << 24) | pn->lap) & 0xfffffff, pn->afh);
    int clock = (int)((pn->clk_offset + pn->first_pkt_time) & 0x3f);
    pn->num_candidates = bto_hop_init_candidates(pn->hops, pn->pattern_channels[0], clock, aliased);
    pn->winnowed = 0;
    pn->hop_reversal_inited = 1;
    pn->have_clk27 = 0;
    pn->aliased = aliased;
    HLOG("%d initial CLK1-27 candidates\n", pn->num_candidates);
    return pn->num_candidates;
}

static int pn_winnow_one(bto_piconet *pn, int offset, int channel, char *log, size_t cap)    /* :305-338 */
{
    int n = bto_hop_winnow(pn->hops, offset, channel, pn->aliased);
    pn->num_candidates = n;
    if (n == 1) {
        uint32_t c0 = 0;
        bto_hop_candidates(pn->hops, &c0, 1);
        pn->clk_offset = (c0 - pn->first_pkt_time) & 0x7ffffff;
        pn->have_clk27 = 1;
        HLOG("\nAcquired CLK1-27 offset = 0x%07x\n", pn->clk_offset);
    } else if (n == 0) {
        bto_piconet_reset(pn, log, cap);
    } else {
        HLOG("%d CLK1-27 candidates remaining\n", n);
    }
    return n;
}

int bto_piconet_winnow(bto_piconet *pn, char *log, size_t cap)                               /* :341-368 */
{
    int n = pn->num_candidates;
    for (; pn->winnowed < pn->packets_observed; pn->winnowed++) {
        int index = pn->pattern_indices[pn->winnowed], channel = pn->pattern_channels[pn->winnowed];
        n = pn_winnow_one(pn, index, channel, log, cap);
        if (!pn->hop_reversal_inited) break;                   /* reset() inside: the pattern is gone */
        if (pn->packets_observed > 0 && pn->winnowed > 0) {    /* the reference also reads entry -1 (not reproduced) */
            int last_index = pn->pattern_indices[pn->winnowed - 1], last_channel = pn->pattern_channels[pn->winnowed - 1];
            if (!pn->looks_like_afh && index == last_index + 1 && channel == last_channel) pn->looks_like_afh = 1;
        }
    }
    return n;
}

/* ---- gr::bluetooth::multi_hopper work() (lib/multi_hopper_impl.cc:76-209) ---- */
struct bto_hopper_block { uint32_t lap; int aliased; bto_piconet pn; };

bto_hopper_block *bto_hopper_block_new(uint32_t lap, int aliased)
{
    bto_hopper_block *b = (bto_hopper_block *)calloc(1, sizeof *b);
    b->lap = lap & 0xffffff; b->aliased = aliased;
    bto_piconet_init(&b->pn, b->lap);
    return b;
}
void bto_hopper_block_free(bto_hopper_block *b) { if (b) { bto_piconet_release(&b->pn); free(b); } }
const bto_piconet *bto_hopper_block_piconet(const bto_hopper_block *b) { return &b->pn; }

void bto_hopper_block_slot(bto_hopper_block *b, uint32_t clkn, int nhits, const int *channels, const char *const *symbols,
                           const int *lens, int low_channel, int high_channel, char *log, size_t cap)
{
    bto_piconet *pn = &b->pn;
    clkn &= 0x7ffffff;
    if (pn->have_clk27) {
        /* hopalong (:152-209): only the predicted channel of this slot */
        uint32_t clock27 = (clkn + pn->clk_offset) & 0x7ffffff;
        int hop = bto_gen_hops(pn->hops)[clock27];
        int obs = b->aliased ? bto_aliased_channel(hop) : hop;
        if (obs < low_channel || obs > high_channel) return;
        for (int i = 0; i < nhits; i++) {
            /* The reference tunes its DDC to the true hop frequency (:166); with an aliasing receiver
             * (25 Msps, every channel folded into 26..50) that offset is congruent, modulo the sample
             * rate, to the offset of the observed channel: the samples are those of channel `obs` (the
             * rotator increment and the tap phases differ by whole turns only). */
            if (channels[i] != obs) continue;
            bto_packet *pkt = bto_packet_new(symbols[i], lens[i], 0, obs);
            if (bto_packet_lap(pkt) == b->lap) {
                HLOG("clock 0x%07x, channel %2d: ", clock27, obs);
                if (bto_packet_header_present(pkt)) bto_packet_decode_print(pkt, pn->uap, clock27, 1, log, cap);
                else HLOG("ID\n");
            }
            bto_packet_free(pkt);
            break;
        }
        return;
    }
    for (int i = 0; i < nhits; i++) {                          /* channels ascending, first hit of each */
        bto_packet *pkt = bto_packet_new(symbols[i], lens[i], clkn, channels[i]);
        if (bto_packet_lap(pkt) == b->lap && bto_packet_header_present(pkt)) {
            if (!pn->have_clk6) {
                bto_piconet_uap_from_header_pkt(pn, pkt, log, cap);
                if (pn->have_clk6) {
                    bto_piconet_init_hop_reversal(pn, b->aliased, log, cap);
                    bto_piconet_winnow(pn, log, cap);
                }
            } else {
                bto_piconet_uap_from_header_pkt(pn, pkt, log, cap);
                if (pn->have_clk6) bto_piconet_winnow(pn, log, cap);
            }
            bto_packet_free(pkt);
            break;
        }
        bto_packet_free(pkt);
    }
}
