/* san_check.c -- the oracle under AddressSanitizer / UndefinedBehaviorSanitizer (SURVEY.md section 5:
 * the reference has out-of-bounds quirks, A.3 Q11, that the restatement must not inherit).  TEST
 * INFRASTRUCTURE like the rest of oracle/.  Build + run:  make -C oracle san
 *
 * Drives every public entry of bt_oracle.c / bt_uap.c (and the cheap part of bt_hop.c) with the captured
 * symbol stream of the reference (tests/golden/channel37.bits.npy), with random symbols of every length
 * around the parsers' boundaries, and with a short noise capture through the float front end.  Any
 * sanitizer report aborts (exit code != 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bt_oracle.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float gauss(void)
{
    float s = 0.f;
    for (int i = 0; i < 12; i++) s += (float)(rnd() & 0xffff) / 65536.0f;
    return s - 6.0f;
}

static char *load_bits(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char hdr[10];
    if (fread(hdr, 1, 10, f) != 10 || memcmp(hdr, "\x93NUMPY", 6) != 0) { fclose(f); return NULL; }
    size_t hlen = hdr[8] | (hdr[9] << 8);
    fseek(f, (long)(10 + hlen), SEEK_SET);
    long at = ftell(f);
    fseek(f, 0, SEEK_END);
    size_t nbytes = (size_t)(ftell(f) - at);
    fseek(f, at, SEEK_SET);
    unsigned char *packed = (unsigned char *)malloc(nbytes);
    if (fread(packed, 1, nbytes, f) != nbytes) { fclose(f); free(packed); return NULL; }
    fclose(f);
    char *bits = (char *)malloc(nbytes * 8);
    for (size_t i = 0; i < nbytes * 8; i++) bits[i] = (char)((packed[i >> 3] >> (7 - (i & 7))) & 1);   /* numpy.packbits: MSB first */
    free(packed);
    *n = nbytes * 8;
    return bits;
}

int main(int argc, char **argv)
{
    int fails = 0;
    /* ---- integer half on the reference's capture ---- */
    size_t n = 0;
    char *bits = load_bits(argc > 1 ? argv[1] : "../tests/golden/channel37.bits.npy", &n);
    if (bits) {
        if (n > 3997342) n = 3997342;
        bto_hit *hits = (bto_hit *)calloc(4096, sizeof(bto_hit));
        int nh = bto_scan_symbols(bits, n, hits, 4096);
        printf("channel37: %d hits\n", nh);
        if (nh != 33) fails++;
        /* packet handlers on every hit, exact-length slices (an over-read shows up as a heap overflow) */
        bto_sniffer *sn = bto_sniffer_new();
        char *log = (char *)malloc(1 << 16);
        for (int i = 0; i < nh; i++) {
            size_t off = (size_t)hits[i].offset, len = n - off < 3125 ? n - off : 3125;
            char *slice = (char *)malloc(len);
            memcpy(slice, bits + off, len);
            bto_sniffer_ac(sn, slice, (int)len, (uint32_t)(off / 625), 37, 20.0, log, 1 << 16);
            free(slice);
        }
        bto_sniffer_free(sn);
        free(log); free(hits);
        /* ragged ends of the searches */
        for (int len = 0; len < 140; len++) {
            char *s = (char *)malloc((size_t)len + 1);
            memcpy(s, bits + 66136 - 10, (size_t)len);
            if (len >= 68) (void)bto_sniff_ac(s, len - 67);
            if (len >= 64) { uint32_t lap; int e; (void)bto_btbb_find_ac(s, len - 63, 1, &lap, &e); }
            if (len >= 56) (void)bto_sniff_aa(s, len - 55, 2402e6);
            (void)bto_header_present(s, len);
            free(s);
        }
        free(bits);
    } else {
        printf("channel37 fixture not found: integer-half capture checks skipped\n");
    }
    /* ---- parsers on random symbols, every type, lengths around every boundary ---- */
    {
        char *log = (char *)malloc(1 << 16);
        for (int trial = 0; trial < 400; trial++) {
            int len = (int)(rnd() % 3200);
            if (trial < 140) len = trial;                          /* the short ones: header / FEC block edges */
            char *s = (char *)malloc((size_t)len + 1);
            for (int i = 0; i < len; i++) s[i] = (char)(rnd() & 1);
            bto_packet *p = bto_packet_new(s, len, rnd() & 0x7ffffff, (int)(rnd() % 79));
            if (p) {
                for (int c = 0; c < 64; c += 7) (void)bto_packet_try_clock(p, c);
                (void)bto_packet_crc_check(p, (int)(rnd() & 63));
                (void)bto_packet_decode_print(p, (int)(rnd() & 255), rnd() & 0x7ffffff, (int)(rnd() & 1), log, 1 << 16);
                bto_packet_free(p);
            }
            if (len > 0) (void)bto_le_print(s, len, 2402e6 + 2e6 * (double)(rnd() % 40), log, 1 << 16);
            free(s);
        }
        free(log);
    }
    /* ---- hop selection (closed form; the 128 MiB table is exercised by the regular tests) ---- */
    {
        bto_hopper *h = bto_hopper_new(0xaf24d952u & 0xfffffffu, 0);
        int seen[79] = {0};
        for (uint32_t clk = 0; clk < 200000; clk += 2) { int ch = bto_single_hop(h, clk); if (ch < 0 || ch > 78) fails++; else seen[ch]++; }
        for (int c = 0; c < 79; c++) if (!seen[c]) fails++;
        bto_hopper_free(h);
    }
    /* ---- float front end: noise (passes the default squelch, so clock recovery and both searches run) ---- */
    for (int mode = 0; mode < 2; mode++) {
        bto_ctx *c = bto_create(8e6, 2476.5e6, 10.0, mode);
        bto_set_le(c, 1);
        int slot = bto_samples_per_slot(c), S = 3;
        float *iq = (float *)malloc(sizeof(float) * 2 * (size_t)slot * S);
        for (int i = 0; i < 2 * slot * S; i++) iq[i] = 0.1f * gauss();
        bto_hit *hits = (bto_hit *)calloc(1024, sizeof(bto_hit));
        int done = 0;
        int nh = bto_run_stream(c, iq, (size_t)slot * S, hits, 1024, &done);
        printf("mode %d: %d slots, %d records on noise\n", mode, done, nh);
        if (done != S) fails++;
        free(hits); free(iq);
        bto_destroy(c);
    }
    printf(fails ? "san_check: %d FAILURES\n" : "san_check: ok\n", fails);
    return fails ? 1 : 0;
}
