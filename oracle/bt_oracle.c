/*
 * bt_oracle.c -- CPU ORACLE (test infrastructure; see bt_oracle.h for the
 * scope statement, the reference file:line map and the parity-pinning note).
 *
 * Build: make -C oracle   (gcc -O3 -march=native -ffp-contract=off -fopenmp)
 * -ffp-contract=off is REQUIRED: every fused multiply-add in this file is an
 * explicit fmaf(), so that the HIP kernels can reproduce the float results
 * bit for bit with __fmaf_rn/__fmul_rn/__fadd_rn.
 */
#include "bt_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SYMBOL_RATE                 1000000.0
#define SYMBOLS_PER_SLOT            625
#define SYMBOLS_FOR_HISTORY         3125          /* multi_block.h: SYMBOLS_FOR_BASIC_RATE_HISTORY */
#define SYMBOLS_PER_SHORTENED_AC    68            /* multi_block.h / packet.h:187                  */
#define SYMBOLS_PER_LE_PREAMBLE_AA  40
#define BASE_FREQUENCY              2402000000.0
#define CHANNEL_WIDTH               1000000.0
#define MMSE_NTAPS                  8
#define MMSE_NSTEPS                 128

typedef struct fir_bank {
    int    ntaps;        /* true length                                  */
    int    blk;          /* block length of the summation order = the decimation (ddc_run) */
    int    ntp;          /* padded to whole blocks: ceil(ntaps / blk) * blk */
    float *tr, *ti;      /* [nch][ntp] reversed complex taps, zero padded */
    double *foff;        /* [nch] frequency offset of each channel (Hz)  */
} fir_bank;

struct bto_ctx {
    double sample_rate, center_freq, target_snr;
    int    mode, mm_policy, le_enable, correlator;
    double sps;                 /* d_samples_per_symbol            */
    double samples_per_slot;    /* d_samples_per_slot (double)     */
    int    slot;                /* (int) d_samples_per_slot        */
    int    decim;               /* d_ddc_decimation_rate           */
    int    history;
    int    first_ch, first_noise;
    int    low_ch, high_ch, nch;
    int    ntaps_ch, ntaps_noise;
    float *h_ch, *h_noise;
    fir_bank ch_bank, noise_bank;
    float  demod_gain;
    /* mm_cr members (multi_block.h:87-93) */
    float  gain_mu, mu0, omega_relative_limit, omega0, gain_omega, omega_mid;
    float  d_mu, d_omega, d_last_sample;     /* live state */
    float  mmse[MMSE_NSTEPS + 1][MMSE_NTAPS];
    float  atan_tab[257];
};

/* ------------------------------------------------------------------------- */
/* exact phase factors                                                        */
/* ------------------------------------------------------------------------- */

/* e^{+j 2 pi k f/fs}, quadrant-exact when k f/fs is a multiple of 1/4 turn.
 * Policy Q3 (bt_oracle.h): GNU Radio keeps 2 pi f/fs as a float and lets the
 * rotator accumulate float error [EXT]; we use the mathematically exact phase. */
static void phase_factor(double f, double fs, long long k, float *re, float *im)
{
    double turns;
    int exact = 0, quad = 0;
    if (f == floor(f) && fs == floor(fs) && fabs(f) < 4e15 && fs > 0 && fs < 4e15) {
        __int128 num = (__int128)k * (__int128)(long long)f;
        __int128 den = (__int128)(long long)fs;
        __int128 r = num % den;
        if (r < 0) r += den;
        if ((4 * r) % den == 0) { exact = 1; quad = (int)((4 * r) / den); }
        turns = (double)(long long)r / (double)(long long)den;
    } else {
        turns = fmod((double)k * f / fs, 1.0);
        if (turns < 0) turns += 1.0;
    }
    if (exact) {
        static const float cr[4] = {1.f, 0.f, -1.f, 0.f}, ci[4] = {0.f, 1.f, 0.f, -1.f};
        *re = cr[quad]; *im = ci[quad];
    } else {
        double a = 2.0 * M_PI * turns;
        *re = (float)cos(a); *im = (float)sin(a);
    }
}

/* ------------------------------------------------------------------------- */
/* [EXT] gr::filter::firdes::low_pass(gain, fs, fc, tw, WIN_HANN)  (GNU Radio 3.7) */
/* ------------------------------------------------------------------------- */

int bto_firdes_ntaps(double fs, double tw)
{
    /* compute_ntaps: max_attenuation(WIN_HANN) = 44 dB */
    int ntaps = (int)(44.0 * fs / (22.0 * tw));
    if ((ntaps & 1) == 0) ntaps++;
    return ntaps;
}

int bto_firdes_low_pass(double gain, double fs, double fc, double tw, float *taps, int cap)
{
    int ntaps = bto_firdes_ntaps(fs, tw);
    if (!taps) return ntaps;
    if (cap < ntaps) return -1;
    int M = (ntaps - 1) / 2;
    double fwT0 = 2.0 * M_PI * fc / fs;
    float Mf = (float)(ntaps - 1);
    for (int n = -M; n <= M; n++) {
        /* window::hann stores 0.5 - 0.5 cos(2 pi n / M) as float */
        float w = (float)(0.5 - 0.5 * cos((2.0 * M_PI * (n + M)) / Mf));
        if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w);
        else        taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w);
    }
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    return ntaps;
}

/* ------------------------------------------------------------------------- */
/* [EXT] gr::fast_atan2f  (GNU Radio 3.7 gnuradio-runtime/lib/math/fast_atan2f.cc) */
/* 255-interval table of atan(i/255) with linear interpolation.                */
/* ------------------------------------------------------------------------- */

static void build_atan_table(float *t)
{
    for (int i = 0; i <= 255; i++) t[i] = (float)atan((double)i / 255.0);
    t[256] = t[255];
}

float bto_fast_atan2f(const bto_ctx *c, float y, float x)
{
    const float TAN_MAP_RES = 0.003921569f;   /* 1/255 */
    const float TAN_MAP_SIZE = 255.0f;
    const float *tab = c->atan_tab;
    float y_abs = fabsf(y), x_abs = fabsf(x), z, base_angle, angle;
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    if (y_abs < x_abs) z = y_abs / x_abs; else z = x_abs / y_abs;
    if (z < TAN_MAP_RES) {
        base_angle = z;
    } else {
        float alpha = z * TAN_MAP_SIZE;
        int index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        base_angle = tab[index];
        base_angle = base_angle + ((tab[index + 1] - tab[index]) * alpha);
    }
    if (x_abs > y_abs) {
        if (x >= 0.0f) angle = (y >= 0.0f) ? base_angle : -base_angle;
        else {
            angle = 3.14159265358979323846f;
            angle = (y >= 0.0f) ? (angle - base_angle) : (base_angle - angle);
        }
    } else {
        if (y >= 0.0f) {
            angle = 1.57079632679489661923f;
            angle = (x >= 0.0f) ? (angle - base_angle) : (angle + base_angle);
        } else {
            angle = -1.57079632679489661923f;
            angle = (x >= 0.0f) ? (angle + base_angle) : (angle - base_angle);
        }
    }
    return angle;
}

/* ------------------------------------------------------------------------- */
/* [EXT] gr::filter::mmse_fir_interpolator_ff  (8 taps, 128 steps)             */
/* Table regenerated: least-squares fractional-delay design over |f| <= 0.25   */
/* cycles/sample (8x8 normal equations), printed to 6 significant digits like  */
/* GNU Radio's interpolator_taps.h literals.  Row 1/128 reproduces the GNU     */
/* Radio literal row digit for digit (tests/test_oracle_float.py).             */
/* ------------------------------------------------------------------------- */

static int solve8(double A[8][9])
{
    for (int c = 0; c < 8; c++) {
        int p = c;
        for (int r = c + 1; r < 8; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (fabs(A[p][c]) < 1e-300) return -1;
        if (p != c) for (int k = 0; k < 9; k++) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
        for (int r = 0; r < 8; r++) if (r != c) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
        }
    }
    for (int r = 0; r < 8; r++) A[r][8] /= A[r][r];
    return 0;
}

static double bsinc(double B, double t)
{
    if (fabs(t) < 1e-12) return 2.0 * B;
    return sin(2.0 * M_PI * B * t) / (M_PI * t);
}

static void build_mmse_table(float tab[MMSE_NSTEPS + 1][MMSE_NTAPS])
{
    const double B = 0.25;
    for (int s = 0; s <= MMSE_NSTEPS; s++) {
        double mu = (double)s / MMSE_NSTEPS;
        double A[8][9];
        for (int k = 0; k < 8; k++) {
            for (int j = 0; j < 8; j++) A[k][j] = bsinc(B, (double)(k - j));
            A[k][8] = bsinc(B, 3.0 + mu - k);
        }
        solve8(A);
        for (int k = 0; k < 8; k++) {
            /* weight of sample k is table entry [7-k]; 6 significant digits */
            char buf[64];
            snprintf(buf, sizeof buf, "%.5e", A[k][8]);
            double v = strtod(buf, NULL);
            if (fabs(v) < 5e-7) v = 0.0;       /* rows 0 and 128 are exact unit taps */
            tab[s][7 - k] = (float)v;
        }
    }
}

float bto_mmse_interpolate(const bto_ctx *c, const float *in, float mu)
{
    int imu = (int)rint(mu * MMSE_NSTEPS);
    if (imu < 0) imu = 0;
    if (imu > MMSE_NSTEPS) imu = MMSE_NSTEPS;
    const float *t = c->mmse[imu];
    float acc = 0.0f;
    for (int k = 0; k < MMSE_NTAPS; k++) acc = fmaf(t[7 - k], in[k], acc);
    return acc;
}

/* ------------------------------------------------------------------------- */
/* construction: lib/multi_block.cc:40-120, :299-342                           */
/* ------------------------------------------------------------------------- */

static double channel_abs_freq(int ch) { return BASE_FREQUENCY + ch * CHANNEL_WIDTH; }

static void build_bank(fir_bank *b, const float *h, int ntaps, int nch, int low_ch,
                       double center, double fs, double extra, int blk)
{
    b->ntaps = ntaps;
    b->blk = blk;
    b->ntp = (ntaps + blk - 1) / blk * blk;
    b->tr = (float *)calloc((size_t)nch * b->ntp, sizeof(float));
    b->ti = (float *)calloc((size_t)nch * b->ntp, sizeof(float));
    b->foff = (double *)calloc(nch, sizeof(double));
    for (int c = 0; c < nch; c++) {
        double foff = channel_abs_freq(low_ch + c) + extra - center;
        b->foff[c] = foff;
        for (int k = 0; k < ntaps; k++) {
            /* [EXT] freq_xlating_fir_filter: ctaps[k] = h[k] * exp(j k theta) */
            float wr, wi;
            phase_factor(foff, fs, k, &wr, &wi);
            int j = ntaps - 1 - k;               /* stored reversed: y = sum_j t[j] x[base+j] */
            b->tr[(size_t)c * b->ntp + j] = h[k] * wr;
            b->ti[(size_t)c * b->ntp + j] = h[k] * wi;
        }
    }
}

static void free_bank(fir_bank *b) { free(b->tr); free(b->ti); free(b->foff); }

bto_ctx *bto_create(double sample_rate, double center_freq, double squelch_db, int mode)
{
    bto_ctx *c = (bto_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->target_snr = squelch_db;
    c->sample_rate = sample_rate;
    c->center_freq = center_freq;
    c->mode = mode;
    c->mm_policy = BTO_MM_WINDOWED_RESET;
    /* multi_LAP searches with libbtbb's btbb_find_ac (lib/multi_LAP_impl.cc:55,93), multi_sniffer
     * with the in-tree classic_packet::sniff_ac (lib/multi_sniffer_impl.cc:110) */
    c->correlator = (mode == BTO_MODE_LAP) ? BTO_CORRELATOR_BTBB : BTO_CORRELATOR_INTREE;
    c->le_enable = 0;

    int slots = 1;
    c->sps = sample_rate / SYMBOL_RATE;
    c->samples_per_slot = (int)SYMBOLS_PER_SLOT * c->sps;
    c->slot = (int)c->samples_per_slot;
    int history_required = (int)slots * c->samples_per_slot;

    c->ntaps_ch = bto_firdes_ntaps(sample_rate, 300000.0);
    c->h_ch = (float *)malloc(sizeof(float) * c->ntaps_ch);
    bto_firdes_low_pass(1.0, sample_rate, 500000.0, 300000.0, c->h_ch, c->ntaps_ch);
    c->ntaps_noise = bto_firdes_ntaps(sample_rate, 10000.0);
    c->h_noise = (float *)malloc(sizeof(float) * c->ntaps_noise);
    bto_firdes_low_pass(1.0, sample_rate, 22500.0, 10000.0, c->h_noise, c->ntaps_noise);

    c->decim = (int)c->sps / 2;
    if (c->decim < 1) c->decim = 1;
    double channel_sps = c->sps / c->decim;

    /* set_channels (:306-342) */
    double center = (center_freq - BASE_FREQUENCY) / CHANNEL_WIDTH;
    double bw = sample_rate / CHANNEL_WIDTH;
    double low_edge = center - bw / 2, high_edge = center + bw / 2;
    double min_w = 0.9;
    int lo = (int)(low_edge + min_w / 2 + 1);
    if (lo < 0) lo = 0;
    int hi = (int)(high_edge - min_w / 2);
    if (hi > 78) hi = 78;
    c->low_ch = lo; c->high_ch = hi;
    c->nch = hi >= lo ? hi - lo + 1 : 0;
    build_bank(&c->ch_bank, c->h_ch, c->ntaps_ch, c->nch, lo, center_freq, sample_rate, 0.0, c->decim);
    build_bank(&c->noise_bank, c->h_noise, c->ntaps_noise, c->nch, lo, center_freq, sample_rate,
               790000.0, c->decim);

    c->demod_gain = (float)(channel_sps / M_PI_2);
    c->gain_mu = 0.175f;
    c->mu0 = 0.32f;
    c->omega_relative_limit = 0.005f;
    c->omega0 = (float)channel_sps;
    c->gain_omega = (float)(.25 * c->gain_mu * c->gain_mu);
    c->omega_mid = c->omega0;
    c->d_mu = c->mu0; c->d_omega = c->omega0; c->d_last_sample = 0.0f;

    int channel_history = (int)(c->ntaps_ch + c->decim * MMSE_NTAPS);
    int noise_history = (int)c->ntaps_noise;
    if (channel_history > noise_history) {
        history_required += channel_history;
        c->first_ch = 0;
        c->first_noise = channel_history - noise_history;
    } else {
        history_required += noise_history;
        c->first_noise = 0;
        c->first_ch = noise_history - channel_history;
    }
    c->history = history_required;
    /* set_symbol_history (:299-303) */
    int nsym = (mode == BTO_MODE_SNIFFER) ? SYMBOLS_FOR_HISTORY : SYMBOLS_PER_SHORTENED_AC;
    c->history = (int)(c->history + (nsym * c->sps));

    build_atan_table(c->atan_tab);
    build_mmse_table(c->mmse);
    return c;
}

void bto_destroy(bto_ctx *c)
{
    if (!c) return;
    free(c->h_ch); free(c->h_noise);
    free_bank(&c->ch_bank); free_bank(&c->noise_bank);
    free(c);
}

void bto_set_mm_policy(bto_ctx *c, int p) { c->mm_policy = p; }
void bto_set_le(bto_ctx *c, int e) { c->le_enable = e; }
void bto_set_correlator(bto_ctx *c, int which) { c->correlator = which; }
int  bto_correlator(const bto_ctx *c) { return c->correlator; }
int bto_history(const bto_ctx *c) { return c->history; }
int bto_samples_per_slot(const bto_ctx *c) { return c->slot; }
int bto_decimation(const bto_ctx *c) { return c->decim; }
int bto_low_channel(const bto_ctx *c) { return c->low_ch; }
int bto_high_channel(const bto_ctx *c) { return c->high_ch; }
int bto_first_channel_sample(const bto_ctx *c) { return c->first_ch; }
int bto_first_noise_sample(const bto_ctx *c) { return c->first_noise; }
int bto_ntaps_channel(const bto_ctx *c) { return c->ntaps_ch; }
int bto_ntaps_noise(const bto_ctx *c) { return c->ntaps_noise; }
const float *bto_channel_taps(const bto_ctx *c) { return c->h_ch; }
const float *bto_noise_taps(const bto_ctx *c) { return c->h_noise; }
const float *bto_mmse_taps(const bto_ctx *c) { return &c->mmse[0][0]; }
const float *bto_atan_table(const bto_ctx *c) { return c->atan_tab; }

/* [EXT] gr::sync_decimator::fixed_rate_ninput_to_noutput(n) = max(0, n - history() + 1) / decimation,
 * and freq_xlating_fir_filter_ccf sets history() = ntaps.  The reference has already taken
 * (history() - 1) off before it asks (multi_block.cc:194), so the filter length comes off twice. */
static int sync_decimator_noutput(int ninput, int ntaps, int decim)
{
    int n = ninput - ntaps + 1;
    return (n < 0 ? 0 : n) / decim;
}
int bto_ddc_out(const bto_ctx *c)
{
    int ddc_samples = c->history - (c->ntaps_ch - 1) - c->first_ch;   /* multi_block.cc:194 */
    return sync_decimator_noutput(ddc_samples, c->ntaps_ch, c->decim); /* :200 */
}
int bto_noise_out(const bto_ctx *c)                                    /* :269 */
{
    return sync_decimator_noutput((int)c->samples_per_slot, c->ntaps_noise, c->decim);
}

/* ------------------------------------------------------------------------- */
/* [EXT] freq_xlating_fir_filter_ccf::work restated, fixed summation order     */
/* ------------------------------------------------------------------------- */

/* SUMMATION ORDER (this repository's to define: VOLK's in the reference is unspecified and GNU Radio is not in the image,
 * bt_oracle.h).  With the reversed taps t[j] and D = the decimation, the filter is taken in BLOCKS of D taps:
 *     G[q] = fmaf chain over r = 0 .. D-1 from +0, j = q D + r:
 *                re: fmaf(tr[j], xr, .) then fmaf(-ti[j], xi, .)     im: fmaf(ti[j], xr, .) then fmaf(tr[j], xi, .)
 *     y    = ((G[0] + G[1]) + G[2]) + ...      (q ascending; taps beyond the filter are exact zeros)
 * -- the polyphase-GEMM form of the decimating FIR: G[q][m] = sum_r t[qD + r] x[mD + r] is a matrix product over the input
 * reshaped into columns of D samples, and y[n] = sum_q G[q][n + q].  The product's exact stage computes it on the fp32 matrix
 * pipe (v_mfma_f32_32x32x2_f32 is bit for bit this fmaf chain, scripts/ubench/exact_mfma.hip), the generic kernels on the VALU.
 * xt: the window transposed into those columns, xt[(r * 2 + part) * ncol + m] = part of x[m D + r]: sixteen consecutive OUTPUTS then
 * read consecutive floats at every (q, r), and the loop over them is the one the compiler vectorises (each output its own chain). */
#define DDC_V 16
static void ddc_run(const fir_bank *b, int chan_idx, double fs, int decim,
                    const float *xt, int ncol, int first, int nout, float *out_iq)
{
    const int D = b->blk, nq = b->ntp / b->blk;
    const float *tr = b->tr + (size_t)chan_idx * b->ntp;
    const float *ti = b->ti + (size_t)chan_idx * b->ntp;
    const double foff = b->foff[chan_idx];
    /* sample first + i D + q D + r = column (i + q + a[r]), row rho[r] */
    int *ra = (int *)malloc(sizeof(int) * 2 * (size_t)D), *rho = ra + D;
    for (int r = 0; r < D; r++) { const int s0 = first + r; ra[r] = s0 / D; rho[r] = s0 - ra[r] * D; }
    (void)decim;
    for (int i0 = 0; i0 < nout; i0 += DDC_V) {
        float yr[DDC_V], yi[DDC_V];
        for (int q = 0; q < nq; q++) {
            float gr[DDC_V], gi[DDC_V];
            for (int l = 0; l < DDC_V; l++) { gr[l] = 0.0f; gi[l] = 0.0f; }
            for (int r = 0; r < D; r++) {
                const float ta = tr[q * D + r], tb = ti[q * D + r];
                const float *pr = xt + (size_t)(2 * rho[r]) * ncol + i0 + ra[r] + q;
                const float *pi = xt + (size_t)(2 * rho[r] + 1) * ncol + i0 + ra[r] + q;
                for (int l = 0; l < DDC_V; l++) {
                    const float vr = pr[l], vi = pi[l];
                    gr[l] = fmaf(ta, vr, gr[l]);
                    gr[l] = fmaf(-tb, vi, gr[l]);
                    gi[l] = fmaf(tb, vr, gi[l]);
                    gi[l] = fmaf(ta, vi, gi[l]);
                }
            }
            if (q == 0) for (int l = 0; l < DDC_V; l++) { yr[l] = gr[l]; yi[l] = gi[l]; }
            else for (int l = 0; l < DDC_V; l++) { yr[l] = yr[l] + gr[l]; yi[l] = yi[l] + gi[l]; }
        }
        for (int l = 0; l < DDC_V && i0 + l < nout; l++) {
            const int i = i0 + l;
            /* rotator: out[i] = y[i] * exp(-j theta D i), restarted per window (Q3) */
            float rr, ri;
            phase_factor(-foff, fs, (long long)decim * i, &rr, &ri);
            out_iq[2 * i]     = fmaf(-yi[l], ri, yr[l] * rr);
            out_iq[2 * i + 1] = fmaf(yi[l], rr, yr[l] * ri);
        }
    }
    free(ra);
}

/* the window in columns of D samples (see ddc_run); zero columns behind the end */
static int transpose_cols(const float *win, int n, int D, int extra_cols, float **xt)
{
    const int ncol = (n + D - 1) / D + extra_cols;
    *xt = (float *)calloc((size_t)2 * D * ncol, sizeof(float));
    for (int i = 0; i < n; i++) {
        const int m = i / D, r = i - m * D;
        (*xt)[(size_t)(2 * r) * ncol + m] = win[2 * i];
        (*xt)[(size_t)(2 * r + 1) * ncol + m] = win[2 * i + 1];
    }
    return ncol;
}

static double mean_mag2(const float *iq, int n)
{
    /* complex_to_mag_squared (float) then double accumulate: multi_block.cc:206-218 */
    double e = 0.0;
    for (int i = 0; i < n; i++) {
        float m = (iq[2 * i] * iq[2 * i]) + (iq[2 * i + 1] * iq[2 * i + 1]);
        e += m;
    }
    return e / n;
}

static int channel_samples_d(bto_ctx *c, int channel, const float *xt, int ncol,
                             float *out_iq, double *energy)
{
    int idx = channel - c->low_ch;
    if (idx < 0 || idx >= c->nch) { *energy = 1.0; return 0; }      /* multi_block.cc:223-225 */
    int nout = bto_ddc_out(c);
    ddc_run(&c->ch_bank, idx, c->sample_rate, c->decim, xt, ncol, c->first_ch, nout, out_iq);
    *energy = mean_mag2(out_iq, nout);
    return nout;
}

static int check_snr_d(bto_ctx *c, int channel, double on_energy, const float *xt, int ncol,
                       double *snr, double *off_energy)
{
    int idx = channel - c->low_ch;
    double off = 0.0;
    if (idx < 0 || idx >= c->nch) off = 1.0;                         /* :288-290 */
    else {
        int nout = bto_noise_out(c);
        float *tmp = (float *)malloc(sizeof(float) * 2 * (size_t)nout);
        ddc_run(&c->noise_bank, idx, c->sample_rate, c->decim, xt, ncol, c->first_noise, nout, tmp);
        off = mean_mag2(tmp, nout);
        free(tmp);
    }
    if (off_energy) *off_energy = off;
    *snr = 10.0 * log10(on_energy / off);
    return *snr >= c->target_snr;
}

int bto_channel_samples(bto_ctx *c, int channel, const float *win, float *out_iq, double *energy)
{
    float *xt;
    int ncol = transpose_cols(win, c->history, c->decim, DDC_V + 2, &xt);
    int n = channel_samples_d(c, channel, xt, ncol, out_iq, energy);
    free(xt);
    return n;
}

int bto_check_snr(bto_ctx *c, int channel, double on_energy, const float *win, double *snr,
                  double *off_energy)
{
    float *xt;
    int ncol = transpose_cols(win, c->history, c->decim, DDC_V + 2, &xt);
    int r = check_snr_d(c, channel, on_energy, xt, ncol, snr, off_energy);
    free(xt);
    return r;
}

/* ------------------------------------------------------------------------- */
/* demod / M&M / slicer: lib/multi_block.cc:123-178, :230-251                  */
/* ------------------------------------------------------------------------- */

void bto_demod(const bto_ctx *c, const float *iq, float *out, int n)
{
    if (n > 0) out[0] = 0.0f;                       /* policy Q1 */
    for (int i = 1; i < n; i++) {
        float ar = iq[2 * i], ai = iq[2 * i + 1], br = iq[2 * i - 2], bi = iq[2 * i - 1];
        /* in[i] * conj(in[i-1]) */
        float pr = fmaf(ai, bi, ar * br);
        float pi = fmaf(ai, br, -(ar * bi));
        out[i] = c->demod_gain * bto_fast_atan2f(c, pi, pr);
    }
}

static inline float slice(float x) { return (x < 0) ? -1.0F : 1.0F; }

static inline float branchless_clip(float x, float clip)
{
    /* [EXT] gr::branchless_clip */
    float x1 = fabsf(x + clip);
    float x2 = fabsf(x - clip);
    x1 -= x2;
    return 0.5f * x1;
}

int bto_mm_cr(bto_ctx *c, const float *in, int ninput_items, float *out, int noutput_items)
{
    unsigned int ii = 0;
    int oo = 0;
    unsigned int ni = (unsigned int)(ninput_items - MMSE_NTAPS);
    if (ninput_items < MMSE_NTAPS) return 0;      /* the reference would wrap; windows are never this short */
    float mm_val;
    while ((oo < noutput_items) && (ii < ni)) {
        out[oo] = bto_mmse_interpolate(c, &in[ii], c->d_mu);
        mm_val = slice(c->d_last_sample) * out[oo] - slice(out[oo]) * c->d_last_sample;
        c->d_last_sample = out[oo];
        c->d_omega = c->d_omega + (c->gain_omega * mm_val);
        c->d_omega = c->omega_mid + branchless_clip(c->d_omega - c->omega_mid, c->omega_relative_limit);
        c->d_mu = c->d_mu + (c->d_omega + (c->gain_mu * mm_val));
        float fl = floorf(c->d_mu);
        ii += (int)fl;
        c->d_mu = c->d_mu - fl;
        oo++;
    }
    return oo;
}

int bto_channel_symbols(bto_ctx *c, const float *iq, int ninput_items, char *symbols, float *soft)
{
    int demod_n = ninput_items - 1;
    if (demod_n <= 0) return 0;
    float *demod_out = (float *)malloc(sizeof(float) * demod_n);
    float *cr_out = (float *)malloc(sizeof(float) * demod_n);
    bto_demod(c, iq, demod_out, demod_n);
    if (c->mm_policy == BTO_MM_WINDOWED_RESET) {    /* policy Q2 */
        c->d_mu = c->mu0; c->d_omega = c->omega0; c->d_last_sample = 0.0f;
    }
    int n = bto_mm_cr(c, demod_out, demod_n, cr_out, demod_n);
    for (int i = 0; i < n; i++) symbols[i] = (cr_out[i] < 0) ? 0 : 1;
    if (soft) memcpy(soft, cr_out, sizeof(float) * n);
    free(demod_out); free(cr_out);
    return n;
}

/* ------------------------------------------------------------------------- */
/* access code: lib/packet_impl.cc:278-364 (lfsr/acgen), :471-510 (check_ac),  */
/* :247-268 (sniff_ac).  Restated from the Bluetooth BCH(64,30) definition.    */
/* ------------------------------------------------------------------------- */

uint32_t bto_air_to_host32(const char *air, int bits)
{
    uint32_t h = 0;
    for (int i = 0; i < bits; i++) h |= ((uint32_t)(air[i] & 1)) << i;
    return h;
}

/* 72 air-order bits of the access code for a LAP.
 * Layout: [0..3] preamble, [4..37] 34 parity bits, [38..61] LAP
 * LSB first, [62..67] Barker, [68..71] trailer.
 * Sync word = BCH(64,30) codeword of (LAP || Barker) pre-scrambled with the PN
 * sequence 0x83848D96BBCC54FC, generator g(D) = 0260534236651 (octal), then
 * scrambled again with the full PN (Bluetooth Core, Baseband 6.3.3). */
static void ac_bits(uint32_t lap, uint8_t bits[72])
{
    /* generator, degree 34, coefficient of D^0 first (same ordering as packet_impl.cc:318) */
    static const uint8_t g[35] = {1,0,0,1,0,1,0,1,1,0,1,1,1,1,0,0,1,0,0,0,1,1,1,0,1,0,1,0,0,0,0,1,1,0,1};
    /* PN, transmitted order p0..p63 = LSB-first bits of 0x83848D96BBCC54FC */
    const uint64_t PN = 0x83848D96BBCC54FCULL;
    uint8_t pn[64], info[30], cw[64];
    for (int i = 0; i < 64; i++) pn[i] = (PN >> i) & 1;
    /* information bits in transmitted order x34..x63: 24 LAP bits LSB first, then Barker */
    for (int i = 0; i < 24; i++) info[i] = (lap >> i) & 1;
    static const uint8_t barker0[6] = {0,0,1,1,0,1};   /* LAP bit23 == 0 */
    static const uint8_t barker1[6] = {1,1,0,0,1,0};   /* LAP bit23 == 1 */
    const uint8_t *bk = ((lap >> 23) & 1) ? barker1 : barker0;
    for (int i = 0; i < 6; i++) info[24 + i] = bk[i];
    /* pre-scramble the information bits with p34..p63 */
    uint8_t x[30];
    for (int i = 0; i < 30; i++) x[i] = info[i] ^ pn[34 + i];
    /* systematic encoding: parity(D) = D^34 x(D) mod g(D); x[29] is the highest power */
    uint8_t reg[34];
    memset(reg, 0, sizeof reg);
    for (int i = 29; i >= 0; i--) {
        uint8_t fb = x[i] ^ reg[33];
        for (int j = 33; j > 0; j--) reg[j] = reg[j - 1] ^ (fb & g[j]);
        reg[0] = fb & g[0];
    }
    for (int i = 0; i < 34; i++) cw[i] = reg[i];
    for (int i = 0; i < 30; i++) cw[34 + i] = x[i];
    /* scramble with the full PN -> sync word in transmitted order */
    for (int i = 0; i < 64; i++) cw[i] ^= pn[i];
    /* preamble: alternate into sync bit 0; trailer: alternate out of sync bit 63 */
    if (cw[0]) { bits[0] = 1; bits[1] = 0; bits[2] = 1; bits[3] = 0; }
    else       { bits[0] = 0; bits[1] = 1; bits[2] = 0; bits[3] = 1; }
    for (int i = 0; i < 64; i++) bits[4 + i] = cw[i];
    if (cw[63]) { bits[68] = 0; bits[69] = 1; bits[70] = 0; bits[71] = 1; }
    else        { bits[68] = 1; bits[69] = 0; bits[70] = 1; bits[71] = 0; }
}

void bto_acgen(uint32_t lap, uint8_t ac[9])
{
    uint8_t bits[72];
    ac_bits(lap & 0xffffff, bits);
    for (int b = 0; b < 9; b++) {
        uint8_t v = 0;
        for (int i = 0; i < 8; i++) v = (uint8_t)((v << 1) | bits[8 * b + i]);   /* MSB first */
        ac[b] = v;
    }
}

int bto_ac_errors(const char *stream, uint32_t lap)
{
    uint8_t bits[72];
    ac_bits(lap & 0xffffff, bits);
    int e = 0;
    for (int i = 0; i < SYMBOLS_PER_SHORTENED_AC; i++) if (bits[i] != (uint8_t)stream[i]) e++;
    return e;
}

int bto_check_ac(const char *stream, uint32_t lap)
{
    return bto_ac_errors(stream, lap) < 7;       /* packet_impl.cc:494 rejects at >= 7 */
}

/* distance LUTs = min Hamming distance to a valid set */
static int popc(unsigned v) { return __builtin_popcount(v); }
static int min_dist(unsigned v, const unsigned *valid, int n)
{
    int best = 99;
    for (int i = 0; i < n; i++) { int d = popc(v ^ valid[i]); if (d < best) best = d; }
    return best;
}

static uint8_t PREAMBLE_DISTANCE[32], BARKER_DISTANCE[128];
static uint8_t LE_PREAMBLE_DISTANCE[512], LE_AA_DISTANCE[4][256];
static uint8_t LE_ACC_HDR_LSB[256], LE_ACC_HDR_MSB[256], LE_DATA_HDR_LSB[256], LE_DATA_HDR_MSB[256];
static uint8_t WHITENING[127], LE_INDICES[40];
static int luts_ready = 0;

static void build_luts(void)
{
    if (luts_ready) return;
    const unsigned pre[2] = {0x0a, 0x15}, bar[2] = {0x27, 0x58};
    for (unsigned v = 0; v < 32; v++) PREAMBLE_DISTANCE[v] = (uint8_t)min_dist(v, pre, 2);
    for (unsigned v = 0; v < 128; v++) BARKER_DISTANCE[v] = (uint8_t)min_dist(v, bar, 2);
    const unsigned lepre[2] = {0x0aa, 0x155};
    for (unsigned v = 0; v < 512; v++) LE_PREAMBLE_DISTANCE[v] = (uint8_t)min_dist(v, lepre, 2);
    const unsigned aa[4] = {0xd6, 0xbe, 0x89, 0x8e};   /* advertising AA 0x8E89BED6, LSB byte first */
    for (int b = 0; b < 4; b++)
        for (unsigned v = 0; v < 256; v++) LE_AA_DISTANCE[b][v] = (uint8_t)popc(v ^ aa[b]);
    unsigned set[256]; int n;
    n = 0; for (unsigned t = 0; t <= 6; t++) { set[n++] = t; set[n++] = 0xc0 | t; }
    for (unsigned v = 0; v < 256; v++) LE_ACC_HDR_LSB[v] = (uint8_t)min_dist(v, set, n);
    n = 0; for (unsigned t = 0x06; t <= 0x24; t++) set[n++] = t;
    for (unsigned v = 0; v < 256; v++) LE_ACC_HDR_MSB[v] = (uint8_t)min_dist(v, set, n);
    n = 0; for (unsigned t = 0; t < 0x20; t++) if (t & 3) set[n++] = t;
    for (unsigned v = 0; v < 256; v++) LE_DATA_HDR_LSB[v] = (uint8_t)min_dist(v, set, n);
    n = 0; for (unsigned t = 0; t < 0x20; t++) set[n++] = t;
    for (unsigned v = 0; v < 256; v++) LE_DATA_HDR_MSB[v] = (uint8_t)min_dist(v, set, n);
    /* whitening sequence: w[n] = w[n-7] ^ w[n-3], seed 1110001 (x^7 + x^4 + 1) */
    static const uint8_t seed[7] = {1,1,1,0,0,0,1};
    for (int i = 0; i < 7; i++) WHITENING[i] = seed[i];
    for (int i = 7; i < 127; i++) WHITENING[i] = WHITENING[i - 7] ^ WHITENING[i - 3];
    /* LE start index per channel index: position in the sequence where the LE
     * whitening LFSR (pos0 = 1, pos1..6 = channel index MSB first) starts */
    for (int ch = 0; ch < 40; ch++) {
        uint8_t p[7], s[7];
        p[0] = 1;
        for (int i = 0; i < 6; i++) p[1 + i] = (ch >> (5 - i)) & 1;
        for (int k = 0; k < 7; k++) {
            uint8_t o = p[6];
            s[k] = o;
            uint8_t q[7] = {o, p[0], p[1], p[2], (uint8_t)(p[3] ^ o), p[4], p[5]};
            memcpy(p, q, 7);
        }
        for (int i = 0; i < 127; i++) {
            int ok = 1;
            for (int k = 0; k < 7 && ok; k++) ok = WHITENING[(i + k) % 127] == s[k];
            if (ok) { LE_INDICES[ch] = (uint8_t)i; break; }
        }
    }
    luts_ready = 1;
}

/* The regenerated tables by the reference's names (lib/packet_impl.cc:84-90, :188-197, :1316-1450),
 * for the digest test against the reference's literals (tests/golden/lut_digests.json). */
int bto_lut(const char *name, uint8_t *out, int cap)
{
    build_luts();
    static const struct { const char *name; const uint8_t *p; int n; } T[] = {
        {"packet::WHITENING_DATA", WHITENING, 127},
        {"classic_packet::PREAMBLE_DISTANCE", PREAMBLE_DISTANCE, 32},
        {"classic_packet::BARKER_DISTANCE", BARKER_DISTANCE, 128},
        {"le_packet::PREAMBLE_DISTANCE", LE_PREAMBLE_DISTANCE, 512},
        {"le_packet::ACCESS_ADDRESS_DISTANCE_0", LE_AA_DISTANCE[0], 256},
        {"le_packet::ACCESS_ADDRESS_DISTANCE_1", LE_AA_DISTANCE[1], 256},
        {"le_packet::ACCESS_ADDRESS_DISTANCE_2", LE_AA_DISTANCE[2], 256},
        {"le_packet::ACCESS_ADDRESS_DISTANCE_3", LE_AA_DISTANCE[3], 256},
        {"le_packet::ACCESS_HEADER_DISTANCE_LSB", LE_ACC_HDR_LSB, 256},
        {"le_packet::ACCESS_HEADER_DISTANCE_MSB", LE_ACC_HDR_MSB, 256},
        {"le_packet::DATA_HEADER_DISTANCE_LSB", LE_DATA_HDR_LSB, 256},
        {"le_packet::DATA_HEADER_DISTANCE_MSB", LE_DATA_HDR_MSB, 256},
        {"le_packet::INDICES", LE_INDICES, 40},
    };
    for (size_t i = 0; i < sizeof T / sizeof T[0]; i++)
        if (strcmp(name, T[i].name) == 0) {
            if (cap < T[i].n) return -1;
            memcpy(out, T[i].p, (size_t)T[i].n);
            return T[i].n;
        }
    return -1;
}

int bto_sniff_ac(const char *stream, int stream_length)
{
    build_luts();
    const int max_distance = 2;
    for (int count = 0; count < stream_length; count++) {
        const char *symbols = &stream[count];
        unsigned preamble = bto_air_to_host32(&symbols[0], 5);
        unsigned barker = bto_air_to_host32(&symbols[61], 7);
        if (PREAMBLE_DISTANCE[preamble] + BARKER_DISTANCE[barker] <= max_distance) {
            uint32_t lap = bto_air_to_host32(&symbols[38], 24);
            if (bto_check_ac(symbols, lap)) return count;
        }
    }
    return -1;
}

/* lib/packet_impl.cc:1285-1314 */
int bto_le_freq2index(double freq)
{
    build_luts();
    int chan = -1;
    if ((freq >= 2402000000.0) && (freq <= 2480000000.0))
        if (fmod(freq, 2000000.0) < 5000.0) chan = (int)((freq - 2402000000.0) / 2000000.0);
    if (chan < 0 || chan > 39) return -1;
    if (chan == 0) return 37;
    if (chan == 12) return 38;
    if (chan == 39) return 39;
    return chan < 12 ? chan - 1 : chan - 2;
}

/* lib/packet_impl.cc:1452-1527 (diagnostic printf omitted) */
/* le_packet_impl::le_packet_impl + print (lib/packet_impl.cc:1529-1664): de-whitened link symbols,
 * header fields, PDU bytes; `stream` = symbols from the preamble on, `avail` of them valid (the
 * reference copies LE_MAX_SYMBOLS = 376 whatever the length is; symbols past `avail` count as 0 here,
 * and PDU bytes past d_pdu[38] -- which the reference reads out of bounds for Length > 39 -- as 0). */
int bto_le_print(const char *stream, int avail, double freq, char *out, size_t cap)
{
    build_luts();
    int index = bto_le_freq2index(freq);
    if (index < 0 || !out || cap == 0) return -1;
    uint8_t link[376];
    for (int i = 0; i < 376; i++) link[i] = (uint8_t)(i < avail ? (stream[i] & 1) : 0);
    for (unsigned i = 40, wi = LE_INDICES[index]; i < 376; i++, wi = (wi + 1) % 127) link[i] ^= WHITENING[wi];
    unsigned aa = 0, header = 0;
    for (int i = 0; i < 32; i++) aa |= (unsigned)link[8 + i] << i;
    for (int i = 0; i < 16; i++) header |= (unsigned)link[40 + i] << i;
    uint8_t pdu[64];
    memset(pdu, 0, sizeof pdu);
    for (unsigned pi = 0, i = 56; i + 8 < 376; pi++, i += 8) {
        unsigned v = 0;
        for (int b = 0; b < 8; b++) v |= (unsigned)link[i + b] << b;
        pdu[pi] = (uint8_t)v;
    }
    size_t n = 0;
#define P(...) do { if (n < cap) n += (size_t)snprintf(out + n, cap - n, __VA_ARGS__); } while (0)
    out[0] = 0;
    if (index >= 37) {
        unsigned type = header & 0xf, txadd = (header >> 6) & 1, rxadd = (header >> 7) & 1, len = (header >> 8) & 0x3f;
        P("BTLE index=%02d, AA=%08x, PDUType=%d, TxAdd=%d, RxAdd=%d, Length=%d\n", index, aa, type, txadd, rxadd, len);
        switch (type) {
        case 0: case 2: case 4: case 6:
            P("  AdvA=%02x%02x%02x%02x%02x%02x\n", pdu[0], pdu[1], pdu[2], pdu[3], pdu[4], pdu[5]);
            P(type == 4 ? "\n  (char) ScanRspData=" : "\n  (char) AdvData=");
            for (unsigned i = 6; i < len; i++) { char c = (char)pdu[i]; if (c < ' ' || c > '~') c = '.'; P(" %c", c); }
            P(type == 4 ? "\n  (byte) ScanRspData=" : "\n  (byte) AdvData=");
            for (unsigned i = 6; i < len; i++) P("%02x", pdu[i]);
            P("\n");
            break;
        case 1:
            P("  AdvA=%02x%02x%02x%02x%02x%02x\n  InitA=%02x%02x%02x%02x%02x%02x\n", pdu[0], pdu[1], pdu[2], pdu[3], pdu[4],
              pdu[5], pdu[6], pdu[7], pdu[8], pdu[9], pdu[10], pdu[11]);
            break;
        case 3:
            P("  ScanA=%02x%02x%02x%02x%02x%02x\n  AdvA=%02x%02x%02x%02x%02x%02x\n", pdu[0], pdu[1], pdu[2], pdu[3], pdu[4],
              pdu[5], pdu[6], pdu[7], pdu[8], pdu[9], pdu[10], pdu[11]);
            break;
        case 5: {
            P("  InitA=%02x%02x%02x%02x%02x%02x\n  AdvA=%02x%02x%02x%02x%02x%02x\n", pdu[0], pdu[1], pdu[2], pdu[3], pdu[4],
              pdu[5], pdu[6], pdu[7], pdu[8], pdu[9], pdu[10], pdu[11]);
            uint32_t caa = pdu[12] | ((uint32_t)pdu[13] << 8) | ((uint32_t)pdu[14] << 16) | ((uint32_t)pdu[15] << 24);
            uint32_t crcinit = pdu[16] | ((uint32_t)pdu[17] << 8) | ((uint32_t)pdu[18] << 16);
            unsigned winsize = pdu[19], winoffset = pdu[20] | (pdu[21] << 8), interval = pdu[22] | (pdu[23] << 8);
            unsigned latency = pdu[24] | (pdu[25] << 8), timeout = pdu[26] | (pdu[27] << 8);
            uint64_t chm = pdu[28] | ((uint64_t)pdu[29] << 8) | ((uint64_t)pdu[30] << 16) | ((uint64_t)pdu[31] << 24) |
                           ((uint64_t)pdu[32] << 32);
            P("  AA=%08x, CRCInit=%06x, WinSize=%d, WinOffset=%d\n", caa, crcinit, winsize, winoffset);
            P("  Interval=%d, Latency=%d, Timeout=%d, ChM=%010lx, Hop=%d, SCA=%d\n", interval, latency, timeout,
              (unsigned long)chm, pdu[33] & 0x1f, (pdu[33] >> 5) & 7);
            break;
        }
        default: break;
        }
    } else {
        P("BTLE index=%02d, AA=%08x, LLID=%d, NESN=%d, SN=%d, MD=%d, Length=%d\n", index, aa, header & 3, (header >> 2) & 1,
          (header >> 3) & 1, (header >> 4) & 1, (header >> 8) & 0x1f);
    }
#undef P
    return (int)n;
}

int bto_sniff_aa(const char *stream, int stream_length, double freq)
{
    build_luts();
    int index = bto_le_freq2index(freq);
    const uint8_t *phlsb, *phmsb;
    if (index >= 37) { phlsb = LE_ACC_HDR_LSB; phmsb = LE_ACC_HDR_MSB; }
    else if (index < 0) return -1;
    else { phlsb = LE_DATA_HDR_LSB; phmsb = LE_DATA_HDR_MSB; }
    for (int count = 0; count < stream_length; count++) {
        const char *symbols = &stream[count];
        unsigned preamble = bto_air_to_host32(&symbols[0], 9);
        char hbuf[16];
        unsigned hi, wi;
        for (hi = 0, wi = LE_INDICES[index]; hi < 16; hi++, wi = (wi + 1) % 127)
            hbuf[hi] = (char)((symbols[hi + 40] & 1) ^ WHITENING[wi]);
        unsigned header_lsb = bto_air_to_host32(&hbuf[0], 8);
        unsigned header_msb = bto_air_to_host32(&hbuf[8], 8);
        int distance = LE_PREAMBLE_DISTANCE[preamble] + phlsb[header_lsb] + phmsb[header_msb];
        int max_distance = 0;
        if (index >= 37) {
            int aa_distance = 0;
            for (int b = 0; b < 4; b++)
                aa_distance += LE_AA_DISTANCE[b][bto_air_to_host32(&symbols[8 + 8 * b], 8)];
            distance += aa_distance;
            max_distance += 2;
        }
        if (distance <= max_distance) return count;
    }
    return -1;
}

/* classic_packet_impl::header_present (lib/packet_impl.cc:1205-1242): `symbols` starts at the
 * access code, `length` = symbols the packet object holds (min(len, 3125), packet_impl.cc:53) */
int bto_header_present(const char *symbols, int length)
{
    if (length > 3125) length = 3125;
    if (length < 126) return 0;
    const char *stream = symbols + 67;
    int be = 0;
    char msb = stream[0] & 1;
    be += (stream[1] & 1) ^ !msb;
    be += (stream[2] & 1) ^ msb;
    be += (stream[3] & 1) ^ !msb;
    be += (stream[4] & 1) ^ msb;
    stream += 5;
    for (int a = 0; a < 54; a += 3) {
        int b = a + 1, cc = a + 2;
        be += ((stream[a] ^ stream[b]) | (stream[b] ^ stream[cc]) | (stream[cc] ^ stream[a])) & 1;
    }
    return be < 5;                                   /* ID_THRESHOLD, packet.h:185 */
}

/* ------------------------------------------------------------------------- */
/* work loops: lib/multi_sniffer_impl.cc:82-166, lib/multi_LAP_impl.cc:65-114  */
/* ------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------
 * [EXT libbtbb -- NOT in /root/reference, version unpinned by the reference (cmake/Modules/
 * FindBTBB.cmake only looks for btbb.h / libbtbb)].  Restatement of the published algorithm of
 * btbb_find_ac() + btbb_init() (libbtbb, lib/src/bluetooth_packet.c), as called by
 * lib/multi_LAP_impl.cc:55 (btbb_init(1)) and :93 (btbb_find_ac(symbols, latest_ac, LAP_ANY, 1, &pkt)):
 *   for every offset `count` < search_length, sync word = 64 symbols from `count` (LSB first):
 *     - gate: BARKER_DISTANCE[sync bits 57..63] <= MAX_BARKER_ERRORS (1); the 7-bit field is then
 *       replaced by the nearest valid pattern (barker_correct[]), not counted as an error;
 *     - codeword = syncword ^ PN; zero BCH(64,30) syndrome -> 0 errors; otherwise the syndrome is
 *       looked up in the map btbb_init(n) builds from all error patterns of <= n bits over sync
 *       bits 0..57 (the Barker bits are excluded there); a match is XORed into the sync word and
 *       ac_errors = number of corrected bits; no match -> not an access code;
 *     - LAP = (syncword >> 34) & 0xffffff; the first offset that passes is returned (this offset is
 *       the sync-word start, 4 symbols after the preamble start bto_sniff_ac returns).
 * PARITY UNPINNED: libbtbb is absent; anchored on the reference's call site only.
 * --------------------------------------------------------------------------------------- */
static uint64_t bch_parity30(uint64_t x30)
{
    /* parity(D) = D^34 x(D) mod g(D), g = 0260534236651 (octal), bit i of the result = D^i */
    const uint64_t GEN = 0260534236651ULL;
    uint64_t rem = (x30 & ((1ULL << 30) - 1)) << 34;
    for (int bit = 63; bit >= 34; bit--)
        if ((rem >> bit) & 1) rem ^= GEN << (bit - 34);
    return rem & ((1ULL << 34) - 1);
}
static uint64_t bch_syndrome(uint64_t codeword)
{
    return (codeword & ((1ULL << 34) - 1)) ^ bch_parity30(codeword >> 34);
}

int bto_btbb_find_ac(const char *stream, int search_length, int max_ac_errors, uint32_t *lap_out,
                     int *ac_errors_out)
{
    build_luts();
    const uint64_t PN = 0x83848D96BBCC54FCULL;
    if (max_ac_errors < 0 || max_ac_errors > 1) return -2;      /* the reference uses 1 */
    static uint64_t synd1[58];
    static int ready = 0;
    if (!ready) { for (int i = 0; i < 58; i++) synd1[i] = bch_syndrome(1ULL << i); ready = 1; }
    for (int count = 0; count < search_length; count++) {
        const char *sym = &stream[count];
        unsigned barker = 0;
        for (int j = 0; j < 7; j++) barker |= ((unsigned)(sym[57 + j] & 1)) << j;
        if (BARKER_DISTANCE[barker] > 1) continue;
        uint64_t sw = 0;
        for (int i = 0; i < 64; i++) sw |= ((uint64_t)(sym[i] & 1)) << i;
        const unsigned fixed = (popc(barker ^ 0x27) <= popc(barker ^ 0x58)) ? 0x27u : 0x58u;
        sw = (sw & 0x01ffffffffffffffULL) | ((uint64_t)fixed << 57);
        uint64_t synd = bch_syndrome(sw ^ PN);
        int errs = 0;
        if (synd) {
            int k = -1;
            if (max_ac_errors >= 1)
                for (int i = 0; i < 58; i++) if (synd1[i] == synd) { k = i; break; }
            if (k < 0) continue;
            sw ^= 1ULL << k;
            errs = 1;
        }
        if (lap_out) *lap_out = (uint32_t)(sw >> 34) & 0xffffff;
        if (ac_errors_out) *ac_errors_out = errs;
        return count;
    }
    return -1;
}

static int search_symbols(const bto_ctx *c, char *symbols, int len, int channel, uint32_t slot,
                          double snr, bto_hit *hits, int max_hits)
{
    int nh = 0;
    double freq = channel_abs_freq(channel);
    if (c->mode == BTO_MODE_LAP) {
        /* multi_LAP_impl.cc:89-101; the search is libbtbb's (default, as in the reference) or the
         * in-tree classic_packet::sniff_ac (BTO_CORRELATOR_INTREE) */
        if (len >= SYMBOLS_PER_SHORTENED_AC) {
            int latest = ((len - SYMBOLS_PER_SHORTENED_AC) < SYMBOLS_PER_SLOT)
                             ? (len - SYMBOLS_PER_SHORTENED_AC) : SYMBOLS_PER_SLOT;
            uint32_t lap = 0; int errs = 0, off;
            if (c->correlator == BTO_CORRELATOR_BTBB) off = bto_btbb_find_ac(symbols, latest, 1, &lap, &errs);
            else {
                off = bto_sniff_ac(symbols, latest);
                if (off >= 0) { lap = bto_air_to_host32(&symbols[off + 38], 24); errs = bto_ac_errors(&symbols[off], lap); }
            }
            if (off >= 0 && nh < max_hits) {
                bto_hit *h = &hits[nh++];
                memset(h, 0, sizeof *h);
                h->slot = slot; h->channel = channel; h->offset = off;
                h->lap = lap; h->ac_errors = errs;
                h->kind = BTO_KIND_AC; h->nsym = len - off; h->snr = snr;
            }
        }
        return nh;
    }
    /* sniffer: classic multi-hit loop :107-127 */
    int symp = 0;
    int limit = ((len - SYMBOLS_PER_SHORTENED_AC) < SYMBOLS_PER_SLOT)
                    ? (len - SYMBOLS_PER_SHORTENED_AC) : SYMBOLS_PER_SLOT;
    while (limit >= 0) {
        int i = bto_sniff_ac(&symbols[symp], limit);
        if (i < 0) break;
        int step = i + SYMBOLS_PER_SHORTENED_AC;
        if (nh < max_hits) {
            bto_hit *h = &hits[nh++];
            memset(h, 0, sizeof *h);
            h->slot = slot; h->channel = channel; h->offset = symp + i;
            h->lap = bto_air_to_host32(&symbols[symp + i + 38], 24);
            h->ac_errors = bto_ac_errors(&symbols[symp + i], h->lap);
            h->kind = BTO_KIND_AC; h->nsym = len - i; h->snr = snr;
        }
        len -= step; symp += step; limit -= step;
    }
    if (c->le_enable) {
        /* LE pass :129-149; note `len` keeps the value the classic pass left (Q6) */
        symp = 0;
        limit = ((len - SYMBOLS_PER_SHORTENED_AC) < SYMBOLS_PER_SLOT)
                    ? (len - SYMBOLS_PER_SHORTENED_AC) : SYMBOLS_PER_SLOT;
        while (limit >= 0) {
            int i = bto_sniff_aa(&symbols[symp], limit, freq);
            if (i < 0) break;
            int step = i + SYMBOLS_PER_LE_PREAMBLE_AA;
            if (nh < max_hits) {
                bto_hit *h = &hits[nh++];
                memset(h, 0, sizeof *h);
                h->slot = slot; h->channel = channel; h->offset = symp + i;
                h->lap = bto_air_to_host32(&symbols[symp + i + 8], 32);
                h->kind = BTO_KIND_AA; h->nsym = len - i; h->snr = snr;
            }
            len -= step; symp += step; limit -= step;
        }
    }
    return nh;
}

static int work_channel(bto_ctx *c, int ch, const float *xt, int ncol, uint32_t slot,
                        bto_hit *hits, int max_hits, float *chbuf, char *symbols)
{
    double e_on, snr;
    int n = channel_samples_d(c, ch, xt, ncol, chbuf, &e_on);
    if (!check_snr_d(c, ch, e_on, xt, ncol, &snr, NULL)) return 0;
    int len = bto_channel_symbols(c, chbuf, n, symbols, NULL);
    return search_symbols(c, symbols, len, ch, slot, snr, hits, max_hits);
}

int bto_work(bto_ctx *c, const float *win, uint32_t slot, bto_hit *hits, int max_hits)
{
    float *xt;
    int ncol = transpose_cols(win, c->history, c->decim, DDC_V + 2, &xt);
    int nout = bto_ddc_out(c);
    float *chbuf = (float *)malloc(sizeof(float) * 2 * (size_t)(nout + 1));
    char *symbols = (char *)calloc((size_t)c->history + 64, 1);
    int nh = 0;
    for (int ch = c->low_ch; ch <= c->high_ch; ch++)
        nh += work_channel(c, ch, xt, ncol, slot, hits + nh, max_hits - nh, chbuf, symbols);
    free(chbuf); free(symbols); free(xt);
    return nh;
}

int bto_run_stream(bto_ctx *c, const float *iq, size_t n_complex, bto_hit *hits, int max_hits,
                   int *slots_done)
{
    int H = c->history, slot = c->slot;
    size_t S = n_complex / (size_t)slot;
    float *win = (float *)malloc(sizeof(float) * 2 * (size_t)H);
    int nh = 0;
    for (size_t k = 0; k < S; k++) {
        /* window k = absolute samples [k*slot-(H-1), k*slot]; negative indices are the
         * zeros GNU Radio pre-fills for history()-1 items [EXT] */
        long long a0 = (long long)k * slot - (H - 1);
        for (int i = 0; i < H; i++) {
            long long a = a0 + i;
            if (a < 0 || (size_t)a >= n_complex) { win[2 * i] = 0.f; win[2 * i + 1] = 0.f; }
            else { win[2 * i] = iq[2 * a]; win[2 * i + 1] = iq[2 * a + 1]; }
        }
        nh += bto_work(c, win, (uint32_t)k, hits + nh, max_hits - nh);
    }
    free(win);
    if (slots_done) *slots_done = (int)S;
    return nh;
}

static int hit_cmp(const void *a, const void *b)
{
    const bto_hit *x = (const bto_hit *)a, *y = (const bto_hit *)b;
    if (x->slot != y->slot) return x->slot < y->slot ? -1 : 1;
    if (x->channel != y->channel) return x->channel < y->channel ? -1 : 1;
    if (x->kind != y->kind) return x->kind < y->kind ? -1 : 1;
    return x->offset < y->offset ? -1 : (x->offset > y->offset);
}

int bto_run_stream_mt(bto_ctx *c, const float *iq, size_t n_complex, bto_hit *hits, int max_hits,
                      int *slots_done, int threads)
{
    int H = c->history, slot = c->slot;
    long long S = (long long)(n_complex / (size_t)slot);
    int nh = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
    omp_set_num_threads(threads);
#endif
    int nout = bto_ddc_out(c);
#pragma omp parallel
    {
        bto_ctx local = *c;                        /* private M&M state (windowed reset) */
        local.mm_policy = BTO_MM_WINDOWED_RESET;
        const int D = local.decim, ncol = (H + D - 1) / D + DDC_V + 2;
        float *xt = (float *)calloc((size_t)2 * D * ncol, sizeof(float));
        float *chbuf = (float *)malloc(sizeof(float) * 2 * (size_t)(nout + 1));
        char *symbols = (char *)calloc((size_t)H + 64, 1);
        bto_hit tmp[64];
#pragma omp for schedule(dynamic, 1)
        for (long long k = 0; k < S; k++) {
            long long a0 = k * slot - (H - 1);
            for (int i = 0; i < H; i++) {
                long long a = a0 + i;
                const int m = i / D, r = i - m * D;
                const int in = !(a < 0 || (size_t)a >= n_complex);
                xt[(size_t)(2 * r) * ncol + m] = in ? iq[2 * a] : 0.f;
                xt[(size_t)(2 * r + 1) * ncol + m] = in ? iq[2 * a + 1] : 0.f;
            }
            for (int ch = local.low_ch; ch <= local.high_ch; ch++) {
                int n = work_channel(&local, ch, xt, ncol, (uint32_t)k, tmp, 64, chbuf, symbols);
                if (n > 0) {
#pragma omp critical
                    {
                        for (int i = 0; i < n && nh < max_hits; i++) hits[nh++] = tmp[i];
                    }
                }
            }
        }
        free(xt); free(chbuf); free(symbols);
    }
    qsort(hits, nh, sizeof(bto_hit), hit_cmp);
    if (slots_done) *slots_done = (int)S;
    return nh;
}

int bto_scan_symbols(const char *symbols, size_t n, bto_hit *hits, int max_hits)
{
    int nh = 0;
    size_t pos = 0;
    while (pos + SYMBOLS_PER_SHORTENED_AC <= n) {
        size_t remaining = n - pos - SYMBOLS_PER_SHORTENED_AC + 1;
        int chunk = remaining > 1000000 ? 1000000 : (int)remaining;
        int i = bto_sniff_ac(symbols + pos, chunk);
        if (i < 0) { pos += chunk; continue; }
        if (nh < max_hits) {
            bto_hit *h = &hits[nh++];
            memset(h, 0, sizeof *h);
            h->offset = (int32_t)(pos + i);
            h->lap = bto_air_to_host32(&symbols[pos + i + 38], 24);
            h->ac_errors = bto_ac_errors(&symbols[pos + i], h->lap);
            h->kind = BTO_KIND_AC;
        }
        pos += (size_t)i + SYMBOLS_PER_SHORTENED_AC;
    }
    return nh;
}
