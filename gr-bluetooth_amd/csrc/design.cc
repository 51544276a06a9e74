// design.cc -- see design.h.  Reference: lib/multi_block.cc:40-120 (constructor),
// :299-303 (set_symbol_history), :306-342 (set_channels); GNU Radio 3.7 firdes /
// freq_xlating_fir_filter / mmse_fir_interpolator / fast_atan2f semantics [EXT] as
// documented in DESIGN.md.
#include "design.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace btgpu {

namespace {

constexpr double kSymbolRate = 1e6;
constexpr double kBaseFrequency = 2402e6;
constexpr double kChannelWidth = 1e6;
constexpr double kPi = 3.14159265358979323846;

struct Turns {                 // an angle as an exact fraction of a turn when possible
    bool rational;
    long long num, den;        // rational: num/den in [0,1)
    double frac;               // otherwise
};

Turns turns_of(double f_hz, double fs_hz, long long k)
{
    Turns t{};
    if (f_hz == std::floor(f_hz) && fs_hz == std::floor(fs_hz) && std::fabs(f_hz) < 4e15 &&
        fs_hz > 0 && fs_hz < 4e15) {
        __int128 den = (long long)fs_hz;
        __int128 r = ((__int128)k * (long long)f_hz) % den;
        if (r < 0) r += den;
        t.rational = true;
        t.num = (long long)r;
        t.den = (long long)den;
    } else {
        t.rational = false;
        double x = std::fmod((double)k * f_hz / fs_hz, 1.0);
        if (x < 0) x += 1.0;
        t.frac = x;
    }
    return t;
}

// e^{+j 2 pi turns}; exact on the axes
void unit_phasor(const Turns &t, float &re, float &im)
{
    if (t.rational && ((__int128)4 * t.num) % t.den == 0) {
        switch ((int)(((__int128)4 * t.num) / t.den)) {
            case 0: re = 1.f; im = 0.f; return;
            case 1: re = 0.f; im = 1.f; return;
            case 2: re = -1.f; im = 0.f; return;
            default: re = 0.f; im = -1.f; return;
        }
    }
    double x = t.rational ? (double)t.num / (double)t.den : t.frac;
    double a = 2.0 * kPi * x;
    re = (float)std::cos(a);
    im = (float)std::sin(a);
}

long long gcdll(long long a, long long b)
{
    a = a < 0 ? -a : a;
    b = b < 0 ? -b : b;
    while (b) { long long t = a % b; a = b; b = t; }
    return a;
}

void build_bank_impl(FilterBank &b, const std::vector<float> &h, int low_ch, int nch, double extra_hz,
                     double center_freq, double fs, int decim)
{
    b.ntaps = (int)h.size();
    b.blk = decim > 0 ? decim : 1;                               // summation order: blocks of `decim` taps (kernels.hip.h ddc_direct_kernel)
    b.ntp = (b.ntaps + b.blk - 1) / b.blk * b.blk;
    b.nch = nch;
    b.taps.assign((size_t)nch * b.ntp * 2, 0.f);
    b.foff.resize(nch);
    long long period = 1;
    bool periodic = true;
    for (int c = 0; c < nch; c++) {
        double foff = kBaseFrequency + (low_ch + c) * kChannelWidth + extra_hz - center_freq;
        b.foff[c] = foff;
        for (int k = 0; k < b.ntaps; k++) {
            float wr, wi;
            unit_phasor(turns_of(foff, fs, k), wr, wi);
            size_t j = (size_t)(b.ntaps - 1 - k);
            b.taps[((size_t)c * b.ntp + j) * 2 + 0] = h[k] * wr;
            b.taps[((size_t)c * b.ntp + j) * 2 + 1] = h[k] * wi;
        }
        Turns step = turns_of(-foff, fs, decim);        // derotation advance per output
        if (!step.rational) periodic = false;
        else {
            long long q = step.den / gcdll(step.num, step.den);
            period = period / gcdll(period, q) * q;
            if (period > 4096) periodic = false;
        }
    }
    b.rot_period = periodic ? (int)period : 0;
    b.rot.clear();
    if (periodic) {
        b.rot.resize((size_t)nch * period * 2);
        for (int c = 0; c < nch; c++)
            for (long long i = 0; i < period; i++) {
                float rr, ri;
                unit_phasor(turns_of(-b.foff[c], fs, (long long)decim * i), rr, ri);
                b.rot[((size_t)c * period + i) * 2 + 0] = rr;
                b.rot[((size_t)c * period + i) * 2 + 1] = ri;
            }
    }
}

double band_sinc(double B, double t)
{
    if (std::fabs(t) < 1e-12) return 2.0 * B;
    return std::sin(2.0 * kPi * B * t) / (kPi * t);
}

// 8-tap MMSE fractional-delay interpolator bank, |f| <= 0.25 cycles/sample, 129 steps,
// coefficients kept to 6 significant digits like GNU Radio's interpolator_taps.h [EXT].
void build_mmse(float *tab)
{
    const double B = 0.25;
    for (int s = 0; s <= kMmseSteps; s++) {
        double mu = (double)s / kMmseSteps;
        double A[8][9];
        for (int k = 0; k < 8; k++) {
            for (int j = 0; j < 8; j++) A[k][j] = band_sinc(B, (double)(k - j));
            A[k][8] = band_sinc(B, 3.0 + mu - k);
        }
        for (int c = 0; c < 8; c++) {                   // Gauss-Jordan, partial pivoting
            int p = c;
            for (int r = c + 1; r < 8; r++)
                if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
            if (p != c)
                for (int k = 0; k < 9; k++) std::swap(A[c][k], A[p][k]);
            for (int r = 0; r < 8; r++)
                if (r != c) {
                    double f = A[r][c] / A[c][c];
                    for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
                }
        }
        for (int k = 0; k < 8; k++) {
            double w = A[k][8] / A[k][k];
            char buf[64];
            std::snprintf(buf, sizeof buf, "%.5e", w);
            double v = std::strtod(buf, nullptr);
            if (std::fabs(v) < 5e-7) v = 0.0;
            tab[s * kMmseTaps + (7 - k)] = (float)v;     // sample k uses table entry 7-k
        }
    }
}

}  // namespace

void build_direct_bank(FilterBank &b, const std::vector<float> &h, int low_ch, int nch, double extra_hz,
                       double center_freq, double fs, int decim)
{
    build_bank_impl(b, h, low_ch, nch, extra_hz, center_freq, fs, decim);
}

int firdes_ntaps(double fs, double tw)
{
    int n = (int)(44.0 * fs / (22.0 * tw));             // Hann: 44 dB [EXT compute_ntaps]
    if ((n & 1) == 0) n++;
    return n;
}

std::vector<float> firdes_low_pass_hann(double gain, double fs, double fc, double tw)
{
    int ntaps = firdes_ntaps(fs, tw);
    std::vector<float> taps(ntaps);
    int M = (ntaps - 1) / 2;
    double fwT0 = 2.0 * kPi * fc / fs;
    float Mf = (float)(ntaps - 1);
    for (int n = -M; n <= M; n++) {
        float w = (float)(0.5 - 0.5 * std::cos((2.0 * kPi * (n + M)) / Mf));
        if (n == 0) taps[n + M] = (float)(fwT0 / kPi * w);
        else taps[n + M] = (float)(std::sin(n * fwT0) / (n * kPi) * w);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    return taps;
}

uint64_t sync_word(uint32_t lap)
{
    const uint64_t PN = 0x83848D96BBCC54FCULL;
    const uint64_t GEN = 0260534236651ULL;               // BCH(64,30) generator, degree 34
    lap &= 0xffffffu;
    uint64_t info = lap;
    const unsigned barker = ((lap >> 23) & 1) ? 0x32u /*110010*/ : 0x0du /*001101*/;
    for (int i = 0; i < 6; i++) info |= (uint64_t)((barker >> (5 - i)) & 1) << (24 + i);
    uint64_t x = (info ^ (PN >> 34)) & ((1ULL << 30) - 1);
    uint64_t rem = x << 34;
    for (int bit = 63; bit >= 34; bit--)
        if ((rem >> bit) & 1) rem ^= GEN << (bit - 34);
    uint64_t cw = (x << 34) | (rem & ((1ULL << 34) - 1));
    return cw ^ PN;                                      // bit i = i-th transmitted sync bit
}

void access_code_68(uint32_t lap, uint64_t &lo, uint32_t &hi)
{
    uint64_t sw = sync_word(lap);
    uint64_t pre = (sw & 1) ? 0x5ULL /*1,0,1,0 air order*/ : 0xAULL /*0,1,0,1*/;
    lo = pre | (sw << 4);
    hi = (uint32_t)(sw >> 60) & 0xf;
}

void access_code_bytes(uint32_t lap, uint8_t ac[9])
{
    uint64_t sw = sync_word(lap);
    uint8_t bits[72];
    uint64_t lo; uint32_t hi;
    access_code_68(lap, lo, hi);
    for (int i = 0; i < 64; i++) bits[i] = (lo >> i) & 1;
    for (int i = 0; i < 4; i++) bits[64 + i] = (hi >> i) & 1;
    const bool last = (sw >> 63) & 1;
    const uint8_t tr1[4] = {0, 1, 0, 1}, tr0[4] = {1, 0, 1, 0};
    for (int i = 0; i < 4; i++) bits[68 + i] = last ? tr1[i] : tr0[i];
    for (int b = 0; b < 9; b++) {
        uint8_t v = 0;
        for (int i = 0; i < 8; i++) v = (uint8_t)((v << 1) | bits[8 * b + i]);
        ac[b] = v;
    }
}

int make_design(const btgpu_config &cfg, Design &o)
{
    if (!(cfg.sample_rate >= 2e6) || !(cfg.sample_rate < 4e9) || !std::isfinite(cfg.center_freq))
        return BTGPU_EINVAL;                             // apps/btrx:66-78 requires >= 2 samples/symbol
    if (cfg.mode != BTGPU_MODE_LAP && cfg.mode != BTGPU_MODE_SNIFFER) return BTGPU_EINVAL;
    int corr = cfg.correlator;
    if (corr == BTGPU_CORRELATOR_AUTO) corr = cfg.mode == BTGPU_MODE_LAP ? BTGPU_CORRELATOR_BTBB : BTGPU_CORRELATOR_INTREE;
    if (corr != BTGPU_CORRELATOR_INTREE && corr != BTGPU_CORRELATOR_BTBB) return BTGPU_EINVAL;
    if (corr == BTGPU_CORRELATOR_BTBB && cfg.mode != BTGPU_MODE_LAP) return BTGPU_EUNSUPPORTED;   // multi_sniffer uses sniff_ac
    o.cfg = cfg;
    btgpu_design &d = o.d;
    std::memset(&d, 0, sizeof d);
    d.correlator = corr;
    const double fs = cfg.sample_rate;
    d.samples_per_symbol = fs / kSymbolRate;
    o.samples_per_slot_d = (int)kSymbolsPerSlot * d.samples_per_symbol;
    d.samples_per_slot = (int)o.samples_per_slot_d;
    int history = (int)(1 * o.samples_per_slot_d);

    o.h_channel = firdes_low_pass_hann(1.0, fs, 500000.0, 300000.0);
    o.h_noise = firdes_low_pass_hann(1.0, fs, 22500.0, 10000.0);
    d.ntaps_channel = (int)o.h_channel.size();
    d.ntaps_noise = (int)o.h_noise.size();

    d.decimation = (int)d.samples_per_symbol / 2;
    if (d.decimation < 1) d.decimation = 1;
    const double channel_sps = d.samples_per_symbol / d.decimation;

    // set_channels
    const double center = (cfg.center_freq - kBaseFrequency) / kChannelWidth;
    const double bw = fs / kChannelWidth;
    int lo = (int)((center - bw / 2) + (0.9 / 2) + 1);
    if (lo < 0) lo = 0;
    int hi = (int)((center + bw / 2) - (0.9 / 2));
    if (hi > 78) hi = 78;
    d.low_channel = lo;
    d.high_channel = hi;
    const int nch = hi >= lo ? hi - lo + 1 : 0;
    if (nch <= 0) return BTGPU_EINVAL;

    build_bank_impl(o.channel, o.h_channel, lo, nch, 0.0, cfg.center_freq, fs, d.decimation);
    build_bank_impl(o.noise, o.h_noise, lo, nch, 790000.0, cfg.center_freq, fs, d.decimation);

    o.demod_gain = (float)(channel_sps / (kPi / 2));
    o.gain_mu = 0.175f;
    o.mu0 = 0.32f;
    o.omega_relative_limit = 0.005f;
    o.omega0 = (float)channel_sps;
    o.gain_omega = (float)(.25 * o.gain_mu * o.gain_mu);
    o.omega_mid = o.omega0;

    const int channel_history = d.ntaps_channel + d.decimation * kMmseTaps;
    const int noise_history = d.ntaps_noise;
    if (channel_history > noise_history) {
        history += channel_history;
        d.first_channel_sample = 0;
        d.first_noise_sample = channel_history - noise_history;
    } else {
        history += noise_history;
        d.first_noise_sample = 0;
        d.first_channel_sample = noise_history - channel_history;
    }
    const int nsym = cfg.mode == BTGPU_MODE_SNIFFER ? kSymbolsHistorySniffer : kSymbolsShortAC;
    d.history = (int)(history + (nsym * d.samples_per_symbol));

    // lib/multi_block.cc:194-200 and :269 hand (ninput - (history() - 1) - first) resp. one slot to
    // the DDC's fixed_rate_ninput_to_noutput(), and gr::sync_decimator [EXT] answers
    // max(0, n - history() + 1) / decimation with history() = ntaps: the filter length comes off a
    // second time, so a window yields 7494/7495 channel outputs (not 7508) and 850 noise outputs
    // (not 1250) at every integer rate.
    {
        const int ddc_samples = d.history - (d.ntaps_channel - 1) - d.first_channel_sample;
        d.ddc_out = std::max(0, ddc_samples - d.ntaps_channel + 1) / d.decimation;
        d.noise_out = std::max(0, d.samples_per_slot - d.ntaps_noise + 1) / d.decimation;
    }
    if (d.ddc_out < 2 * kMmseTaps || d.noise_out < 1) return BTGPU_EINVAL;

    // the shared output grid needs whole outputs per slot
    o.segmented = false;                                         // (a Design may be reused for another configuration)
    if (d.samples_per_slot % d.decimation != 0) {
        // consecutive windows sit on different decimation phases (odd samples per symbol >= 5): no
        // shared output grid; every window is filtered on its own, like the reference does
        o.segmented = true;
        o.outs_per_slot = d.ddc_out;
        o.blocks_per_window = 1;
        o.tail = 0;
    } else {
        o.outs_per_slot = d.samples_per_slot / d.decimation;
        o.blocks_per_window = d.ddc_out / o.outs_per_slot;
        o.tail = d.ddc_out % o.outs_per_slot;
    }

    for (int i = 0; i <= 255; i++) o.atan_tab[i] = (float)std::atan((double)i / 255.0);
    o.atan_tab[256] = o.atan_tab[255];
    build_mmse(o.mmse);

    // affine access-code tables: AC(lap) = AC(0) ^ XOR_b col[b]
    access_code_68(0, o.ac.a0_lo, o.ac.a0_hi);
    uint64_t col_lo[24]; uint32_t col_hi[24];
    for (int b = 0; b < 24; b++) {
        uint64_t l; uint32_t h;
        access_code_68(1u << b, l, h);
        col_lo[b] = l ^ o.ac.a0_lo;
        col_hi[b] = h ^ o.ac.a0_hi;
    }
    for (int byte = 0; byte < 3; byte++)
        for (int v = 0; v < 256; v++) {
            uint64_t l = 0; uint32_t h = 0;
            for (int b = 0; b < 8; b++)
                if ((v >> b) & 1) { l ^= col_lo[8 * byte + b]; h ^= col_hi[8 * byte + b]; }
            o.ac.byte_lo[byte][v] = l;
            o.ac.byte_hi[byte][v] = h;
        }

    // parity column of each LAP bit taken alone: D^34 (D^k) mod g(D)
    for (int k = 0; k < 24; k++) {
        const uint64_t GEN = 0260534236651ULL;
        uint64_t rem = 1ULL << (34 + k);
        for (int bit = 63; bit >= 34; bit--)
            if ((rem >> bit) & 1) rem ^= GEN << (bit - 34);
        o.ac.btbb_pcol[k] = rem & ((1ULL << 34) - 1);
    }

    // ---- LE tables, regenerated from their rules ----
    {
        auto mind = [](unsigned v, const std::vector<unsigned> &set) {
            int best = 99;
            for (unsigned s : set) { int dd = __builtin_popcount(v ^ s); if (dd < best) best = dd; }
            return (uint8_t)best;
        };
        std::vector<unsigned> acc_lsb, acc_msb, dat_lsb, dat_msb;
        for (unsigned t = 0; t <= 6; t++) { acc_lsb.push_back(t); acc_lsb.push_back(0xc0 | t); }
        for (unsigned t = 0x06; t <= 0x24; t++) acc_msb.push_back(t);
        for (unsigned t = 0; t < 0x20; t++) { if (t & 3) dat_lsb.push_back(t); dat_msb.push_back(t); }
        for (unsigned v = 0; v < 256; v++) {
            o.le.hdr[0][v] = mind(v, acc_lsb); o.le.hdr[1][v] = mind(v, acc_msb);
            o.le.hdr[2][v] = mind(v, dat_lsb); o.le.hdr[3][v] = mind(v, dat_msb);
        }
        uint8_t wseq[127] = {1, 1, 1, 0, 0, 0, 1};           // x^7 + x^4 + 1 whitening sequence
        for (int i = 7; i < 127; i++) wseq[i] = wseq[i - 7] ^ wseq[i - 3];
        // classic whitening: the same register run from position 6 = 1, positions 0..5 = CLK1..CLK6
        // (the output is the bit leaving position 6); the header uses the first 18 bits
        for (int clk = 0; clk < 64; clk++) {
            uint8_t p[7];
            for (int i = 0; i < 6; i++) p[i] = (clk >> i) & 1;
            p[6] = 1;
            uint32_t m = 0;
            for (int k = 0; k < 18; k++) {
                const uint8_t ob = p[6];
                m |= (uint32_t)ob << k;
                uint8_t q[7] = {ob, p[0], p[1], p[2], (uint8_t)(p[3] ^ ob), p[4], p[5]};
                std::memcpy(p, q, 7);
            }
            o.wh.first18[clk] = m;
        }
        for (int idx = 0; idx < 40; idx++) {
            // LE whitening LFSR: position 0 = 1, positions 1..6 = channel index MSB first
            uint8_t p[7], s7[7];
            p[0] = 1;
            for (int i = 0; i < 6; i++) p[1 + i] = (idx >> (5 - i)) & 1;
            for (int k = 0; k < 7; k++) {
                uint8_t ob = p[6];
                s7[k] = ob;
                uint8_t q[7] = {ob, p[0], p[1], p[2], (uint8_t)(p[3] ^ ob), p[4], p[5]};
                std::memcpy(p, q, 7);
            }
            int start = 0;
            for (int i = 0; i < 127; i++) {
                bool ok = true;
                for (int k = 0; k < 7 && ok; k++) ok = wseq[(i + k) % 127] == s7[k];
                if (ok) { start = i; break; }
            }
            uint16_t m = 0;
            for (int i = 0; i < 16; i++) m |= (uint16_t)(wseq[(start + i) % 127] << i);
            o.le.whiten16[idx] = m;
        }
        for (int ch = 0; ch < 79; ch++) {
            // le_packet::freq2index (lib/packet_impl.cc:1285-1314): even MHz only
            int idx = -1;
            if (ch % 2 == 0) {
                int chan = ch / 2;
                idx = chan == 0 ? 37 : chan == 12 ? 38 : chan == 39 ? 39 : (chan < 12 ? chan - 1 : chan - 2);
            }
            o.le.index_of_channel[ch] = (int8_t)idx;
        }
    }

    d.channelizer = cfg.channelizer == BTGPU_CHANNELIZER_AUTO ? BTGPU_CHANNELIZER_DIRECT
                                                              : cfg.channelizer;
    return BTGPU_OK;
}

}  // namespace btgpu
