// design_fast.cc -- host design of the polyphase fast path (DESIGN.md "Fast path"):
//  * channel bank: the SAME 667-tap prototype as the direct form, evaluated as a 100-branch
//    polyphase filter + 100-point DFT (exact restructuring; float rounding differs);
//  * noise bank: the reference's 20001-tap 22.5 kHz low-pass is replaced by an equivalent
//    two-stage cascade -- stage 1: polyphase bank with an order-6 B-spline prototype at hop
//    R = 5*decim (deep nulls at every alias of the noise band), stage 2: an L3-tap FIR at
//    the stage-1 rate whose coefficients are the least-squares fit that makes the COMPOSITE
//    impulse response equal the reference filter (fit error ~1e-6 of the DC gain) -- and the
//    per-slot mean |y|^2 is taken with band-limited quadrature weights on the stage-2 grid.
#include <cmath>
#include <cstring>
#include <numeric>

#include "design.h"

namespace btgpu {
namespace {

constexpr double kPi = 3.14159265358979323846;

void phasor_turns(double turns, float &re, float &im)
{
    double t = turns - std::floor(turns);
    double q4 = t * 4.0;
    double qr = std::round(q4);
    if (std::fabs(q4 - qr) < 1e-12) {
        switch (((int)qr) & 3) {
            case 0: re = 1.f; im = 0.f; return;
            case 1: re = 0.f; im = 1.f; return;
            case 2: re = -1.f; im = 0.f; return;
            default: re = 0.f; im = -1.f; return;
        }
    }
    re = (float)std::cos(2.0 * kPi * t);
    im = (float)std::sin(2.0 * kPi * t);
}

long long gcdll(long long a, long long b)
{
    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
    while (b) { long long t = a % b; a = b; b = t; }
    return a;
}

// bins are 1 MHz apart: channel offset = (m + delta) MHz with integer m, delta in [0,1)
bool build_pfb(PfbBank &b, const std::vector<double> &proto, int D, const std::vector<double> &foff_hz,
               double fs)
{
    // bins of 1 MHz: M = fs / 1 MHz must be an even integer (two output instants per symbol)
    const int M = (int)std::llround(fs / 1e6);
    if (std::fabs(fs - 1e6 * M) > 1e-3 || M < 2 || M > kPfbM || M % 2 != 0) return false;
    if ((2 * D) % M != 0) return false;
    b.M = M;
    const int nch = (int)foff_hz.size();
    const double lowoff = foff_hz[0] / 1e6;
    const double delta = lowoff - std::floor(lowoff);
    b.D = D;
    b.L = (int)proto.size();
    b.Q = (b.L + M - 1) / M;
    b.S = 2 * D / M;
    b.real_taps = std::fabs(delta) < 1e-12;
    b.taps.assign((size_t)b.Q * M * 2, 0.f);
    for (int j = 0; j < b.L; j++) {
        float wr, wi;
        phasor_turns(-delta * j / M, wr, wi);
        const double h = proto[b.L - 1 - j];
        b.taps[2 * j + 0] = (float)(h * wr);
        b.taps[2 * j + 1] = (float)(h * wi);
    }
    b.twiddle.resize(200);
    for (int m1 = 0; m1 < 10; m1++)
        for (int p2 = 0; p2 < 10; p2++) {
            float wr, wi;
            phasor_turns(-(double)(m1 * p2) / 100.0, wr, wi);
            b.twiddle[2 * (m1 * 10 + p2) + 0] = wr;
            b.twiddle[2 * (m1 * 10 + p2) + 1] = wi;
        }
    // derotation period
    long long period = 1;
    for (int c = 0; c < nch; c++) {
        // turns per output = -foff * D / fs ; foff in Hz integer
        long long num = (long long)std::llround(foff_hz[c]) * D;
        long long den = (long long)std::llround(fs);
        long long r = ((num % den) + den) % den;
        long long q = den / gcdll(r, den);
        period = period / gcdll(period, q) * q;
        if (period > 4096) return false;
    }
    b.rot_period = (int)period;
    b.binpos.resize(nch);
    b.binnat.resize(nch);
    b.rho.resize((size_t)nch * 2);
    b.rho_real = true;
    b.krot.resize((size_t)nch * period * 2);
    for (int c = 0; c < nch; c++) {
        const double off = foff_hz[c] / 1e6;             // (m + delta)
        const long long m = (long long)std::floor(off - delta + 0.5);
        if (std::fabs(off - (m + delta)) > 1e-9) return false;
        const int mm = (int)(((m % M) + M) % M);
        b.binpos[c] = M == kPfbM ? 10 * (mm % 10) + mm / 10 : mm;
        b.binnat[c] = mm;
        {
            // one-step rotation: exp(-j 2 pi foff D / fs), exact on the quarter-turn grid
            long long num1 = (long long)std::llround(foff_hz[c]) * D, den1 = (long long)std::llround(fs);
            long long r1 = ((num1 % den1) + den1) % den1;
            float rr1, ri1;
            phasor_turns(-(double)r1 / (double)den1, rr1, ri1);
            b.rho[2 * c] = rr1; b.rho[2 * c + 1] = ri1;
            if (ri1 != 0.f || std::fabs(rr1) != 1.f) b.rho_real = false;
        }
        // C_m = exp(+j 2 pi (m+delta)(L-1)/M)
        const double cturn = off * (b.L - 1) / M;
        for (long long t = 0; t < period; t++) {
            // exact rational rotation: -(foff*D*t mod fs)/fs
            long long num = (long long)std::llround(foff_hz[c]) * D;
            long long den = (long long)std::llround(fs);
            __int128 r = ((__int128)num * t) % den;
            double rturn = -(double)(long long)r / (double)den;
            double cr = std::cos(2 * kPi * (cturn - std::floor(cturn))), ci = std::sin(2 * kPi * (cturn - std::floor(cturn)));
            float rr, ri;
            phasor_turns(rturn, rr, ri);
            // (cr + j ci) * (rr + j ri); C_m on a quarter-turn grid is handled by phasor_turns too
            float c_r, c_i;
            phasor_turns(cturn, c_r, c_i);
            (void)cr; (void)ci;
            b.krot[((size_t)c * period + t) * 2 + 0] = c_r * rr - c_i * ri;
            b.krot[((size_t)c * period + t) * 2 + 1] = c_r * ri + c_i * rr;
        }
    }
    // eight bins, eight consecutive channels (8 Msps): fold the bin of channel 0 into the branch taps,
    //   a[j] e^{-j 2 pi r j / M} with r = bin(0)  <=>  bin(c) -> bin(c) - r = c,
    // so that the kernel can run a natural-order radix-2 FFT instead of the M x nch product
    b.natural = false;
    if (M == 8 && nch == 8) {
        bool consecutive = true;
        for (int c = 0; c < nch; c++) consecutive = consecutive && b.binnat[c] == (b.binnat[0] + c) % M;
        if (consecutive) {
            const int r = b.binnat[0];
            b.real_taps = true;
            for (int j = 0; j < b.L; j++) {
                float wr, wi;
                phasor_turns(-(delta + r) * j / M, wr, wi);
                const double h = proto[b.L - 1 - j];
                b.taps[2 * j + 0] = (float)(h * wr);
                b.taps[2 * j + 1] = (float)(h * wi);
                if (b.taps[2 * j + 1] != 0.f) b.real_taps = false;
            }
            for (int c = 0; c < nch; c++) b.binnat[c] = b.binpos[c] = c;
            b.natural = true;
        }
    }
    if (M < kPfbM) {
        b.dftw.resize((size_t)M * nch * 2);
        for (int pidx = 0; pidx < M; pidx++)
            for (int c = 0; c < nch; c++) {
                float wr, wi;
                phasor_turns(-(double)((long long)pidx * b.binnat[c] % M) / (double)M, wr, wi);
                b.dftw[((size_t)pidx * nch + c) * 2 + 0] = wr;
                b.dftw[((size_t)pidx * nch + c) * 2 + 1] = wi;
            }
    }
    b.available = true;
    return true;
}

}  // namespace

std::vector<float> pack_branch_major(const PfbBank &b)
{
    const int M = b.M, Q = b.Q;
    std::vector<float> out;
    if (b.real_taps) {
        const int QP = (Q + 3) & ~3;
        out.assign((size_t)M * QP, 0.f);
        for (int q = 0; q < Q; q++)
            for (int p = 0; p < M; p++) out[(size_t)p * QP + q] = b.taps[2 * ((size_t)q * M + p)];
    } else {
        const int QP = (Q + 1) & ~1;
        out.assign((size_t)M * QP * 2, 0.f);
        for (int q = 0; q < Q; q++)
            for (int p = 0; p < M; p++) {
                out[((size_t)p * QP + q) * 2 + 0] = b.taps[2 * ((size_t)q * M + p) + 0];
                out[((size_t)p * QP + q) * 2 + 1] = b.taps[2 * ((size_t)q * M + p) + 1];
            }
    }
    return out;
}

std::vector<uint16_t> make_dft_pass2_map(int rows, int lanes, int sweeps)
{
    // residue classes of the tasks
    std::vector<std::vector<uint16_t>> cls(16);
    for (int row = 0; row < rows; row++)
        for (int m1 = 0; m1 < 10; m1++) cls[(row + m1) & 15].push_back((uint16_t)(row << 4 | m1));
    std::vector<size_t> next(16, 0);
    std::vector<uint16_t> map((size_t)lanes * sweeps, 0xffff);
    // lane j of a half-wave: residues 0..7 on {0-3, 12-15} and {16-19, 28-31}, residues 8..15 on
    // {4-11} and {20-27}
    static const int res_of_lane[32] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
                                        0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7};
    for (size_t i = 0; i < map.size(); i++) {
        const int r = res_of_lane[i & 31];
        if (next[r] < cls[r].size()) map[i] = cls[r][next[r]++];
    }
    for (int r = 0; r < 16; r++)
        if (next[r] != cls[r].size()) return std::vector<uint16_t>();      // does not fit: caller refuses
    return map;
}

namespace {

double bessel_i0(double x)
{
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 60; k++) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

bool solve_dense(std::vector<double> &A, std::vector<double> &b, int n)
{
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)p * n + c])) p = r;
        if (std::fabs(A[(size_t)p * n + c]) < 1e-300) return false;
        if (p != c) {
            for (int k = 0; k < n; k++) std::swap(A[(size_t)c * n + k], A[(size_t)p * n + k]);
            std::swap(b[c], b[p]);
        }
        for (int r = c + 1; r < n; r++) {
            double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            if (f == 0.0) continue;
            for (int k = c; k < n; k++) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[(size_t)r * n + k] * b[k];
        b[r] = s / A[(size_t)r * n + r];
    }
    return true;
}

}  // namespace

bool exact_rows_available(const Design &des)
{
    const FilterBank &b = des.channel;
    const int D = des.d.decimation;
    if (des.segmented || b.rot_period <= 0 || D < 2 || D > 50 || b.ntp / b.blk > 14 || b.blk != D) return false;
    if (D > 25 && D != 50) return false;                                   // (instantiated: 2 .. 25 and 50)
    if (des.outs_per_slot != kExactSlotRows) return false;                 // (tiles never straddle a slot: 1250 rows = 11 tiles wherever the grid is shared)
    for (int c = 0; c < b.nch; c++)
        for (int k = 0; k < b.rot_period; k++) {                           // window k starts at grid row k * outs_per_slot
            const size_t i = ((size_t)c * b.rot_period + (size_t)(((long long)k * des.outs_per_slot) % b.rot_period)) * 2;
            if (!(std::fabs(b.rot[i]) == 1.0f && b.rot[i + 1] == 0.0f)) return false;
        }
    return true;
}

int make_fast_path(const Design &des, FastPath &fp)
{
    const btgpu_design &d = des.d;
    const double fs = des.cfg.sample_rate;
    const int nch = d.high_channel - d.low_channel + 1;
    fp = FastPath();

    // ---- channel bank: same prototype, polyphase evaluation ----
    {
        std::vector<double> proto(des.h_channel.begin(), des.h_channel.end());
        build_pfb(fp.channel, proto, d.decimation, des.channel.foff, fs);
    }
    // (the channel bank may be unavailable at other rates; the staged squelch below is generic)

    // ---- noise bank ----
    NoiseStage &ns = fp.noise;
    ns.R = 5 * d.decimation;
    if (d.samples_per_slot % ns.R != 0) return BTGPU_OK;
    ns.outs = d.samples_per_slot / ns.R;
    const int order = 6;
    std::vector<double> p(1, 1.0);
    for (int o = 0; o < order; o++) {
        std::vector<double> q(p.size() + ns.R - 1, 0.0);
        for (size_t i = 0; i < p.size(); i++)
            for (int k = 0; k < ns.R; k++) q[i + k] += p[i] / ns.R;
        p.swap(q);
    }
    ns.L1 = (int)p.size();
    ns.L3 = 80;
    const int clen = ns.L1 + ns.R * (ns.L3 - 1);
    const int nh = d.ntaps_noise;
    if (clen < nh) return BTGPU_OK;
    ns.pad = (clen - nh) / 2;
    ns.Jm = 6;
    if (ns.R == 250 && d.decimation == 50) {
        // 100 Msps: put the stage-1 grid on the channel bank's tile grid, so that the fused kernel (pfb100f.hip.h)
        // marches its staged input once for both banks: stage-1 instant u starts at
        // first_noise_sample - pad - Jm R + R u, the channel tiles at first_channel_sample - D + 1250 tile; the
        // pad nearest the centred one that makes their difference a multiple of R (any pad in [0, clen - nh] places
        // the reference filter inside the composite, the fit below adapts)
        const long long base = (long long)d.first_channel_sample - d.decimation - d.first_noise_sample + (long long)ns.Jm * ns.R;
        const int want = (int)((((-base) % ns.R) + ns.R) % ns.R);            // pad mod R
        int best = -1;
        for (int c = want; c <= clen - nh; c += ns.R)
            if (best < 0 || std::abs(c - (clen - nh) / 2) < std::abs(best - (clen - nh) / 2)) best = c;
        if (best >= 0) ns.pad = best;
    }
    std::vector<double> tgt(clen, 0.0);
    for (int i = 0; i < nh; i++) tgt[ns.pad + i] = des.h_noise[i];
    // normal equations (Toeplitz autocorrelation of the prototype at lags of R)
    std::vector<double> ac(ns.L3, 0.0);
    for (int lag = 0; lag < ns.L3; lag++) {
        double s = 0.0;
        long long sh = (long long)lag * ns.R;
        for (long long k = sh; k < ns.L1; k++) s += p[k] * p[k - sh];
        ac[lag] = s;
    }
    std::vector<double> N((size_t)ns.L3 * ns.L3), rhs(ns.L3);
    for (int i = 0; i < ns.L3; i++) {
        for (int j = 0; j < ns.L3; j++) N[(size_t)i * ns.L3 + j] = ac[std::abs(i - j)];
        double s = 0.0;
        for (int k = 0; k < ns.L1; k++) s += p[k] * tgt[(size_t)ns.R * i + k];
        rhs[i] = s;
    }
    if (!solve_dense(N, rhs, ns.L3)) return BTGPU_OK;
    ns.h3.resize(ns.L3);
    for (int i = 0; i < ns.L3; i++) ns.h3[i] = (float)rhs[i];
    {
        std::vector<double> comp(clen, 0.0);
        for (int i = 0; i < ns.L3; i++)
            for (int k = 0; k < ns.L1; k++) comp[(size_t)ns.R * i + k] += (double)ns.h3[i] * p[k];
        double e = 0, nrm = 0;
        for (int k = 0; k < clen; k++) e += std::fabs(comp[k] - tgt[k]);
        for (int k = 0; k < nh; k++) nrm += std::fabs((double)des.h_noise[k]);
        ns.fit_l1_error = e / nrm;
    }
    // quadrature weights: sum_{i=0}^{noise_out-1} f[i] from samples f[U*J], U = R / decim = 5
    const int U = ns.R / d.decimation;
    const int half = ns.Jm * U;
    std::vector<double> phi(2 * half + 1);
    {
        const double beta = 10.0, fcut = 0.8 / (2.0 * U);   // cycles/sample at the fine rate
        double sum = 0;
        for (int n = 0; n <= 2 * half; n++) {
            double x = n - half;
            double s = x == 0 ? 2 * fcut : std::sin(2 * kPi * fcut * x) / (kPi * x);
            double r = x / half;
            double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1 - r * r))) / bessel_i0(beta);
            phi[n] = s * w;
            sum += phi[n];
        }
        for (double &v : phi) v *= U / sum;
    }
    // the reference averages noise_out (850 of a slot's 1250) fine-rate outputs: only the stage-2
    // outputs under that span carry weight
    ns.nw = (d.noise_out + U - 1) / U + 2 * ns.Jm;
    ns.weights.assign(ns.nw, 0.0);
    for (int a = 0; a < ns.nw; a++) {
        const int J = a - ns.Jm;
        double s = 0;
        for (int i = 0; i < d.noise_out; i++) {
            int idx = i - U * J + half;
            if (idx >= 0 && idx <= 2 * half) s += phi[idx];
        }
        ns.weights[a] = s;
    }
    // stage 1: polyphase bank on the noise offsets (+790 kHz) where the 100-bin kernel applies,
    // otherwise a direct-form bank with the (much shorter) B-spline prototype at hop R
    build_pfb(ns.pfb, p, ns.R, des.noise.foff, fs);
    {
        std::vector<float> pf(p.begin(), p.end());
        build_direct_bank(ns.direct, pf, d.low_channel, nch, 790000.0, des.cfg.center_freq, fs, ns.R);
    }
    ns.available = true;
    (void)nch;
    return BTGPU_OK;
}

}  // namespace btgpu
