// pfbm.hip.h -- polyphase channelizer for the capture rates below 100 Msps (M = fs / 1 MHz bins, even,
// 4..50: 4, 8, 10, 16, 20, 50 Msps ...).  Same algebra as pfb100.hip.h,
//   u_p[t] = sum_q a[M q + p] x[x0 + D t + M q + p],     Y_c[t] = sum_p u_p[t] W[p][c],   W = e^{-j 2 pi p m_c / M},
// i.e. the reference's per-channel "complex band-pass FIR, decimate, de-rotate"
// (freq_xlating_fir_filter_ccf [EXT], lib/multi_block.cc:204,275) for all channels of the capture at
// once: Q = 7 multiply-adds per branch and instant instead of ntaps per channel and instant.  With so few
// bins the DFT is the plain M x nch product (an FFT would save nothing below M = 16 and little at 20), read
// from an LDS table.  Channel bank: no de-rotation -- demod needs Y[t] conj(Y[t-1]) rho, see pfb100.hip.h
// -- angles and |Y|^2 tile sums leave in the layouts of the direct path (d[g][drow], ptile / phead for
// block_sum_kernel).  Noise bank (staged squelch, stage 1): Z[c][t] de-rotated.
//
// One workgroup = TT new output instants (250 for M <= 10, else 50; + 1 halo instant for the demod), 256 lanes:
//   0  input span -> LDS (aligned 16-byte loads, unconditional), taps and DFT table -> LDS
//   A  lane = (instant, branch): Q taps from LDS
//   B  lane = (channel, run of R instants): M-term DFT row per instant, demod against the previous instant,
//      energy sums; angles through an LDS tile, out as whole rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfb100.hip.h"

#pragma clang fp contract(fast)

namespace btgpu {

constexpr int kPfbmThreads = 256;
// new instants per tile: a divisor of the 1250 outputs of a slot, large enough that a tile is worth a workgroup
// (M * TT branch outputs) and small enough that its rows fit the LDS
inline int pfbm_tile(int M) { return M <= 10 ? 250 : 50; }

struct PfbmParams {
    const float2 *x; long long x_len; long long x0;
    int M, D, Q, TT;             // TT: new output instants per tile
    long long T;                 // output instants in total
    const float2 *taps;          // [Q*M]
    const float2 *dftw;          // [M][nsel]
    int nsel;
    const float2 *rho; int rho_real;
    const float2 *krot; int rot_period;      // noise bank; channel bank: BTGPU_FLAG_DEBUG_Y only
    int ntiles;
    float *d; int drow;          // [T][drow] time-major
    double *ptile, *phead; int tiles_per_block, tail;
    double *pfine;               // [nsel][ntiles * TT / 25] or null: |Y|^2 sums per 25 instants (F8 form; the exact stage's burst scan,
                                 // which the 250-instant tiles of this bank are too coarse for)
    float gain;
    float2 *Z; long long zstride;
    // squelch stage 1 riding on the channel bank's staged input (the FN form of the 8-bin bank, pfbm_fuse_noise): the noise bank's
    // instants n_first + n_per_tile * tile + j, j < n_per_tile, hop n_D, 15 taps per branch, out of the same LDS span
    int n_on, n_per_tile, n_D, n_rot_period;
    int n_first;                 // tile 0's first noise instant (the banks start at different samples: tiles are paired where their spans overlap)
    int stage_lo, stage_len;     // the staged span: samples [gs + stage_lo, + stage_len) of a tile (gs = the channel bank's first sample)
    int n_ofs;                   // sample of (noise instant j = 0, branch 0, tap 0) relative to the staged span's first sample
    long long n_T;               // noise instants in total
    const float2 *n_taps, *n_krot;
    float2 *n_Z; long long n_zstride;
};

// phase B of the channel bank, lane = (run, channel): instants per run such that nsel * ceil(TT / R) <= 256 lanes
__host__ __device__ inline int pfbm_run(int nsel, int TT)
{
    int R = nsel * TT <= kPfbmThreads ? 1 : (nsel * TT + kPfbmThreads - 1) / kPfbmThreads;
    while (nsel * ((TT + R - 1) / R) > kPfbmThreads) R++;
    return R;
}

inline size_t pfbm_lds_bytes(int M, int D, int Q, int nsel, bool chan, int stage_len = 0)
{
    const int tt = pfbm_tile(M), nt = tt + (chan ? 1 : 0);
    const int span = 2 * (((stage_len > 0 ? stage_len : D * (nt - 1) + Q * M) + 3) / 2);
    size_t cf = (size_t)span + (size_t)Q * M + (size_t)M * nsel + (size_t)nt * (M + 1);
    size_t fl = chan ? (size_t)tt * nsel + 2 * 256 : 0;
    return cf * sizeof(float2) + fl * sizeof(float) + 64;
}

// MC, QC: M and Q as compile-time constants (0 = take them from the parameters): the common geometries
// (8 and 20 Msps) get fully unrolled tap and DFT loops with the lane's taps / DFT column in registers
// natural-order 8-point DFT, Y_k = sum_p u_p e^{-j 2 pi p k / 8}: radix-2 decimation in frequency, in place
__device__ __forceinline__ void fft8(cf (&u)[8])
{
    const float h = 0.70710678118654752f;
    auto mulmj = [](cf v) { return mk(v.y, -v.x); };                 // * (-j)
    cf a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = u[i] + u[i + 4]; b[i] = u[i] - u[i + 4]; }
    b[1] = mk(b[1].x + b[1].y, b[1].y - b[1].x) * mk(h, h);          // * e^{-j pi/4}
    b[2] = mulmj(b[2]);
    b[3] = mk(b[3].y - b[3].x, -b[3].x - b[3].y) * mk(h, h);         // * e^{-j 3 pi/4}
    auto dft4 = [&](cf (&v)[4], int o) {
        const cf c0 = v[0] + v[2], c1 = v[1] + v[3], d0 = v[0] - v[2], d1 = mulmj(v[1] - v[3]);
        u[o] = c0 + c1; u[o + 4] = c0 - c1; u[o + 2] = d0 + d1; u[o + 6] = d0 - d1;
    };
    dft4(a, 0);
    dft4(b, 1);
}

// F8: eight bins = the eight channels in natural order (PfbBank::natural): phase B is one lane per instant
// with the 8-point FFT above instead of the M x nch product
// FN (with F8, CHAN): squelch stage 1 (the 8-bin noise bank, 15 taps per branch, hop n_D) computed from the same staged input --
// a launch, a second read of the input and 0.42 ms of a 2 ms step at C8 (pfbm_kernel<false, false, 8, 15, true> stays the
// stand-alone form).
template <bool REAL, bool CHAN, int MC = 0, int QC = 0, bool F8 = false, bool FN = false>
__global__ __launch_bounds__(kPfbmThreads) void pfbm_kernel(PfbmParams p)
{
    static_assert(!F8 || MC == 8, "the FFT variant is the 8-bin bank");
    static_assert(!FN || (F8 && CHAN), "stage 1 rides on the 8-bin channel bank only");
    constexpr int NTH = kPfbmThreads;
    const int TT = p.TT, NT = TT + (CHAN ? 1 : 0);
    const int M = MC ? MC : p.M, D = p.D, Q = QC ? QC : p.Q, nsel = p.nsel;
    const int UST = M + 1;                                       // odd pitch for even M: lanes (t, p) of phase A spread over the banks
    HIP_DYNAMIC_SHARED(float4, lds4)
    cf *lds = (cf *)lds4;
    const int N4 = ((FN ? p.stage_len : D * (NT - 1) + Q * M) + 3) / 2;   // 16-byte pieces of the input span
    cf *xs = lds;                                                // [2 N4]
    cf *s_taps = xs + 2 * N4;                                    // [Q M]
    cf *s_w = s_taps + Q * M;                                    // [M][nsel]
    cf *U = s_w + M * nsel;                                      // [NT][UST]
    float *s_d = (float *)(U + ((NT * UST + 1) & ~1));           // [TT][nsel] angles on their way to d (CHAN), 16-byte aligned
    float *s_part = s_d + TT * nsel;                             // [256][2] run sums (CHAN)
    const int l = threadIdx.x;
    const int tile = blockIdx.x;
    const long long t0 = (long long)tile * TT - (CHAN ? 1 : 0);  // global instant of local 0

    // ---- stage the input span (16-byte aligned start, every load unconditional) and the tables
    const long long gs = p.x0 + (long long)D * t0 + (FN ? p.stage_lo : 0);
    const int c_ofs = FN ? -p.stage_lo : 0;                      // the channel bank's first sample inside the staged span
    const long long a0 = gs & ~1LL;
    const int shift = (int)(gs - a0);
    {
        const bool interior = a0 >= 0 && a0 + 2LL * N4 <= p.x_len;
        for (int i0 = 0; i0 < N4; i0 += 4 * NTH) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + l + j * NTH < N4 ? i0 + l + j * NTH : N4 - 1;
                if (interior) v[j] = ((const float4 *)(p.x + a0))[i];
                else {
                    const long long a = a0 + 2LL * i;
                    const long long ac = a < 0 ? 0 : (a < p.x_len ? a : p.x_len - 1);
                    const long long bc = a + 1 < 0 ? 0 : (a + 1 < p.x_len ? a + 1 : p.x_len - 1);
                    const float2 q0 = p.x[ac], q1 = p.x[bc];
                    const bool in0 = a >= 0 && a < p.x_len, in1 = a + 1 >= 0 && a + 1 < p.x_len;
                    v[j] = make_float4(in0 ? q0.x : 0.f, in0 ? q0.y : 0.f, in1 ? q1.x : 0.f, in1 ? q1.y : 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = i0 + l + j * NTH; if (i < N4) ((float4 *)xs)[i] = v[j]; }
        }
        for (int i = l; i < Q * M; i += NTH) s_taps[i] = ((const cf *)p.taps)[i];
        for (int i = l; i < M * nsel; i += NTH) s_w[i] = ((const cf *)p.dftw)[i];
    }
    __syncthreads();

    // ---- phase A: branch filters, lane = (instant tl, branch pp)
    if (MC && QC && NTH % (MC ? MC : 1) == 0) {
        // the branch of a lane never changes (M divides the workgroup): its taps stay in registers
        constexpr int MM = MC ? MC : 1, QQ = QC ? QC : 1;
        const int pp = l % MM;
        cf a[QQ];
#pragma unroll
        for (int q = 0; q < QQ; q++) a[q] = s_taps[q * MM + pp];
        for (int tl = l / MM; tl < NT; tl += NTH / MM) {
            const cf *z = xs + shift + c_ofs + D * tl + pp;
            cf u = mk(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const cf v = z[q * MM];
                if (REAL) u = a[q].xx * v + u;
                else { u = a[q].xx * v + u; u = mk(-a[q].y, a[q].y) * v.yx + u; }
            }
            U[tl * UST + pp] = u;
        }
        if (FN) {
            // stage 1's branch filters, lane = (instant j, branch pp): 15 complex taps straight from global memory (a lane's branch
            // never changes), results in the not yet used angle tile
            cf an[15];
#pragma unroll
            for (int q = 0; q < 15; q++) an[q] = ((const cf *)p.n_taps)[q * 8 + pp];
            cf *U2 = (cf *)s_d;                                    // [n_per_tile][9]
            for (int j = l / 8; j < p.n_per_tile; j += NTH / 8) {
                const cf *z = xs + shift + p.n_ofs + p.n_D * j + pp;
                cf u = mk(0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 15; q++) {
                    const cf v = z[q * 8];
                    u = an[q].xx * v + u; u = mk(-an[q].y, an[q].y) * v.yx + u;
                }
                U2[j * 9 + pp] = u;
            }
        }
    } else {
        for (int i = l; i < NT * M; i += NTH) {
            const int tl = i / M, pp = i - tl * M;
            const cf *z = xs + shift + D * tl + pp;
            const cf *a = s_taps + pp;
            cf u = mk(0.f, 0.f);
            for (int q = 0; q < Q; q++) {
                const cf v = z[q * M], t = a[q * M];
                if (REAL) u = t.xx * v + u;
                else { u = t.xx * v + u; u = mk(-t.y, t.y) * v.yx + u; }
            }
            U[tl * UST + pp] = u;
        }
    }
    __syncthreads();

    if (F8 && CHAN) {
        // ---- phase B, lane = instant: FFT of the lane's own U row (written back in place), then the demod
        // against the neighbour's row; |Y|^2 through an LDS tile (the dead input span) to the per-channel sums
        float *s_m = (float *)xs;                                  // [TT][8]
        cf y[8];
        if (l < NT) {
            cf *u = U + l * UST;
#pragma unroll
            for (int c = 0; c < 8; c++) y[c] = u[c];
            fft8(y);
#pragma unroll
            for (int c = 0; c < 8; c++) u[c] = y[c];
        }
        if (FN && l < p.n_per_tile) {
            // stage 1: FFT, de-rotate, eight coalesced row stores (the stand-alone noise bank's epilogue); the angle tile it read from
            // is written behind the barrier
            const long long tn = (long long)p.n_first + (long long)tile * p.n_per_tile + l;
            if (tn < p.n_T) {
                const cf *u2 = (const cf *)s_d + l * 9;
                cf z8[8];
#pragma unroll
                for (int c = 0; c < 8; c++) z8[c] = u2[c];
                fft8(z8);
                const int ph = (int)(tn % p.n_rot_period);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const cf kr = ((const cf *)p.n_krot)[(size_t)c * p.n_rot_period + ph];
                    ((cf *)p.n_Z)[(size_t)c * p.n_zstride + tn] = cmulf(z8[c], kr);
                }
            }
        }
        __syncthreads();
        if (l >= 1 && l < NT) {
            const long long t = t0 + l;
            const bool live = t < p.T;
            const DemodConst kc = demod_constants(p.gain);
            const cf *ub = U + (l - 1) * UST;
            float dd[8], mm[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const cf rho = ((const cf *)p.rho)[c];             // uniform: scalar loads
                const cf ya = y[c], yb = ub[c];
                mm[c] = live ? ya.x * ya.x + ya.y * ya.y : 0.f;
                const cf ybr = cmulf(yb, mk(rho.x, -rho.y));
                const cf pq = ybr.xx * ya + ybr.yy * mk(ya.y, -ya.x);
                dd[c] = demod_poly(kc, pq.x, pq.y);
            }
            float4 *od = (float4 *)(s_d + (l - 1) * 8), *om = (float4 *)(s_m + (l - 1) * 8);
            od[0] = make_float4(dd[0], dd[1], dd[2], dd[3]); od[1] = make_float4(dd[4], dd[5], dd[6], dd[7]);
            om[0] = make_float4(mm[0], mm[1], mm[2], mm[3]); om[1] = make_float4(mm[4], mm[5], mm[6], mm[7]);
            if (p.Z && live) {                                                              // BTGPU_FLAG_DEBUG_Y
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const cf kr = ((const cf *)p.krot)[(size_t)c * p.rot_period + (int)(t % p.rot_period)];
                    ((cf *)p.Z)[(size_t)c * p.zstride + t] = cmulf(y[c], kr);
                }
            }
        }
        __syncthreads();
        {
            const long long g1 = t0 + 1;
            const long long rows = p.T - g1 < TT ? p.T - g1 : TT;
            const int n4 = (int)rows * 2;                          // drow == nsel == 8: the tile is one contiguous piece of d
            float4 *dst = (float4 *)(p.d + (size_t)g1 * 8);
            for (int i = l; i < n4; i += NTH) dst[i] = ((const float4 *)s_d)[i];
        }
        {
            // lane = (channel, run of 8 instants)
            const int c = l & 7, run = l >> 3;
            const int hr = p.tail % TT;
            const bool want_head = hr > 0 && tile % p.tiles_per_block == p.tail / TT;
            float sum = 0.f, head = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = run * 8 + k;
                if (r < TT) {
                    const float m = s_m[r * 8 + c];
                    sum += m;
                    if (want_head && r < hr) head += m;
                }
            }
            s_part[2 * l] = sum; s_part[2 * l + 1] = head;
        }
        __syncthreads();
        if (l < 8) {
            double sacc = 0.0, hacc = 0.0;
            for (int r = 0; r < NTH / 8; r++) { sacc += (double)s_part[2 * (r * 8 + l)]; hacc += (double)s_part[2 * (r * 8 + l) + 1]; }
            p.ptile[(size_t)l * p.ntiles + tile] = sacc;
            p.phead[(size_t)l * p.ntiles + tile] = hacc;
        }
        if (p.pfine && l < 8 * (TT / 25)) {                        // lane = (channel, 25 instants): the per-instant |Y|^2 are still in LDS
            const int c = l & 7, f = l >> 3;
            double s = 0.0;
            for (int r = 25 * f; r < 25 * f + 25; r++) s += (double)s_m[r * 8 + c];
            p.pfine[(size_t)c * ((size_t)p.ntiles * (TT / 25)) + (size_t)tile * (TT / 25) + f] = s;
        }
    } else if (F8) {
        // noise bank, lane = instant: FFT, de-rotate, eight coalesced row stores
        if (l < NT && t0 + l < p.T) {
            const long long t = t0 + l;
            cf y[8];
            const cf *u = U + l * UST;
#pragma unroll
            for (int c = 0; c < 8; c++) y[c] = u[c];
            fft8(y);
            const int ph = (int)(((unsigned)(t0 % p.rot_period) + (unsigned)l) % (unsigned)p.rot_period);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const cf kr = ((const cf *)p.krot)[(size_t)c * p.rot_period + ph];
                ((cf *)p.Z)[(size_t)c * p.zstride + t] = cmulf(y[c], kr);
            }
        }
    } else
    // ---- phase B: DFT rows + epilogue, lane = (channel c, run of R instants)
    if (CHAN) {
        // instants per lane: the smallest R whose nrun = ceil(TT / R) runs of all nsel channels fit the workgroup
        // (ceil(nsel TT / NTH) alone does not guarantee it: nsel = 29, TT = 50 gives R = 6, nrun = 9, 261 lanes)
        const int R = pfbm_run(nsel, TT);
        const int nrun = (TT + R - 1) / R;
        const int run = l / nsel, c = l - run * nsel;
        const bool on = run < nrun;
        float sum = 0.f, head = 0.f;
        if (on) {
            const DemodConst kc = demod_constants(p.gain);
            const cf rho = ((const cf *)p.rho)[c];
            const int tl0 = 1 + run * R;
            const int hr = p.tail % TT;
            const bool want_head = hr > 0 && tile % p.tiles_per_block == p.tail / TT;     // block-uniform
            constexpr int MW = MC && MC <= 20 ? MC : 1;
            cf wreg[MW];                                           // this channel's DFT column (small M)
            if (MC && MC <= 20) {
#pragma unroll
                for (int pp = 0; pp < MW; pp++) wreg[pp] = s_w[pp * nsel + c];
            }
            auto bin = [&](int tl) {
                const cf *u = U + tl * UST;
                cf y = mk(0.f, 0.f);
                if (MC && MC <= 20) {
#pragma unroll
                    for (int pp = 0; pp < MW; pp++) y = cmulf(u[pp], wreg[pp]) + y;
                } else {
                    const cf *w = s_w + c;
                    for (int pp = 0; pp < M; pp++) y = cmulf(u[pp], w[pp * nsel]) + y;
                }
                return y;
            };
            cf yb = bin(tl0 - 1);
            for (int k = 0; k < R && tl0 + k < NT; k++) {
                const cf ya = bin(tl0 + k);
                const long long t = t0 + tl0 + k;
                if (t < p.T) {
                    const float m = ya.x * ya.x + ya.y * ya.y;
                    sum += m;
                    if (want_head && tl0 + k - 1 < hr) head += m;
                    const cf ybr = cmulf(yb, mk(rho.x, -rho.y));
                    const cf pq = ybr.xx * ya + ybr.yy * mk(ya.y, -ya.x);                // Y[t] conj(Y[t-1]) rho
                    s_d[(tl0 + k - 1) * nsel + c] = demod_poly(kc, pq.x, pq.y);
                    if (p.Z) {                                                            // BTGPU_FLAG_DEBUG_Y
                        const cf kr = ((const cf *)p.krot)[(size_t)c * p.rot_period + (int)(t % p.rot_period)];
                        ((cf *)p.Z)[(size_t)c * p.zstride + t] = cmulf(ya, kr);
                    }
                }
                yb = ya;
            }
        }
        s_part[2 * l] = sum; s_part[2 * l + 1] = head;
        __syncthreads();
        // angles: TT rows of nsel floats -> d[g][drow] (one row per nsel lanes; whole rows when drow == nsel)
        {
            const long long g1 = t0 + 1;
            const long long rows = p.T - g1 < TT ? p.T - g1 : TT;
            for (int i = l; i < (int)rows * nsel; i += NTH) {
                const int r = i / nsel, cc = i - r * nsel;
                p.d[(size_t)(g1 + r) * p.drow + cc] = s_d[i];
            }
        }
        if (l < nsel) {
            double s = 0.0, h = 0.0;
            for (int r = 0; r < nrun; r++) { s += (double)s_part[2 * (r * nsel + l)]; h += (double)s_part[2 * (r * nsel + l) + 1]; }
            p.ptile[(size_t)l * p.ntiles + tile] = s;
            p.phead[(size_t)l * p.ntiles + tile] = h;
        }
    } else {
        // noise bank: every (instant, channel) -> Z, de-rotated
        const unsigned ph0 = (unsigned)(t0 % p.rot_period);
        for (int c = 0; c < nsel; c++)
            for (int tl = l; tl < NT; tl += NTH) {
                const long long t = t0 + tl;
                if (t >= p.T) continue;
                const cf *u = U + tl * UST;
                const cf *w = s_w + c;
                cf y = mk(0.f, 0.f);
                for (int pp = 0; pp < M; pp++) y = cmulf(u[pp], w[pp * nsel]) + y;
                const cf kr = ((const cf *)p.krot)[(size_t)c * p.rot_period + (int)((ph0 + (unsigned)tl) % (unsigned)p.rot_period)];
                ((cf *)p.Z)[(size_t)c * p.zstride + t] = cmulf(y, kr);
            }
    }
}

}  // namespace btgpu
