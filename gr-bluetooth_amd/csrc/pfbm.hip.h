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
    float gain;
    float2 *Z; long long zstride;
};

inline size_t pfbm_lds_bytes(int M, int D, int Q, int nsel, bool chan)
{
    const int tt = pfbm_tile(M), nt = tt + (chan ? 1 : 0);
    const int span = 2 * ((D * (nt - 1) + Q * M + 3) / 2);
    size_t cf = (size_t)span + (size_t)Q * M + (size_t)M * nsel + (size_t)nt * (M + 1);
    size_t fl = chan ? (size_t)tt * nsel + 2 * 256 : 0;
    return cf * sizeof(float2) + fl * sizeof(float) + 64;
}

template <bool REAL, bool CHAN>
__global__ __launch_bounds__(kPfbmThreads) void pfbm_kernel(PfbmParams p)
{
    constexpr int NTH = kPfbmThreads;
    const int TT = p.TT, NT = TT + (CHAN ? 1 : 0);
    const int M = p.M, D = p.D, Q = p.Q, nsel = p.nsel;
    const int UST = M + 1;                                       // odd pitch for even M: lanes (t, p) of phase A spread over the banks
    HIP_DYNAMIC_SHARED(float4, lds4)
    cf *lds = (cf *)lds4;
    const int N4 = (D * (NT - 1) + Q * M + 3) / 2;               // 16-byte pieces of the input span
    cf *xs = lds;                                                // [2 N4]
    cf *s_taps = xs + 2 * N4;                                    // [Q M]
    cf *s_w = s_taps + Q * M;                                    // [M][nsel]
    cf *U = s_w + M * nsel;                                      // [NT][UST]
    float *s_d = (float *)(U + NT * UST);                        // [TT][nsel] angles on their way to d (CHAN)
    float *s_part = s_d + TT * nsel;                             // [256][2] run sums (CHAN)
    const int l = threadIdx.x;
    const int tile = blockIdx.x;
    const long long t0 = (long long)tile * TT - (CHAN ? 1 : 0);  // global instant of local 0

    // ---- stage the input span (16-byte aligned start, every load unconditional) and the tables
    const long long gs = p.x0 + (long long)D * t0;
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
    __syncthreads();

    // ---- phase B: DFT rows + epilogue, lane = (channel c, run of R instants)
    if (CHAN) {
        const int R = nsel * TT <= NTH ? 1 : (nsel * TT + NTH - 1) / NTH;      // instants per lane
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
            auto bin = [&](int tl) {
                const cf *u = U + tl * UST;
                const cf *w = s_w + c;
                cf y = mk(0.f, 0.f);
                for (int pp = 0; pp < M; pp++) y = cmulf(u[pp], w[pp * nsel]) + y;
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
        for (int i = l; i < NT * nsel; i += NTH) {
            const int c = i / NT, tl = i - c * NT;
            const long long t = t0 + tl;
            if (t >= p.T) continue;
            const cf *u = U + tl * UST;
            const cf *w = s_w + c;
            cf y = mk(0.f, 0.f);
            for (int pp = 0; pp < M; pp++) y = cmulf(u[pp], w[pp * nsel]) + y;
            const cf kr = ((const cf *)p.krot)[(size_t)c * p.rot_period + (int)(t % p.rot_period)];
            ((cf *)p.Z)[(size_t)c * p.zstride + t] = cmulf(y, kr);
        }
    }
}

}  // namespace btgpu
