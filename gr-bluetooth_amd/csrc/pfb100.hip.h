// pfb100.hip.h -- 100-bin (1 MHz spacing at 100 Msps) polyphase channelizer for gfx950.
//
// For every output instant t the bank needs, for each of the 100 polyphase branches p,
//   u_p[t] = sum_q a[100 q + p] * x[x0 + D t + 100 q + p]               (Q taps per branch)
// followed by a 100-point DFT over p (10 x 10 Cooley-Tukey).  This is algebraically the
// reference's per-channel "complex band-pass FIR, decimate, de-rotate" (freq_xlating_fir_filter_ccf
// [EXT], called from lib/multi_block.cc:204,275) for all channels at once: ~27 FMA + ~70 flop per
// input sample instead of ~2100 FMA.  The de-rotation itself is never applied on the channel
// path: quadrature demodulation (multi_block::demod, lib/multi_block.cc:158-168) only needs
// y[t] conj(y[t-1]) = Y[t] conj(Y[t-1]) * rho with the per-channel constant rho = exp(-j 2 pi f D / fs)
// (+-1 on the integer-MHz grid), and |y|^2 = |Y|^2.
//
// Work decomposition (one workgroup = NT consecutive output instants, NTH = 256 lanes):
//   0  the input span of the tile is staged through LDS with aligned 16-byte loads, all loads of
//      a lane in flight before its first LDS store; XCD-aware tile order keeps the filter-length
//      overlap of neighbouring tiles in one L2.
//   A  lanes (p, r): branch p, instants of parity r.  Because 2 D is a multiple of 100 the
//      samples a branch needs for instant t+2 are the ones of instant t shifted by S = 2D/100
//      taps: each lane keeps a Q-deep register window and reads every input sample from LDS
//      exactly once.  Rows U[t][0..99] at a pitch of 106 complex.
//   B1 lane (t, p2): 10-point DFT over p1 (p = 10 p1 + p2), twiddle, in place.  The pitch of 106
//      (= 10 mod 32 eight-byte banks) makes the stride-10 accesses of consecutive lanes hit
//      consecutive banks.
//   B2 lane (t, m1) by a host-built table (design.h make_dft_pass2_map): 80 contiguous bytes in,
//      10-point DFT over p2, bin m = m1 + 10 m2 out -- channel rows to Y[t][m] (pitch 113, in the
//      dead input tile), noise rows in place.  The table gives every 16-lane access group sixteen
//      different (t + m1) mod 16, which is what keeps both access shapes conflict-free.
//   C  epilogue.  CHANNEL bank: lane (run, channel) walks <= 9 consecutive instants of Y:
//      quadrature demod against the previous instant (odd minimax polynomial for the arctangent,
//      quadrant logic on sign bits -- no table, no compare/select chains), |Y|^2 sums; d is
//      written time-major [g][80] (the window kernel's lanes = channels read it coalesced).
//      NOISE bank: stage-1 output Z.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"

// The polyphase path is a tolerance path: let the compiler fuse a*b+c here.
#pragma clang fp contract(fast)

namespace btgpu {

constexpr int kPfbUst = 106;     // LDS pitch (complex) of the DFT rows U
constexpr int kPfbYst = 113;     // LDS pitch (complex) of the bin rows Y (= 1 mod 16)
// size (complex) of the LDS region shared by the input span, the pass-1 twiddles behind it and the bin rows
constexpr int kPfbRegion(int span, int ysz) { return (((span + 100 > ysz ? span + 100 : ysz)) + 1) & ~1; }

// Quadrature demodulation of the tolerance path: gain * atan2(pi, pr) of (pr, pi), the product
// a conj(b) already rotated by rho.  gr::fast_atan2f [EXT] is a 256-interval table with linear
// interpolation (1.3e-6 rad off the true arctangent); here an odd degree-13 minimax polynomial
// (3.4e-7 rad in float) stands in for it, and the octant / quadrant unfolding works on sign bits:
//   A = atan(min/max) in [0, pi/4],  q = pi/4 - A
//   first quadrant   a1 = pi/4 - q sgn(|pr| - |pi|)       (x-dominant: A, y-dominant: pi/2 - A)
//   left half plane  a2 = pi/2 + (a1 - pi/2) sgn(pr)       (pr < 0: pi - a1)
//   lower half plane a2 sgn(pi)
// with every constant and coefficient pre-multiplied by the gain.  (0, 0) gives 0 like the
// reference; zeros are made positive first so that their sign bits decide like ">= 0" does.
struct DemodConst { float c[7]; float q4, q2; };
__host__ __device__ __forceinline__ DemodConst demod_constants(float gain)
{
    DemodConst k;
    k.c[0] = gain * 9.9999611154e-01f; k.c[1] = gain * -3.3317368021e-01f; k.c[2] = gain * 1.9807815191e-01f;
    k.c[3] = gain * -1.3233340512e-01f; k.c[4] = gain * 7.9623641276e-02f; k.c[5] = gain * -3.3604192045e-02f;
    k.c[6] = gain * 6.8117834093e-03f;
    k.q4 = gain * 0.78539816339744831f;
    k.q2 = 2.0f * k.q4;                                           // exactly twice q4: (0, 0) comes out as 0
    return k;
}
struct PfbParams {
    const float2 *x; long long x_len; long long x0;   // x index of tap 0 for output instant 0
    int D;
    long long T;                 // output instants in total
    const float2 *taps;          // branch-major packed: [100][8] floats (real taps, Q = 7) or [100][(Q+1)&~1] complex
    const float2 *twiddle;       // [100]
    int nsel;
    const int *binpos;           // [nsel] position of the channel's bin in the in-place DFT output (noise banks)
    const int *binnat;           // [nsel] the bin itself, 0..99 (channel epilogue)
    const float2 *krot;          // [nsel][rot_period] de-rotation (noise banks; channel bank: BTGPU_FLAG_DEBUG_Y only)
    int rot_period;
    const float2 *rho;           // [nsel] per-step rotation of the channel (channel epilogue)
    int rho_real;                // every rho is +-1
    const uint16_t *b2map;       // [2][NTH] lane -> (row << 4 | m1) of DFT pass 2, 0xffff = idle
    int ntiles;
    // channel epilogue
    float *d;                    // [T][80] time-major (row stride 80 floats)
    float *dcol;                 // [ntiles][80][TT] the same angles tile by tile, channel-major inside a tile: one lane of
                                 // finish_kernel follows ONE channel, here its TT instants are 4 TT contiguous bytes
                                 // (in d they sit 320 bytes apart: a cache line per sample).  null: not written
    double *ptile;               // [nsel][ntiles]
    double *phead;               // [nsel][ntiles]  sum of the first (tail % TT) instants of each tile
    int tiles_per_block, tail, nb;
    float gain;
    DemodConst kc;               // demod_constants(gain), formed on the host (pfb100f_kernel: kernel arguments = scalar registers)
    // noise epilogue / debug copy of the de-rotated channel output
    float2 *Z; long long zstride;   // [nsel][zstride]
    // fused noise stage 1 (FUSEN): the channel tile's staged input also feeds the NU = 5 noise-bank
    // instants whose first tap lies in the tile's 1250 new samples
    const float2 *n_taps;        // [100][16] complex, branch-major
    const int *n_binpos;         // [nsel]
    const float2 *n_krot;        // [nsel][n_period]
    int n_period;
    int n_off;                   // first owned noise instant starts n_off samples after the tile start
    int n_u0;                    // noise instant index owned first by tile 0 (tile b: n_u0 + 5 b)
    int pre_tiles;               // extra tiles in front (noise grid starts earlier than the channel grid)
    long long n_T;               // noise instants in total
    float2 *n_Z; long long n_zstride;
    unsigned long long *prof;    // optional [grid][8] per-phase cycle sums of wave 0 (BTGPU_PFB_PROF diagnostics)
    int dbg;                     // timing experiments only (BTGPU_PFB_DBG): 1 no input loads, 2 no d stores, 4 no Z stores
};

// Complex values are two-float ext vectors: with contraction enabled a * b + c on them is one
// v_pk_fma_f32 (two FMAs per lane per issue); swizzles and sign flips map onto the packed
// instructions' op_sel / neg modifiers.
typedef float cf __attribute__((ext_vector_type(2)));
typedef float cf2 __attribute__((ext_vector_type(4)));       // two complex values = one 16-byte LDS access
__device__ __forceinline__ cf mk(float re, float im) { cf v = {re, im}; return v; }
__device__ __forceinline__ cf cmulf(cf a, cf b)
{
    return a.xx * b + mk(-a.y, a.y) * b.yx;
}
__device__ __forceinline__ cf mulmj(cf u) { return mk(u.y, -u.x); }     // -j u

// forward 5-point DFT (kernel e^{-j 2 pi k n / 5})
__device__ __forceinline__ void dft5(const cf x0, const cf x1, const cf x2, const cf x3, const cf x4, cf *X)
{
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const cf a1 = x1 + x4, a2 = x2 + x3, b1 = x1 - x4, b2 = x2 - x3;
    X[0] = x0 + (a1 + a2);
    const cf t1 = x0 + c1 * a1 + c2 * a2;
    const cf t2 = x0 + c2 * a1 + c1 * a2;
    const cf u1 = mulmj(s1 * b1 + s2 * b2);
    const cf u2 = mulmj(s2 * b1 - s1 * b2);
    X[1] = t1 + u1;
    X[4] = t1 - u1;
    X[2] = t2 + u2;
    X[3] = t2 - u2;
}

// forward 10-point DFT, in place on v[0..9]
__device__ __forceinline__ void dft10(cf *v)
{
    cf E[5], O[5];
    dft5(v[0], v[2], v[4], v[6], v[8], E);
    dft5(v[1], v[3], v[5], v[7], v[9], O);
    const cf w1 = mk(0.80901699437494742f, -0.58778525229247313f);
    const cf w2 = mk(0.30901699437494742f, -0.95105651629515357f);
    const cf w3 = mk(-0.30901699437494742f, -0.95105651629515357f);
    const cf w4 = mk(-0.80901699437494742f, -0.58778525229247313f);
    const cf o0 = O[0], o1 = cmulf(O[1], w1), o2 = cmulf(O[2], w2), o3 = cmulf(O[3], w3), o4 = cmulf(O[4], w4);
    v[0] = E[0] + o0; v[5] = E[0] - o0;
    v[1] = E[1] + o1; v[6] = E[1] - o1;
    v[2] = E[2] + o2; v[7] = E[2] - o2;
    v[3] = E[3] + o3; v[8] = E[3] - o3;
    v[4] = E[4] + o4; v[9] = E[4] - o4;
}

__device__ __forceinline__ float demod_poly(const DemodConst &k, float pr, float pi)
{
    const uint32_t SIGN = 0x80000000u;
    pr = pr + 0.0f; pi = pi + 0.0f;                               // -0 -> +0
    const float ax = fabsf(pr), ay = fabsf(pi);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
    const float z = mn * __builtin_amdgcn_rcpf(mx);
    const float s = z * z;
    float P = k.c[6];
    P = P * s + k.c[5]; P = P * s + k.c[4]; P = P * s + k.c[3];
    P = P * s + k.c[2]; P = P * s + k.c[1]; P = P * s + k.c[0];
    const float q = k.q4 - z * P;                                 // gain (pi/4 - atan z) >= 0
    const float dd = ax - ay;                                     // >= 0: x-dominant
    const uint32_t t1 = __float_as_uint(q) ^ (__float_as_uint(dd) & SIGN);
    const float w = -k.q4 - __uint_as_float(t1);                  // gain (a1 - pi/2) <= 0
    const uint32_t t2 = __float_as_uint(w) ^ (__float_as_uint(pr) & SIGN);
    const float a2 = k.q2 + __uint_as_float(t2);
    return __uint_as_float(__float_as_uint(a2) ^ (__float_as_uint(pi) & SIGN));
}

// The same angle for arguments that are never -0 (the caller forms them with an FMA onto +0): no canonicalising
// additions; c5 = k.c[5] in a vector register (an FMA takes one scalar operand).
__device__ __forceinline__ float demod_poly_pz(const DemodConst &k, float c5, float pr, float pi)
{
    const uint32_t SIGN = 0x80000000u;
    const float ax = fabsf(pr), ay = fabsf(pi);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
    const float z = mn * __builtin_amdgcn_rcpf(mx);
    const float s = z * z;
    float P = k.c[6];
    P = P * s + c5; P = P * s + k.c[4]; P = P * s + k.c[3];
    P = P * s + k.c[2]; P = P * s + k.c[1]; P = P * s + k.c[0];
    const float q = k.q4 - z * P;                                 // gain (pi/4 - atan z) >= 0
    const float dd = ax - ay;                                     // >= 0: x-dominant
    const uint32_t t1 = __float_as_uint(q) ^ (__float_as_uint(dd) & SIGN);
    const float w = -k.q4 - __uint_as_float(t1);                  // gain (a1 - pi/2) <= 0
    const uint32_t t2 = __float_as_uint(w) ^ (__float_as_uint(pr) & SIGN);
    const float a2 = k.q2 + __uint_as_float(t2);
    return __uint_as_float(__float_as_uint(a2) ^ (__float_as_uint(pi) & SIGN));
}

// Two of those angles, written in lockstep: each is a chain of ~20 dependent instructions, and the compiler's scheduler
// keeps two independent calls one behind the other -- a wave then waits out every link of both chains.
__device__ __forceinline__ void demod_poly_pz2(const DemodConst &k, float c5, float pr0, float pi0, float pr1, float pi1,
                                               float &a0, float &a1)
{
    const uint32_t SIGN = 0x80000000u;
    const float ax0 = fabsf(pr0), ay0 = fabsf(pi0), ax1 = fabsf(pr1), ay1 = fabsf(pi1);
    const float mx0 = fmaxf(fmaxf(ax0, ay0), 1e-37f), mx1 = fmaxf(fmaxf(ax1, ay1), 1e-37f);
    const float mn0 = fminf(ax0, ay0), mn1 = fminf(ax1, ay1);
    const float r0 = __builtin_amdgcn_rcpf(mx0), r1 = __builtin_amdgcn_rcpf(mx1);
    const float z0 = mn0 * r0, z1 = mn1 * r1;
    const float s0 = z0 * z0, s1 = z1 * z1;
    float P0 = k.c[6], P1 = k.c[6];
    P0 = P0 * s0 + c5; P1 = P1 * s1 + c5;
    P0 = P0 * s0 + k.c[4]; P1 = P1 * s1 + k.c[4];
    P0 = P0 * s0 + k.c[3]; P1 = P1 * s1 + k.c[3];
    P0 = P0 * s0 + k.c[2]; P1 = P1 * s1 + k.c[2];
    P0 = P0 * s0 + k.c[1]; P1 = P1 * s1 + k.c[1];
    P0 = P0 * s0 + k.c[0]; P1 = P1 * s1 + k.c[0];
    const float q0 = k.q4 - z0 * P0, q1 = k.q4 - z1 * P1;
    const float dd0 = ax0 - ay0, dd1 = ax1 - ay1;
    const uint32_t t10 = __float_as_uint(q0) ^ (__float_as_uint(dd0) & SIGN), t11 = __float_as_uint(q1) ^ (__float_as_uint(dd1) & SIGN);
    const float w0 = -k.q4 - __uint_as_float(t10), w1 = -k.q4 - __uint_as_float(t11);
    const uint32_t t20 = __float_as_uint(w0) ^ (__float_as_uint(pr0) & SIGN), t21 = __float_as_uint(w1) ^ (__float_as_uint(pr1) & SIGN);
    const float b0 = k.q2 + __uint_as_float(t20), b1 = k.q2 + __uint_as_float(t21);
    a0 = __uint_as_float(__float_as_uint(b0) ^ (__float_as_uint(pi0) & SIGN));
    a1 = __uint_as_float(__float_as_uint(b1) ^ (__float_as_uint(pi1) & SIGN));
}

// XCD-aware tile order: consecutive tiles (which share the filter-length halo of their input
// span) run on the same XCD so the overlap is an L2 hit.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int Q, int S, int NT, bool REAL, bool CHAN, int NTH, bool FUSEN = false>
__global__ __launch_bounds__(NTH, (FUSEN ? 3 * NTH / 256 : 1)) void pfb100_kernel(PfbParams p)
{
    constexpr int DH = S * 50;                               // hop: 2 D = S * 100
    constexpr int NQ = 15, NR = 250, NU = 5;                 // fused noise bank: taps/branch, hop, instants per tile
    static_assert(!FUSEN || (CHAN && DH == 50 && NT == 26), "fused noise stage needs the C79 channel geometry");
    // NTH = 256: four waves, one per SIMD; the fused noise branches and the second sweeps of the DFT passes
    // run behind the channel work.  NTH = 512: eight waves share one tile -- channel branches on waves 0-3,
    // noise branches on waves 4-7, every DFT pass in one sweep, epilogue runs of five instants -- the
    // tile lives half as long on the same LDS footprint, i.e. twice the waves per SIMD hide latency.
    static_assert(NTH == 256 || NTH == 512, "lane roles below are laid out for four or eight waves");
    constexpr int NSW = NTH == 256 ? 2 : 1;                  // sweeps of DFT pass 2 (host table: [NSW][NTH])
    constexpr int M = 100;
    constexpr int UST = kPfbUst, YST = kPfbYst;
    constexpr int TT = CHAN ? NT - 1 : NT;                   // new output instants per tile
    constexpr int NROWS = NT + (FUSEN ? NU : 0);             // DFT rows: channel instants, then the noise instants
    static_assert(NT % 2 == 0, "NT must be even");
    HIP_DYNAMIC_SHARED(float4, lds4)
    cf *lds = (cf *)lds4;
    constexpr int SPAN_C = DH * (NT - 1) + Q * M;            // input samples the channel instants need
    constexpr int SPAN_N = (NR - 1) + NR * (NU - 1) + NQ * M;   // ... and the owned noise instants
    constexpr int SPAN = (FUSEN && SPAN_N > SPAN_C) ? SPAN_N : SPAN_C;
    constexpr int N4 = (SPAN + 3) / 2;                       // 16-byte pieces staged (aligned start: +1 sample)
    constexpr int span = 2 * N4;                             // samples resident in LDS
    constexpr int YSZ = CHAN ? NT * YST : 0;                 // the input tile is dead after phase A -> bin rows Y
    constexpr int ASZ = kPfbRegion(span, YSZ);
    cf *xs = lds;                                            // [span]
    cf *Y = lds;                                             // [NT][YST]      (after phase A)
    cf *s_tw = lds + ASZ - 100;                              // [100] twiddles of pass 1, behind the input span (pass 2 may overwrite them)
    cf *U = lds + ASZ;                                       // [NROWS][UST]
    // no static LDS at all: the channel banks must leave room for a tail workgroup beside three of them
    cf *s_krot = U + NROWS * UST;                            // [80 * 4]  (noise-only bank)
    int *s_binpos = (int *)(s_krot + 80 * 4);                // [80]      (noise-only bank)
    float *s_part = (float *)U;                              // [NTH / 80][80][2] run sums: the channel rows of U are dead after pass 2
    float *s_d = (float *)U + (NTH / 80) * 80 * 2;           // [TT][80] angles of the tile on their way to d
    static_assert(!CHAN || ((NTH / 80) * 80 * 2) % 4 == 0, "s_d must be 16-byte aligned");
    float *s_dc = s_d + TT * 80;                             // [80][TT] the same tile channel-major (-> dcol)
    static_assert(!CHAN || (NTH / 80) * 80 * 2 + 2 * TT * 80 <= 2 * NT * UST, "run sums and the angle tiles must fit the dead DFT rows");
    const bool krot_lds = !CHAN && p.rot_period <= 4 && p.nsel <= 80;
    const int l = threadIdx.x;

    // one workgroup = one tile; XCD-aware order (pre-tiles of the fused noise bank come first)
    const int ntl = p.ntiles + (FUSEN ? p.pre_tiles : 0);
    const int tile_u = xcd_remap(blockIdx.x, ntl);
    const int tile = tile_u - (FUSEN ? p.pre_tiles : 0);

    unsigned long long tprev = p.prof ? clock64() : 0ULL;
    auto mark = [&](int k) {
        if (p.prof) {
            const unsigned long long now = clock64();
            if (l == 0) p.prof[(size_t)blockIdx.x * 8 + k] += now - tprev;   // wave 0 of each tile
            tprev = now;
        }
    };

    // Every global load of the prologue is unconditional (indices clamped, values selected
    // afterwards): a load under a lane-dependent branch is waited for at the end of that branch,
    // which would serialise the ~30 loads of a lane into as many memory round trips.
    auto load_piece_edge = [&](long long a0, int i) -> float4 {
        // 16-byte piece i (two samples) of the span starting at the even sample a0, stream edges included
        const long long a = a0 + 2 * (long long)i;
        const long long ac = a < 0 ? 0 : (a + 1 < p.x_len ? a : (p.x_len >= 2 ? p.x_len - 2 : 0));
        const long long bc = ac + 1 < p.x_len ? ac + 1 : ac;
        const float2 q0 = p.x[ac], q1 = p.x[bc];
        float4 v;
        // sample a is q0 when a was in range, sample a + 1 is q1 (or q0 when a was clamped up by one)
        const bool in0 = a >= 0 && a < p.x_len, in1 = a + 1 >= 0 && a + 1 < p.x_len;
        const float2 s0 = (a == ac) ? q0 : q1;                     // a == x_len - 1: clamped to x_len - 2
        const float2 s1 = (a + 1 == ac) ? q0 : q1;                 // a == -1: clamped to 0
        v.x = in0 ? s0.x : 0.f; v.y = in0 ? s0.y : 0.f;
        v.z = in1 ? s1.x : 0.f; v.w = in1 ? s1.y : 0.f;
        return v;
    };

    // phase-A lane roles and branch taps (fixed for the whole tile)
    cf a[Q];

    // Noise-bank roles: 500 tasks (instant i, branch pp) of 15 complex taps: branch l % 100, instants
    // 0..2 in lanes 0..99 and 3..4 in lanes 100..199 (a lane needs the taps of one branch only; an
    // even 2-2-1 split needs a second tap set per lane and measured slower)
    int nz_pp = 0, nz_i0 = 0, nz_cnt = 0;
    cf an[FUSEN ? NQ : 1];
    const int ln = NTH == 256 ? l : l - 256;                     // noise role index
    if (FUSEN && ln >= 0 && ln < 200) { nz_pp = ln % 100; nz_i0 = ln < 100 ? 0 : 3; nz_cnt = ln < 100 ? 3 : 2; }

    const int a_pp = l & 127, a_r = (l >> 7) & 1;
    const bool a_on = a_pp < M && l < 256;
    const long long t0 = (long long)tile * TT - (CHAN ? 1 : 0);   // global instant of local 0

    // roles of the later phases, fetched behind the input loads
    constexpr int NZT = FUSEN ? (80 * NU + NTH - 1) / NTH : 1;   // noise outputs per lane (phase C')
    const int nz_u0 = FUSEN ? p.n_u0 + NU * tile : 0;            // first noise instant owned by this tile
    int nz_pos[NZT]; cf nz_rot[NZT];
    uint32_t b2task = 0xffffffffu;                               // both sweeps of pass 2: lo | hi << 16
    constexpr int CH = NTH / 80;                                 // epilogue: CH runs of <= RUN instants per channel
    constexpr int RUN = (TT + CH - 1) / CH;
    const int e_chunk = l / 80, e_c = l % 80;
    const bool e_on = CHAN && e_chunk < CH && e_c < p.nsel;
    int e_pos = 0; cf e_rho = mk(1.f, 0.f);

    // ---- stage the input span.  The tile starts at the even sample a0 <= gs so that every piece is
    // a 16-byte aligned load; all loads of a lane are issued before its first LDS store (one
    // memory latency per tile instead of one per loop trip).
    const long long gs = p.x0 + (long long)DH * t0;
    const long long a0 = gs & ~1LL;
    const int shift = (int)(gs - a0);
    {
        constexpr int PER = (N4 + NTH - 1) / NTH;
        float4 v[PER];
        const bool interior = a0 >= 0 && a0 + 2LL * N4 <= p.x_len;
        if (p.dbg & 1) {
#pragma unroll
            for (int j = 0; j < PER; j++) v[j] = make_float4(1.f, 0.5f, -0.25f, 0.125f);
        } else if (interior) {                               // block-uniform: one straight run of loads
            const float4 *xb = (const float4 *)(p.x + a0);
#pragma unroll
            for (int j = 0; j < PER; j++) v[j] = xb[l + j * NTH < N4 ? l + j * NTH : N4 - 1];
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++) v[j] = load_piece_edge(a0, l + j * NTH < N4 ? l + j * NTH : N4 - 1);
        }
        // tables -> LDS and the constants of this lane's roles, issued behind the input loads:
        // memory returns in order, so the staging wait excludes them
        const cf tw = ((const cf *)p.twiddle)[l < 100 ? l : 99];
        b2task = (uint32_t)p.b2map[l] | ((uint32_t)(NSW == 2 ? p.b2map[NTH + l] : (uint16_t)0xffffu) << 16);
        // branch taps, branch-major and packed (design.h pack_branch_major): a lane's Q taps are one or two
        // 16-byte loads per four real / two complex taps instead of Q 8-byte loads a branch stride apart.
        // (Measured: the 22 per-lane table loads of a fused tile cost 10 % of the kernel by their number, not
        // their bytes -- the same loads from one address cost the same.)
        {
            const int pp = a_on ? a_pp : 0;
            if (REAL) {
                constexpr int QP = (Q + 3) & ~3;
                const float4 *tp = (const float4 *)p.taps + pp * (QP / 4);
#pragma unroll
                for (int k = 0; k < QP / 4; k++) {
                    const float4 t = tp[k];
                    if (4 * k + 0 < Q) a[4 * k + 0] = mk(t.x, 0.f);
                    if (4 * k + 1 < Q) a[4 * k + 1] = mk(t.y, 0.f);
                    if (4 * k + 2 < Q) a[4 * k + 2] = mk(t.z, 0.f);
                    if (4 * k + 3 < Q) a[4 * k + 3] = mk(t.w, 0.f);
                }
            } else {
                constexpr int QP = (Q + 1) & ~1;
                const float4 *tp = (const float4 *)p.taps + pp * (QP / 2);
#pragma unroll
                for (int k = 0; k < QP / 2; k++) {
                    const float4 t = tp[k];
                    a[2 * k] = mk(t.x, t.y);
                    if (2 * k + 1 < Q) a[2 * k + 1] = mk(t.z, t.w);
                }
            }
        }
        if (FUSEN) {
            constexpr int QP = (NQ + 1) & ~1;
            const float4 *tp = (const float4 *)p.n_taps + nz_pp * (QP / 2);
#pragma unroll
            for (int k = 0; k < QP / 2; k++) {
                const float4 t = tp[k];
                an[2 * k] = mk(t.x, t.y);
                if (2 * k + 1 < NQ) an[2 * k + 1] = mk(t.z, t.w);
            }
            // bin position and de-rotation factor of this lane's noise outputs (phase C')
            const int np = p.n_period;
            const int ph0 = ((nz_u0 % np) + np) % np;             // block-uniform
#pragma unroll
            for (int j = 0; j < NZT; j++) {
                const int i = l + j * NTH < p.nsel * NU ? l + j * NTH : p.nsel * NU - 1;
                const int c = i / NU;
                int ph = ph0 + i % NU;
                ph = ph >= np ? ph - np : ph;
                nz_pos[j] = p.n_binpos[c];
                nz_rot[j] = ((const cf *)p.n_krot)[(size_t)c * np + ph];
            }
        }
        if (CHAN) {
            const int cc = e_c < p.nsel ? e_c : p.nsel - 1;
            e_pos = p.binnat[cc];
            e_rho = ((const cf *)p.rho)[cc];
        } else {
            constexpr int NK = (80 * 4 + NTH - 1) / NTH;
            const int nkr = p.nsel * p.rot_period;
            const int bp = p.binpos[l < p.nsel ? l : p.nsel - 1];
            if (l < p.nsel && l < 80) s_binpos[l] = bp;
            if (krot_lds) {
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const int i = l + k * NTH;
                    const cf kr = ((const cf *)p.krot)[i < nkr ? i : nkr - 1];
                    if (i < nkr) s_krot[i] = kr;
                }
            }
        }
        if (l < 100) s_tw[l] = tw;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int i = l + j * NTH;
            if (i < N4) ((float4 *)xs)[i] = v[j];
        }
    }
    __syncthreads();
    mark(0);

    // ---- phase A: polyphase branch filters ----
    if (a_on && tile >= 0) {
        const int pp = a_pp, r = a_r;
        const cf *z = xs + shift + DH * r + pp;
        cf zw[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) zw[q] = z[q * M];
#pragma unroll
        for (int tau = 0; tau < NT / 2; tau++) {
            cf u = mk(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if (REAL) u = a[q].xx * zw[q] + u;
                else { u = a[q].xx * zw[q] + u; u = mk(-a[q].y, a[q].y) * zw[q].yx + u; }
            }
            U[(2 * tau + r) * UST + pp] = u;
            if (tau + 1 < NT / 2) {
#pragma unroll
                for (int q = 0; q + S < Q; q++) zw[q] = zw[q + S];
#pragma unroll
                for (int q = (Q - S > 0 ? Q - S : 0); q < Q; q++) zw[q] = z[(q + S * (tau + 1)) * M];
            }
        }
    }
    if (FUSEN) {
        // noise bank branches (taps preloaded above): nz_cnt instants of branch nz_pp -> rows NT..
        for (int i = nz_i0; i < nz_i0 + nz_cnt; i++) {
            const cf *zz = xs + shift + p.n_off + NR * i + nz_pp;
            cf u = mk(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const cf v = zz[q * M];
                u = an[q].xx * v + u;
                u = mk(-an[q].y, an[q].y) * v.yx + u;
            }
            U[(NT + i) * UST + nz_pp] = u;
        }
    }
    __syncthreads();
    mark(1);

    // ---- phase B1: DFT over p1 (p = 10 p1 + p2), twiddle e^{-j 2 pi m1 p2 / 100}, in place ----
    constexpr int NTASK = NROWS * 10;
    for (int i = l; i < NTASK; i += NTH) {
        const int row = i / 10, p2 = i - 10 * row;
        cf *col = U + row * UST + p2;
        const cf *tw = s_tw + p2;
        cf v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = col[10 * k];
        dft10(v);
        col[0] = v[0];                                           // m1 = 0: twiddle 1
#pragma unroll
        for (int k = 1; k < 10; k++) col[10 * k] = cmulf(v[k], tw[10 * k]);
    }
    __syncthreads();
    mark(2);

    // ---- phase B2: DFT over p2.  Channel rows: bin m = m1 + 10 m2 -> Y[row][m] (the input tile is
    // dead); noise rows (and the noise-only bank) stay in place: bin at position 10 m1 + m2 ----
#pragma unroll
    for (int sw = 0; sw < NSW; sw++) {
        const uint32_t task = sw == 0 ? (b2task & 0xffffu) : (b2task >> 16);
        if (task != 0xffffu) {
            const int row = (int)(task >> 4), m1 = (int)(task & 15u);
            cf2 *src = (cf2 *)(U + row * UST + 10 * m1);         // 16-byte aligned: UST and 10 m1 are even
            cf v[10];
#pragma unroll
            for (int k = 0; k < 5; k++) { const cf2 t = src[k]; v[2 * k] = t.xy; v[2 * k + 1] = t.zw; }
            dft10(v);
            if (CHAN && row < NT) {
                cf *dst = Y + row * YST + m1;
#pragma unroll
                for (int k = 0; k < 10; k++) dst[10 * k] = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) { cf2 t; t.xy = v[2 * k]; t.zw = v[2 * k + 1]; src[k] = t; }
            }
        }
    }
    __syncthreads();
    mark(3);

    // ---- phase C ----
    // owned noise instants -> stage-1 output Z (de-rotated; consumed by noise_stage2_kernel): the bins are
    // fetched here, unconditionally, and leave after the channel epilogue, whose arithmetic covers the
    // LDS round trip
    cf nz_val[NZT];
    if (FUSEN) {
#pragma unroll
        for (int j = 0; j < NZT; j++) {
            const int i = l + j * NTH < p.nsel * NU ? l + j * NTH : p.nsel * NU - 1;
            nz_val[j] = U[(NT + i % NU) * UST + nz_pos[j]];
        }
    }
    mark(4);
    if (!CHAN) {
        const uint32_t period = (uint32_t)p.rot_period;
        const uint32_t ph_t0 = (uint32_t)(((uint32_t)tile * (uint32_t)TT) % period);   // phase index of local instant 0
        for (int i = l; i < p.nsel * NT; i += NTH) {
            const int c = i / NT, tl = i % NT;
            const long long t = t0 + tl;
            if (t >= p.T) continue;
            uint32_t ph = ph_t0 + (uint32_t)tl;
            ph = ph >= period ? ph % period : ph;
            const cf kr = krot_lds ? s_krot[c * p.rot_period + ph] : ((const cf *)p.krot)[(size_t)c * p.rot_period + ph];
            const cf y = cmulf(U[tl * UST + s_binpos[c]], kr);
            ((cf *)p.Z)[(size_t)c * p.zstride + t] = y;
        }
    }
    // Channel epilogue.  Lane (run, c): channel c, <= RUN consecutive instants, the previous instant's
    // bin in registers.  Runs cover local instants 1 .. NT-1 (local 0 is the halo instant).
    if (CHAN && tile >= 0) {
        if (e_on) {
            const int tl0 = 1 + e_chunk * RUN;
            const DemodConst kc = demod_constants(p.gain);
            const cf *yc = Y + e_pos;
            // rows past the tile's last instant (the last run is shorter) read on into the DFT rows: finite
            // values that are never used
            cf y[RUN + 1];
#pragma unroll
            for (int k = 0; k <= RUN; k++) y[k] = yc[(tl0 - 1 + k) * YST];
            float sum = 0.f;
            // The TT rows of d this tile owns are one contiguous block of 320 TT bytes: the angles cross the
            // LDS tile s_d[TT][80] (lanes = channels: consecutive banks) and leave as 16-byte pieces below,
            // a quarter of the store instructions a lane-per-angle store needs
            float *drow = s_d + (tl0 - 1) * 80 + e_c;
            float *dcolp = s_dc + e_c * TT + (tl0 - 1);         // lanes = channels: pitch TT = 25 floats, odd -> all banks
            // one output: |Y|^2 into the tile sum, Y[t] conj(Y[t-1]) rho -> angle -> d[t][c]
            auto one_real = [&](int k) {
                const cf ya = y[k + 1], yb = y[k] * e_rho.xx;                  // rho = +-1
                sum += ya.x * ya.x + ya.y * ya.y;
                const cf pp = yb.xx * ya + yb.yy * mk(ya.y, -ya.x);
                const float a = demod_poly(kc, pp.x, pp.y);
                drow[k * 80] = a;
                dcolp[k] = a;
            };
            auto one_any = [&](int k) {
                const cf ya = y[k + 1], yb = cmulf(y[k], mk(e_rho.x, -e_rho.y));   // conj(Y[t-1] conj(rho)) = conj(Y[t-1]) rho
                sum += ya.x * ya.x + ya.y * ya.y;
                const cf pp = yb.xx * ya + yb.yy * mk(ya.y, -ya.x);
                const float a = demod_poly(kc, pp.x, pp.y);
                drow[k * 80] = a;
                dcolp[k] = a;
            };
            constexpr int LAST = TT - (CH - 1) * RUN;            // instants of the last run
            const bool whole = t0 + NT <= p.T;                   // block-uniform: every instant of the tile exists
            // instants of this run that exist: inside the tile and inside the stream
            const long long left = p.T - (t0 + tl0);
            int nval = NT - tl0 < RUN ? NT - tl0 : RUN;
            nval = left < nval ? (int)(left < 0 ? 0 : left) : nval;
            if (whole && p.rho_real) {
#pragma unroll
                for (int k = 0; k < LAST; k++) one_real(k);
                if (e_chunk < CH - 1) {
#pragma unroll
                    for (int k = LAST; k < RUN; k++) one_real(k);
                }
            } else if (whole) {
#pragma unroll
                for (int k = 0; k < LAST; k++) one_any(k);
                if (e_chunk < CH - 1) {
#pragma unroll
                    for (int k = LAST; k < RUN; k++) one_any(k);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RUN; k++) if (k < nval) one_any(k);
            }
            s_part[(e_chunk * 80 + e_c) * 2 + 0] = sum;
            // head sum (first tail % TT instants of the tile): only the tile that holds the end of a
            // window's last partial block is ever asked for it (block_sum_kernel)
            const int hr = p.tail % TT;
            float head = 0.f;
            if (hr > 0 && tile % p.tiles_per_block == p.tail / TT) {            // block-uniform
#pragma unroll
                for (int k = 0; k < RUN; k++)
                    if (k < nval && tl0 + k - 1 < hr) head += y[k + 1].x * y[k + 1].x + y[k + 1].y * y[k + 1].y;
            }
            s_part[(e_chunk * 80 + e_c) * 2 + 1] = head;
            if (p.Z) {                                          // BTGPU_FLAG_DEBUG_Y: the de-rotated channel output
                const int period = p.rot_period;
                int ph = (int)((t0 + tl0) % period);
                for (int k = 0; k < nval; k++) {
                    const cf kr = ((const cf *)p.krot)[e_c * period + ph];
                    ((cf *)p.Z)[(size_t)e_c * p.zstride + (t0 + tl0 + k)] = cmulf(yc[(tl0 + k) * YST], kr);
                    ph = ph + 1 == period ? 0 : ph + 1;
                }
            }
        }
    }
    if (FUSEN) {                                                 // (pre-tiles too: they own noise instants only)
#pragma unroll
        for (int j = 0; j < NZT; j++) {
            const int i = l + j * NTH;
            const int c = i / NU, ui = i % NU;
            const int u = nz_u0 + ui;
            if (i < p.nsel * NU && u >= 0 && u < p.n_T && !(p.dbg & 4))
                ((cf *)p.n_Z)[(size_t)c * p.n_zstride + u] = cmulf(nz_val[j], nz_rot[j]);
        }
    }
    if (CHAN && tile >= 0) {
        __syncthreads();
        mark(5);
        {
            // d rows of the tile: TT * 20 pieces of 16 bytes, contiguous in HBM (row stride = row size)
            const long long g1 = t0 + 1;                             // first owned instant
            const long long rows = p.T - g1 < TT ? p.T - g1 : TT;    // ... inside the stream
            float4 *dst = (float4 *)(p.d + (size_t)g1 * 80);
            if (!(p.dbg & 2))
                for (int i = l; i < (int)rows * 20; i += NTH) dst[i] = ((const float4 *)s_d)[i];
            if (p.dcol && !(p.dbg & 2)) {
                // the whole tile, instants past the stream included (never read): 80 TT floats = TT * 20 pieces
                static_assert((80 * TT) % 4 == 0, "tile of dcol in 16-byte pieces");
                float4 *dc = (float4 *)(p.dcol + (size_t)tile * (80 * TT));
                for (int i = l; i < TT * 20; i += NTH) dc[i] = ((const float4 *)s_dc)[i];
            }
        }
        if (l < p.nsel) {
            double sum = 0.0, head = 0.0;
            for (int k = 0; k < CH; k++) {
                sum += (double)s_part[(k * 80 + l) * 2];
                head += (double)s_part[(k * 80 + l) * 2 + 1];
            }
            p.ptile[(size_t)l * p.ntiles + tile] = sum;
            p.phead[(size_t)l * p.ntiles + tile] = head;         // first (tail % TT) instants of this tile
        }
        mark(6);
    }
}

// tile sums -> per-slot-block sums P[c][b] and block-head sums Pt[c][b] (first `tail` instants of
// block b), the layout the direct path produces.  tail may span several tiles (multi_LAP: 144).
__global__ __launch_bounds__(256) void block_sum_kernel(const double *__restrict__ ptile, const double *__restrict__ phead,
                                 int ntiles, int tiles_per_block, int tail_tiles,
                                 double *__restrict__ P, double *__restrict__ Pt, int nb, int nch)
{
    // one wave per (channel, block): lanes read the block's tile sums contiguously, then a fixed
    // shuffle tree (deterministic order)
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nb * nch) return;
    const int c = i / nb, b = i % nb;
    double s = 0.0, h = 0.0;
    for (int k = lane; k < tiles_per_block; k += 64) {
        const int t = b * tiles_per_block + k;
        if (t < ntiles) {
            const double v = ptile[(size_t)c * ntiles + t];
            s += v;
            if (k < tail_tiles) h += v;
            else if (k == tail_tiles) h += phead[(size_t)c * ntiles + t];
        }
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); h += __shfl_down(h, off, 64); }
    if (lane == 0) {
        P[(size_t)c * nb + b] = s;
        Pt[(size_t)c * nb + b] = h;
    }
}

// ------------------------------------------------------------------------------------
// Noise stage 2: y^[J] = sum_i h3[i] Z[c][J + i]; E_off * noise_out = sum_J w[J - outs k] |y^[J]|^2
// over slot k's nw stage-2 outputs (quadrature weights incl. the band-limited edge correction;
// neighbouring slots share 2 Jm outputs).  One workgroup = one channel, KS consecutive slots:
// the stage-1 samples of the run are staged once, every y^ is computed once (six adjacent
// outputs per lane, conflict-free 16-byte LDS reads, wave-uniform taps), |y^|^2 goes to LDS and 32 lanes
// per slot form its weighted sum in double.
// ------------------------------------------------------------------------------------
constexpr int kS2Slots = 8;
// The tile sums -> block sums reduction (block_sum_kernel) rides along as `rows` extra rows of workgroups (blockIdx.y < rows,
// the channels follow): 0.05 ms of launch ramp and tail for a few microseconds of work when it runs as a kernel of its own.
struct BlockSumArgs {
    const double *ptile = nullptr, *phead = nullptr;
    int ntiles = 0, tiles_per_block = 0, tail_tiles = 0;
    double *P = nullptr, *Pt = nullptr;
    int nb = 0, nch = 0;
    int rows = 0;                    // extra grid rows that do this work (0: none)
};
constexpr int kS2SumRows = 16;
// LDS of noise_stage2_kernel: the stage-1 samples of a run of slots (+ the taps' reach + the read-ahead of the last lane,
// even) and one float per output
__host__ __device__ constexpr int s2_samples(int outs, int nw, int L3) { return (outs * (kS2Slots - 1) + nw + L3 + 8 + 1) & ~1; }
__host__ __device__ constexpr size_t s2_lds_bytes(int outs, int nw, int L3)
{
    return (size_t)s2_samples(outs, nw, L3) * sizeof(float2) + (size_t)(outs * (kS2Slots - 1) + nw + 4) * sizeof(float);
}
__global__ __launch_bounds__(256) void noise_stage2_kernel(
    const float2 *__restrict__ Z, long long zstride, int outs, int nw, int L3,
    const float *__restrict__ h3, const double *__restrict__ w, double *__restrict__ Qn, int S, BlockSumArgs bs)
{
    HIP_DYNAMIC_SHARED(float4, lds4)
    if ((int)blockIdx.y < bs.rows) {
        // (the FIRST rows of the grid: dispatched first, they run beside the stage-2 workgroups -- as the last rows they
        // were a tail of their own and the launch took as long as the two kernels it replaced)
        // the arithmetic of block_sum_kernel (one wave per (channel, block), lane k = tile k of the block, fixed shuffle
        // tree), strided over the pairs -- with 16 lanes per pair where a block has at most 16 tiles (100 Msps: 10), four
        // pairs per wave and pass: the tree's upper levels only add the zeros of the idle lanes, so the sums are the same
        // bit for bit, and a wave makes a quarter of the passes (each one a dependent load from HBM: with one pair per
        // wave these rows, ~100 passes per wave, outlasted the stage-2 workgroups of the launch)
        const int row = (int)blockIdx.y;
        const int g = bs.tiles_per_block <= 16 ? 16 : 64, per = 64 / g;
        const int nwaves = bs.rows * (int)gridDim.x * (int)(blockDim.x >> 6);
        const int lane = threadIdx.x & 63, k = lane % g, sub = lane / g;
        const int wave = (row * (int)gridDim.x + (int)blockIdx.x) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        for (int i0 = wave * per; i0 < bs.nb * bs.nch; i0 += nwaves * per) {
            const int i = i0 + sub;
            const bool on = i < bs.nb * bs.nch;
            const int c = on ? i / bs.nb : 0, b = on ? i % bs.nb : 0;
            double s = 0.0, h = 0.0;
            for (int kk = k; kk < bs.tiles_per_block; kk += g) {
                const int t = b * bs.tiles_per_block + kk;
                if (on && t < bs.ntiles) {
                    const double v = bs.ptile[(size_t)c * bs.ntiles + t];
                    s += v;
                    if (kk < bs.tail_tiles) h += v;
                    else if (kk == bs.tail_tiles) h += bs.phead[(size_t)c * bs.ntiles + t];
                }
            }
            for (int off = g / 2; off > 0; off >>= 1) { s += __shfl_down(s, off, g); h += __shfl_down(h, off, g); }
            if (k == 0 && on) {
                bs.P[(size_t)c * bs.nb + b] = s;
                bs.Pt[(size_t)c * bs.nb + b] = h;
            }
        }
        return;
    }
    const int k0 = blockIdx.x * kS2Slots, c = (int)blockIdx.y - bs.rows;
    const int ks = (S - k0) < kS2Slots ? (S - k0) : kS2Slots;        // slots in this run
    const int nout = outs * (ks - 1) + nw;                            // y^ needed
    const int need = nout + L3 - 1;                                   // stage-1 samples needed
    float2 *zs = (float2 *)lds4;                                      // [need + 8]
    float *m2 = (float *)(zs + s2_samples(outs, nw, L3));             // [nout]
    const float2 *z = Z + (size_t)c * zstride + (long long)k0 * outs;
    __shared__ float h3s[128];                                        // taps (L3 <= 128), read as LDS broadcasts
    for (int i = threadIdx.x; i < L3; i += blockDim.x) h3s[i] = h3[i];
    // unconditional (clamped) loads in batches of five, so that a lane's loads are in flight
    // together instead of one memory round trip per element
    for (int base = 0; base < need + 8; base += 5 * (int)blockDim.x) {
        float2 v[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = base + k * (int)blockDim.x + (int)threadIdx.x;
            v[k] = z[i < need ? i : need - 1];
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = base + k * (int)blockDim.x + (int)threadIdx.x;
            if (i < need + 8) zs[i] = i < need ? v[k] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    // SIX adjacent outputs per lane: one 16-byte LDS read (two samples) feeds 24 FMAs, taps ascending per output (the
    // order of the two- and four-output forms: the same sums bit for bit).  Six, not four or eight: neighbouring
    // lanes then read 16-byte chunks THREE apart, and the four groups of 16 lanes a ds_read_b128 is serviced in
    // ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) land on 16 distinct chunks mod 16 -- no bank conflict, no
    // swizzle.  (Four per lane, round 2: chunks two apart, every read a two-way conflict -- 34 % of the kernel's LDS
    // cycles -- and the kernel LDS-bound; eight per lane with an XOR swizzle: conflict-free too, but five address
    // instructions per read and only 184 of 256 lanes busy: slower than four.)  A slot's lanes start 3 * 31 = 93
    // chunks after the previous slot's and the slots are outs / 2 = 125 chunks apart: 32 more, a multiple of 16, so the
    // pattern holds across slot boundaries at 100 Msps; other geometries only lose the guarantee, not correctness.
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *zv = (const v4f *)zs;
    // only the outputs a slot's quadrature weights touch are formed: nw (182) of every `outs` (250) -- the
    // squelch averages the reference's noise_out = 850 of the 1250 2-Msps outputs of a slot.  Compact index
    // jc -> (slot jc / nwp, output jc % nwp) with nwp = nw rounded up to six: 31 lanes per slot, 248 of 256 lanes busy.
    const int nwp = (nw + 5) / 6 * 6;
    for (int jc = 6 * threadIdx.x; jc < ks * nwp; jc += 6 * blockDim.x) {
        const int sl = jc / nwp, r = jc - sl * nwp;
        const int j = sl * outs + r;                                 // even: outs is even, r a multiple of six
        v2f y0 = {0.f, 0.f}, y1 = y0, y2 = y0, y3 = y0, y4 = y0, y5 = y0;   // (re, im) pairs -> v_pk_fma_f32
        const v4f *zq = zv + (j >> 1);
        v4f q0 = zq[0], q1 = zq[1], q2 = zq[2], q3;
        // taps 2m, 2m+1 against the eight samples A, B, C, D.xy; the four registers rotate through the steps (no moves)
#define BTGPU_S2STEP(A, B, C, D, mm)                                                                      \
        {                                                                                                 \
            D = zq[3 + (mm)];                                                                             \
            const float ha = h3s[2 * (mm)], hb = h3s[2 * (mm) + 1];                                       \
            const v2f a = {ha, ha}, b = {hb, hb};                                                         \
            y0 = __builtin_elementwise_fma(a, A.xy, y0); y0 = __builtin_elementwise_fma(b, A.zw, y0);     \
            y1 = __builtin_elementwise_fma(a, A.zw, y1); y1 = __builtin_elementwise_fma(b, B.xy, y1);     \
            y2 = __builtin_elementwise_fma(a, B.xy, y2); y2 = __builtin_elementwise_fma(b, B.zw, y2);     \
            y3 = __builtin_elementwise_fma(a, B.zw, y3); y3 = __builtin_elementwise_fma(b, C.xy, y3);     \
            y4 = __builtin_elementwise_fma(a, C.xy, y4); y4 = __builtin_elementwise_fma(b, C.zw, y4);     \
            y5 = __builtin_elementwise_fma(a, C.zw, y5); y5 = __builtin_elementwise_fma(b, D.xy, y5);     \
        }
        int m = 0;
        for (; m + 4 <= L3 / 2; m += 4) {                            // L3 is even
            BTGPU_S2STEP(q0, q1, q2, q3, m)
            BTGPU_S2STEP(q1, q2, q3, q0, m + 1)
            BTGPU_S2STEP(q2, q3, q0, q1, m + 2)
            BTGPU_S2STEP(q3, q0, q1, q2, m + 3)
        }
        for (; m < L3 / 2; m++) {
            BTGPU_S2STEP(q0, q1, q2, q3, m)
            q0 = q1; q1 = q2; q2 = q3;
        }
#undef BTGPU_S2STEP
        const v2f yy[6] = {y0, y1, y2, y3, y4, y5};
#pragma unroll
        for (int k = 0; k < 6; k++) if (r + k < nw) m2[j + k] = (yy[k].x * yy[k].x) + (yy[k].y * yy[k].y);
    }
    __syncthreads();
    // slot sums: 32 lanes per slot, fixed order (lane partial sums combined by shuffles)
    const int s = threadIdx.x >> 5, lane = threadIdx.x & 31;
    if (s < ks) {
        double acc = 0.0;
        const float *mm = m2 + s * outs;
        for (int j = lane; j < nw; j += 32) acc += w[j] * (double)mm[j];
        for (int off = 16; off > 0; off >>= 1) acc += __shfl_down(acc, off, 32);
        if (lane == 0) Qn[(size_t)c * S + k0 + s] = acc;
    }
}

}  // namespace btgpu
