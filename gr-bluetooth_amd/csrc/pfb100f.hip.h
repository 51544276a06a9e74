// pfb100f.hip.h -- the C79 hot kernel: 100-bin polyphase channelizer + squelch stage 1 + quadrature demod,
// one workgroup = a RUN of KT tiles of 25 output instants, taken with stride `nruns` (round 3).
//
// Same algebra and the same LDS layouts as pfb100_kernel<7,1,26,REAL,true,NTH,true> (pfb100.hip.h, which stays
// the kernel of the non-fused banks): what the reference computes per channel with freq_xlating_fir_filter_ccf
// [EXT] (lib/multi_block.cc:204 channel filter, :275 noise filter) and multi_block::demod (:158-168), for all 79
// channels at once.  What changed, and why (measured on the round-2 kernel, profiles/r02_d_*):
//
//  * 23 % of a tile's life was the wait for its input span (and for ~30 per-lane table loads behind it), with only
//    three tiles per CU to hide it.  Here a workgroup walks KT tiles: branch taps, lane roles and twiddles are
//    fetched ONCE, and the input of the next tile is loaded into registers right after the first barrier of the
//    current one -- a whole tile of arithmetic lies between the loads and their first use.  Tile k of workgroup b
//    is b + k * nruns (not b * KT + k): neighbouring tiles stay concurrent, so the input overlap of two tiles and
//    the two halves of a shared Z line still meet in L2 (the consecutive order re-fetched 2.06 GB and wrote
//    3.2 GB per launch; the strided one is back at 1.16 / 2.44 GB, profiles/r03_j_pmc_hbm.json).
//  * Phase A read the staged input 3.3 times (channel branches once with a register window, the five noise
//    instants of the tile 15 taps each straight from LDS: 7500 of the 10 750 eight-byte reads per tile).  The
//    host now places the squelch stage-1 grid ON the channel grid (design_fast.cc: n_off = 0), so that noise
//    instant i, branch p, tap q needs the sample the channel lane (p, r = i mod 2) holds at march step
//    q + (5 i - r) / 2: one march of 25 (r = 0) / 22 (r = 1) reads per lane serves 13 channel instants and 3 / 2
//    noise instants.
//  * The global stores of tile n (Z, d, dcol, tile sums) are issued behind the staging of tile n+1, so that the
//    wait for the prefetched input never sits behind fresh stores; d / dcol / Z go out non-temporal.
//  * NTH = 256 with three workgroups per CU (<= 168 VGPRs) is the default.  NTH = 320 (one sweep per DFT pass
//    instead of two) is kept as BTGPU_BANK=run320 for A/B: it still spills and is slower (profiles/r03_h_*).
//  * The lane index is laundered once per tile (BTGPU_OPAQUE): without it LICM hoists every per-phase address out
//    of the tile loop and the kernel spills ~80 registers.
//  * Second half of round 3 (the default, OPT 255): a lean epilogue (27 instead of 41 vector instructions per
//    demodulated instant, instants in lockstep pairs), stores and prefetch loads addressed as uniform base + 32-bit
//    lane offset, and a wave priority per phase (staging 2, march 0, DFT passes 3, epilogue 1): three workgroups in
//    different phases share a CU, and a tile's life is the latency of its waves' dependent chains (14 300 cycles with
//    the workgroup alone on the CU, 19 500 with three: profiles/r03_l_*), not issue slots -- the phases that end in a
//    barrier the whole workgroup waits at go first.  502 M -> 417 M vector instructions, 1.34 -> 1.20 ms on one box.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfb100.hip.h"

#pragma clang fp contract(fast)

namespace btgpu {

constexpr int kBankKT = 5;                   // tiles per workgroup of the KT-5 variants (the default launches KT = 10)

// One march of the staged input for branch pp and instant parity R (wave-uniform): channel instants 2 tau + R,
// tau = 0..12, out of a 7-deep register window; noise instants i = R, R + 2, .. from the same samples:
//   noise_i[pp] = sum_q an[q] z[q + s_i],  s_i = (5 i - R) / 2        (n_off = 0: the two grids coincide)
// Multiply-accumulates of the branch filters, written with scalar FMAs on purpose: a packed v_pk_fma_f32 issues at
// half rate on this chip (scripts/ubench/valu_rates.hip), so two packed FMAs cost what four scalar ones do -- but the
// packed form needs the tap as (re, re) and (-im, im) register pairs, which this compiler materialises instead of
// using op_sel: 60 extra registers for the 15 squelch taps, and with them the next tile's prefetched input in scratch.
__device__ __forceinline__ cf cmac(cf acc, cf tap, cf z)          // acc + tap * z
{
    acc.x = fmaf(tap.x, z.x, acc.x); acc.x = fmaf(-tap.y, z.y, acc.x);
    acc.y = fmaf(tap.x, z.y, acc.y); acc.y = fmaf(tap.y, z.x, acc.y);
    return acc;
}
__device__ __forceinline__ cf rmac(cf acc, float tap, cf z)       // acc + tap * z, real tap
{
    acc.x = fmaf(tap, z.x, acc.x); acc.y = fmaf(tap, z.y, acc.y);
    return acc;
}
// Channel taps of a lane: seven real taps (REAL) or seven complex ones
template <bool REAL> struct ChanTaps { float re[7]; float im[REAL ? 1 : 7]; };
template <bool REAL>
__device__ __forceinline__ cf chan_mac(const ChanTaps<REAL> &t, int q, cf w, cf u)
{
    if (REAL) return rmac(u, t.re[q], w);
    return cmac(u, mk(t.re[q], t.im[q]), w);
}

// An ordering-only dependence (no instruction is emitted): `x` cannot be formed before `dep` exists.  Used to keep the
// LDS reads of the march a fixed number of steps ahead of the arithmetic -- left alone, the scheduler issues all 25 reads
// of the unrolled march up front (50 registers), and the register allocator answers by evicting the next tile's
// prefetched input to scratch memory, which puts a full memory round trip back on every tile's critical path.
// BTGPU_OPAQUE(x): the value of x is unknown to the optimiser from here on (no instruction either).  The lane index goes
// through it at the top of every tile: everything derived from it -- LDS and HBM addresses of every phase -- is then
// recomputed per tile (a few dozen integer operations) instead of being hoisted out of the tile loop, where ~40 such
// invariants stayed live through all phases and pushed the prefetched input out of the register file.
// (the two macros live in kernels.hip.h)

// DO_CH / DO_SQ: the eight-wave form (NTH = 512) gives the channel branches and the squelch branches to different waves
template <int R, bool REAL, int OPT, bool DO_CH = true, bool DO_SQ = true>
__device__ __forceinline__ void march_branch(const cf *z, const ChanTaps<REAL> &a, const cf (&an)[15], cf *U, int pp)
{
    constexpr int M = 100, Q = 7, NT = 26, NQ = 15, UST = kPfbUst;
    constexpr int NI = R == 0 ? 3 : 2;                           // noise instants of this parity: 0,2,4 / 1,3
    // march steps: max(12 + 7, s_last + 15); the channel instants alone end at step 19, the squelch instants start at s_0 = R
    constexpr int NN = DO_SQ ? (R == 0 ? 25 : 22) : (NT / 2 - 1 + Q);
    constexpr int N0 = DO_CH ? 0 : (R == 0 ? 0 : 2);
    constexpr int LA = ((OPT & 2) && DO_CH && DO_SQ) ? 8 : 4;    // LDS reads in flight ahead of the arithmetic (the eight-wave form has 80 registers)
    // Sample-major: a sample is read once and feeds, at once, every sum it belongs to -- up to seven channel instants
    // (tap n - tau of instant tau) and up to three squelch instants (tap n - s_i).  No register window; the live state is
    // the (at most) seven + three accumulators, and neighbouring sums are independent work for the VALU.
    cf acc[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) acc[k] = mk(0.f, 0.f);
    cf u[NT / 2];
    cf zq[LA];
    int zo = 0;                                                  // always 0: the handle the ordering dependence is hung on
    auto rd = [&](int n) { return z[n * M + zo]; };
#pragma unroll
    for (int k = 0; k < LA; k++) zq[k] = rd(N0 + k);
#pragma unroll
    for (int n = N0; n < NN; n++) {
        const cf zn = zq[(n - N0) % LA];
        float last = zn.x;
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = R + 2 * k, s = (5 * i - R) / 2, q = n - s;
            if (DO_SQ && q >= 0 && q < NQ) { acc[k] = cmac(acc[k], an[q], zn); last = acc[k].y; }
        }
#pragma unroll
        for (int tau = 0; tau < NT / 2; tau++) {
            const int q = n - tau;
            if (!DO_CH) continue;
            if (q == 0) u[tau] = mk(0.f, 0.f);
            if (q >= 0 && q < Q) {
                if (REAL && (OPT & 4)) u[tau] = a.re[q] * zn + u[tau];       // packed form (A/B of the issue rates)
                else u[tau] = chan_mac<REAL>(a, q, zn, u[tau]);
                last = u[tau].y;
            }
            if (q == Q - 1) U[(2 * tau + R) * UST + pp] = u[tau];
        }
        if (n + LA < NN) {
            BTGPU_AFTER(zo, last);                               // read n + LA only once step n has been worked off
            zq[(n - N0) % LA] = rd(n + LA);
        }
    }
    if (DO_SQ) {
#pragma unroll
        for (int k = 0; k < NI; k++) U[(NT + R + 2 * k) * UST + pp] = acc[k];
    }
}

// (exploration, OPT 256: wave priority per phase from p.dbg bits 8..23, one nibble each for staging, march, DFT passes, epilogue)
__device__ __forceinline__ void setprio_dyn(int v)
{
    if (v == 0) __builtin_amdgcn_s_setprio(0);
    else if (v == 1) __builtin_amdgcn_s_setprio(1);
    else if (v == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// OPT 512 (NTH = 512 only): four waves per SIMD, i.e. two workgroups per CU in <= 128 registers, instead of six in 80
// OPT (experiments, A/B on the device): 1 = epilogue not fenced, 2 = march reads 8 steps ahead, 4 = packed channel MACs,
// 8 = lean epilogue (needs rho = +-1: p.rho_real), 16 = epilogue instants in lockstep pairs on whole tiles, 32 = the DFT passes at raised wave priority, 64 = staging / stores too (2), 128 = epilogue at 1
template <int NTH, bool REAL, int KT, int OPT = 0>
__global__ __launch_bounds__(NTH, NTH == 512 ? ((OPT & 512) ? 4 : 6) : (NTH == 320 ? 4 : 3)) void pfb100f_kernel(PfbParams p)   // (HIP: the second number is waves per SIMD)
{
    static_assert(NTH == 256 || NTH == 320 || NTH == 512, "lane roles are laid out for four, five or eight waves");
    constexpr bool SPLIT = NTH == 512;                           // eight waves: channel branches on waves 0..3, squelch branches on waves 4..7
    constexpr int M = 100, Q = 7, DH = 50, NT = 26, TT = NT - 1, NQ = 15, NR = 250, NU = 5;
    constexpr int NROWS = NT + NU, UST = kPfbUst, YST = kPfbYst;
    constexpr int SPAN = (NR - 1) + NR * (NU - 1) + NQ * M;      // 2749 samples: the noise instants reach furthest
    constexpr int N4 = (SPAN + 3) / 2, span = 2 * N4;            // 16-byte pieces staged (aligned start: + 1 sample)
    constexpr int ASZ = kPfbRegion(span, NT * YST);
    constexpr int PER = (N4 + NTH - 1) / NTH;
    constexpr int NSW = NTH == 256 ? 2 : 1;                      // sweeps of a DFT pass
    constexpr int NTASK = NROWS * 10;
    static_assert(NSW * NTH >= NTASK, "a DFT pass must fit its sweeps");
    constexpr int CH = SPLIT ? 5 : NTH / 80, RUN = (TT + CH - 1) / CH;          // (eight waves: five runs of five instants on 400 lanes)
    constexpr int NZT = (80 * NU + NTH - 1) / NTH;
    HIP_DYNAMIC_SHARED(float4, lds4)
    cf *lds = (cf *)lds4;
    cf *xs = lds;                                                // [span]        input span of the tile
    cf *Y = lds;                                                 // [NT][YST]     bin rows (the span is dead after phase A)
    cf *U = lds + ASZ;                                           // [NROWS][UST]  DFT rows
    cf *s_tw = U + NROWS * UST;                                  // [100]         pass-1 twiddles (stored once per workgroup)
    float *s_part = (float *)U;                                  // [CH][80][2]   run sums (the channel rows of U are dead after pass 2)
    float *s_d = (float *)U + CH * 80 * 2;                       // [TT][80]      angles on their way to d
    float *s_dc = s_d + TT * 80;                                 // [80][TT]      the same tile channel-major (-> dcol)
    static_assert((CH * 80 * 2) % 4 == 0 && CH * 80 * 2 + 2 * TT * 80 <= 2 * NT * UST, "epilogue tiles must fit the dead DFT rows");
    // the lean epilogue reads up to two bin rows past the tile's last instant without a clamp (values never used): they must
    // at least lie inside this workgroup's LDS allocation (the span / bin-row region and the DFT rows are one piece)
    static_assert((NT + 2) * YST + M <= ASZ + NROWS * UST, "the epilogue's read-ahead must stay inside the allocation");
    const int l0 = threadIdx.x;

    // ---- the run of tiles of this workgroup (XCD-aware: neighbouring runs share their halo in one L2) ----
    const int ntl = p.ntiles + p.pre_tiles;                      // pre-tiles own squelch instants only
    const int nruns = (int)gridDim.x;                            // (about KT tiles each: bank_runs, bank_launch.h)
    // Tile k of this workgroup is tu0 + k nruns: at any moment the resident workgroups work on (about) CONSECUTIVE tiles,
    // like one-tile workgroups would -- so the filter-length overlap of neighbouring input spans is read by neighbours at
    // the same time (L2 hit) and the partial cache lines of Z that neighbouring tiles share are merged in the L2.  With a
    // workgroup walking consecutive tiles instead, both reuses lie a whole tile time apart, which the L2 (4 MB per XCD)
    // does not bridge under this kernel's streaming: measured 1.7-2.1 GB fetched for 1.15 GB of input and 3.2 GB written
    // for 2.4 (profiles/r03_e_pmc_hbm.json, r03_f).  The XCD remap keeps each XCD on one contiguous stretch per step.
    const int tu0 = xcd_remap(blockIdx.x, nruns);
    const int tstep = nruns;

    // ---- what a lane FETCHES for its roles, once per workgroup (what it can compute is recomputed per tile) ----
    ChanTaps<REAL> a;
    cf an[NQ];
    if (!SPLIT) {
        const int app = l0 & 127;
        const int pp = (app < M && l0 < (SPLIT ? 512 : 256)) ? app : 0;
        if (REAL) {
            const float4 *tp = (const float4 *)p.taps + pp * 2;                  // [100][8] floats
            const float4 t0 = tp[0], t1 = tp[1];
            a.re[0] = t0.x; a.re[1] = t0.y; a.re[2] = t0.z; a.re[3] = t0.w; a.re[4] = t1.x; a.re[5] = t1.y; a.re[6] = t1.z;
        } else {
            const float4 *tp = (const float4 *)p.taps + pp * 4;                  // [100][8] complex
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 t = tp[k];
                a.re[2 * k] = t.x; a.im[REAL ? 0 : 2 * k] = t.y;
                if (2 * k + 1 < Q) { a.re[2 * k + 1] = t.z; a.im[REAL ? 0 : 2 * k + 1] = t.w; }
            }
        }
        const float4 *np = (const float4 *)p.n_taps + pp * 8;                    // [100][16] complex
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 t = np[k];
            an[2 * k] = mk(t.x, t.y);
            if (2 * k + 1 < NQ) an[2 * k + 1] = mk(t.z, t.w);
        }
    }
    if (l0 < 100) s_tw[l0] = ((const cf *)p.twiddle)[l0];        // visible after the first barrier of the first tile
    uint32_t b2task = (uint32_t)p.b2map[l0];
    if (NSW == 2) b2task |= (uint32_t)p.b2map[NTH + l0] << 16; else b2task |= 0xffff0000u;
    int e_pos; cf e_rho;
    {
        const int ec = l0 % 80, ecc = ec < p.nsel ? ec : p.nsel - 1;
        e_pos = p.binnat[ecc];
        e_rho = ((const cf *)p.rho)[ecc];
    }
    int nz_pos[NZT];                                             // bin position of this lane's squelch outputs
#pragma unroll
    for (int j = 0; j < NZT; j++) {
        const int i = l0 + j * NTH < p.nsel * NU ? l0 + j * NTH : p.nsel * NU - 1;
        nz_pos[j] = p.n_binpos[i / NU];
    }
    const DemodConst &kc = p.kc;                                 // kernel arguments: scalar registers

    auto span_start = [&](int tile) -> long long { return (p.x0 + (long long)DH * ((long long)tile * TT - 1)) & ~1LL; };
    auto interior = [&](int tile) -> bool {                      // block-uniform
        const long long a0 = span_start(tile);
        return a0 >= 0 && a0 + 2LL * N4 <= p.x_len;
    };
    // Input span of a tile.  Interior tiles (all but a handful at the two ends of the stream): one straight run of
    // aligned 16-byte loads into registers, issued a whole tile ahead of their use.  Tiles that touch a stream edge
    // are staged in place when their turn comes, element by element with clamped addresses (zeros outside).
    // (named registers, not an array: an array that lives across the tile loop stays in scratch memory)
    static_assert(PER <= 6, "explicit prefetch registers below");
    float4 v0, v1, v2, v3, v4, v5;
    v0 = v1 = v2 = v3 = v4 = v5 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_span = [&](int tile, int l) {
        // uniform base + 32-bit lane offset (scalar base registers, no 64-bit lane arithmetic); only the last piece clamps
        const char *xb = (const char *)(p.x + span_start(tile));
        const unsigned o = (unsigned)l * 16u;
        auto at = [&](int j) {
            const unsigned oj = (j + 1) * NTH <= N4 ? o : (unsigned)(l + j * NTH < N4 ? l : N4 - 1 - j * NTH) * 16u;
            return *(const float4 *)(xb + (size_t)j * NTH * 16 + oj);
        };
        v0 = at(0); if (PER > 1) v1 = at(1); if (PER > 2) v2 = at(2); if (PER > 3) v3 = at(3);
        if (PER > 4) v4 = at(4); if (PER > 5) v5 = at(5);
    };
    auto stage_edge = [&](int tile, int l) {
        const long long a0 = span_start(tile);
#pragma unroll 1
        for (int i = l; i < N4; i += NTH) {
            const long long s0 = a0 + 2LL * i, s1 = s0 + 1;
            const bool in0 = s0 >= 0 && s0 < p.x_len, in1 = s1 >= 0 && s1 < p.x_len;
            const float2 q0 = p.x[in0 ? s0 : 0], q1 = p.x[in1 ? s1 : 0];
            ((float4 *)xs)[i] = make_float4(in0 ? q0.x : 0.f, in0 ? q0.y : 0.f, in1 ? q1.x : 0.f, in1 ? q1.y : 0.f);
        }
    };
    // angle tiles + tile sums of a finished tile -> HBM (reads the epilogue's LDS tiles: call between its barrier
    // and the next barrier, before phase A of the following tile writes the DFT rows again)
    auto copy_out = [&](int tile, int l) {
        const long long g1 = (long long)tile * TT;                               // first owned instant
        const long long rows = p.T - g1 < TT ? p.T - g1 : TT;                    // ... inside the stream
        // streaming stores (non-temporal): 16 KB per tile that nobody reads before the next kernel -- kept out of the L2
        // they would otherwise flush the input overlap of the next tile and the half-written lines of Z out of
        typedef float f4 __attribute__((ext_vector_type(4)));
        // TT * 20 = 500 pieces of 16 bytes per layout, two per lane: the four LDS reads first, then the four stores (a loop
        // with a run-time trip count would pay an LDS round trip in front of every store)
        static_assert(TT * 20 <= 2 * NTH, "two pieces per lane");
        const int n_d = (int)rows * 20, i1 = l + NTH;
        const f4 a0 = ((const f4 *)s_d)[l], a1 = ((const f4 *)s_d)[i1 < TT * 20 ? i1 : l];
        const f4 b0 = ((const f4 *)s_dc)[l], b1 = ((const f4 *)s_dc)[i1 < TT * 20 ? i1 : l];
        // (uniform base + 32-bit lane offset: the store takes the base from scalar registers, no 64-bit lane arithmetic)
        char *dst = (char *)(p.d + (size_t)g1 * 80);
        const unsigned o0 = (unsigned)l * 16u, o1 = (unsigned)i1 * 16u;
        if (l < n_d) __builtin_nontemporal_store(a0, (f4 *)(dst + o0));
        if (i1 < n_d) __builtin_nontemporal_store(a1, (f4 *)(dst + o1));
        if (p.dcol) {
            char *dc = (char *)(p.dcol + (size_t)tile * (80 * TT));
            if (l < TT * 20) __builtin_nontemporal_store(b0, (f4 *)(dc + o0));
            if (i1 < TT * 20) __builtin_nontemporal_store(b1, (f4 *)(dc + o1));
        }
        if (l < p.nsel) {
            double sum = 0.0, head = 0.0;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                sum += (double)s_part[(k * 80 + l) * 2];
                head += (double)s_part[(k * 80 + l) * 2 + 1];
            }
            const unsigned po = ((unsigned)l * (unsigned)p.ntiles + (unsigned)tile) * 8u;
            *(double *)((char *)p.ptile + po) = sum;
            *(double *)((char *)p.phead + po) = head;            // first (tail % TT) instants of this tile
        }
    };

    // optional per-phase cycle sums, per wave (BTGPU_PFB_PROF diagnostics; scripts/pfb_phases.py): slot k of wave w of this
    // workgroup = cycles from the previous mark to mark k, barrier waits included
    // (ten tiles per workgroup: 16 slots per wave -- marks 5.. split each interval into its work and its barrier wait)
    constexpr int PSL = KT >= 10 ? 16 : 8;
    unsigned long long tprev = p.prof ? clock64() : 0ULL;
    auto mark = [&](int k) {
        if (p.prof) {
            const unsigned long long now = clock64();
            if ((l0 & 63) == 0) p.prof[((size_t)blockIdx.x * (NTH / 64) + (l0 >> 6)) * PSL + k] += now - tprev;
            tprev = now;
        }
    };
    const int shift = (int)((p.x0 - DH) & 1LL);                  // the span starts at an even sample: same for every tile (DH TT is even)
    const int np = p.n_period;
    if (interior(tu0 - p.pre_tiles)) load_span(tu0 - p.pre_tiles, l0);
    int prev_tile = -1, prev_u0 = 0;                             // the tile whose results are waiting (LDS tiles, squelch outputs in registers)
    bool prev_any = false;
    static_assert(NZT <= 2, "two named register pairs carry the squelch outputs");
    cf zv0 = mk(0.f, 0.f), zv1 = zv0, zr0 = zv0, zr1 = zv0;      // squelch bins of this lane and their de-rotation factors
    unsigned nzu0, nzu1 = 0x80000000u, nzo0, nzo1 = 0, nzk0, nzk1 = 0;           // ui (0..4) and byte offset of (c, ui) in n_Z of this lane's outputs
    {
        auto nz_where = [&](int i, unsigned &ui, unsigned &zo, unsigned &ko) {
            ui = i < p.nsel * NU ? (unsigned)(i % NU) : 0x80000000u;
            zo = i < p.nsel * NU ? ((unsigned)(i / NU) * (unsigned)p.n_zstride + (unsigned)(i % NU)) * 8u : 0u;
            ko = i < p.nsel * NU ? (unsigned)(i / NU) * (unsigned)p.n_period : 0u;     // row of the de-rotation table
        };
        nz_where(l0, nzu0, nzo0, nzk0);
        if (NZT > 1) nz_where(l0 + NTH, nzu1, nzo1, nzk1);
    }
    auto flush_prev = [&](int ptile, int pu0, int l) {
        // squelch outputs (c, u = pu0 + ui) -> n_Z[c][u]: byte offset and ui of this lane's two outputs were formed once, in
        // front of the tile loop (32-bit: the launch checks that the plane fits), the base pointer is uniform
#pragma unroll
        for (int j = 0; j < NZT; j++) {
            const unsigned ui = j == 0 ? nzu0 : nzu1, zo = j == 0 ? nzo0 : nzo1;
            const unsigned u = (unsigned)pu0 + ui;               // lanes without an output carry ui = 2^31: never < n_T
            char *zb = (char *)p.n_Z + (long long)pu0 * 8;
            if (u < (unsigned)p.n_T)                             // (negative u wraps far above n_T)
                *(cf *)(zb + zo) = cmulf(j == 0 ? zv0 : zv1, j == 0 ? zr0 : zr1);
        }
        if (ptile >= 0) copy_out(ptile, l);
    };
    for (int tu = tu0; tu < ntl; tu += tstep) {
        const int tile = tu - p.pre_tiles;
        const long long t0 = (long long)tile * TT - 1;           // global instant of local row 0 (the halo instant)
        const int nz_u0 = p.n_u0 + NU * tile;                    // first squelch instant owned by this tile
        int l = l0;
        BTGPU_OPAQUE(l);                                         // (see the macro: keeps per-phase addresses out of the loop-invariant set)
        const int a_pp = l & 127, a_r = (l >> 7) & 1;
        const bool a_on = a_pp < M && l < (SPLIT ? 512 : 256);
        const int e_chunk = l / 80, e_c = l - 80 * e_chunk;
        const bool e_on = e_chunk < CH && e_c < p.nsel;
        // ---- input span -> LDS, FIRST: the only vector-memory operations outstanding at this point are the prefetch loads
        // issued a whole tile ago (and the two de-rotation factors behind them).  The previous tile's stores -- squelch
        // outputs, angle tiles, tile sums -- are issued only now, behind the staging, in the same barrier interval: issued
        // in front of it (as the epilogue's last act, round-3 first form) they sat between the loads and their s_waitcnt,
        // and every tile waited for freshly issued HBM writes to retire.
        if (OPT & 256) setprio_dyn((p.dbg >> 8) & 15);
        else if (OPT & 64) __builtin_amdgcn_s_setprio(2);        // (staging + the previous tile's stores: little arithmetic)
        if (interior(tile)) {
            auto put = [&](int j, const float4 &q) { const int i = l + j * NTH; if (i < N4) ((float4 *)xs)[i] = q; };
            put(0, v0); if (PER > 1) put(1, v1); if (PER > 2) put(2, v2); if (PER > 3) put(3, v3);
            if (PER > 4) put(4, v4); if (PER > 5) put(5, v5);
        } else stage_edge(tile, l);
        if (PSL > 8) mark(5);
        if (prev_any) flush_prev(prev_tile, prev_u0, l);
        if (PSL > 8) mark(6);
        prev_tile = tile; prev_any = true;
        __syncthreads();
        if (OPT & 256) setprio_dyn((p.dbg >> 12) & 15);
        else if (OPT & 64) __builtin_amdgcn_s_setprio(0);
        mark(0);
        // the next tile's input: in flight under this tile's arithmetic (eight waves: requested behind the march -- the three
        // registers per lane it lands in are what the march's tap registers need)
        if (!SPLIT && tu + tstep < ntl && interior(tile + tstep)) load_span(tile + tstep, l);
        // de-rotation factors of this lane's squelch outputs (consumed when the outputs leave, at the top of the next tile)
        {
            const int ph0 = ((nz_u0 % np) + np) % np;            // block-uniform
#pragma unroll
            for (int j = 0; j < NZT; j++) {
                // (row offset c * np and ui formed once per workgroup; lanes without an output read row 0)
                int ph = ph0 + (int)((j == 0 ? nzu0 : nzu1) & 7u);
                ph = ph >= np ? ph - np : ph;
                (j == 0 ? zr0 : zr1) = *(const cf *)((const char *)p.n_krot + ((j == 0 ? nzk0 : nzk1) + (unsigned)ph) * 8u);
            }
        }

        if (PSL > 8) mark(7);
        // ---- phase A: branch filters of the channel bank and of squelch stage 1, one march of the span ----
        if (a_on) {
            const cf *z = xs + shift + DH * a_r + a_pp;
            // (pre-tiles, tile < 0: their channel rows are garbage nobody reads -- no branch inside the march)
            if (!SPLIT) {
                if (a_r == 0) march_branch<0, REAL, OPT>(z, a, an, U, a_pp);      // wave-uniform
                else march_branch<1, REAL, OPT>(z, a, an, U, a_pp);
            } else if (l < 256) {                                                // waves 0..3: the channel branches
                // (eight waves, 80 registers: a lane's taps are fetched per tile -- 16 KB of tables that stay in the L1 / L2 -- and
                // are dead outside the march, instead of 37 registers held through every phase)
                ChanTaps<REAL> at;
                cf dummy[NQ];
                const float4 *tp = (const float4 *)p.taps + a_pp * 2;
                const float4 t0 = tp[0], t1 = tp[1];
                at.re[0] = t0.x; at.re[1] = t0.y; at.re[2] = t0.z; at.re[3] = t0.w; at.re[4] = t1.x; at.re[5] = t1.y; at.re[6] = t1.z;
                if (a_r == 0) march_branch<0, REAL, OPT, true, false>(z, at, dummy, U, a_pp);
                else march_branch<1, REAL, OPT, true, false>(z, at, dummy, U, a_pp);
            } else {                                                             // waves 4..7: the squelch branches
                ChanTaps<REAL> dummy;
                cf ant[NQ];
                const float4 *np_ = (const float4 *)p.n_taps + a_pp * 8;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float4 t = np_[k];
                    ant[2 * k] = mk(t.x, t.y);
                    if (2 * k + 1 < NQ) ant[2 * k + 1] = mk(t.z, t.w);
                }
                if (a_r == 0) march_branch<0, REAL, OPT, false, true>(z, dummy, ant, U, a_pp);
                else march_branch<1, REAL, OPT, false, true>(z, dummy, ant, U, a_pp);
            }
        }
        if (PSL > 8) mark(8);
        __syncthreads();
        mark(1);
        if (SPLIT && tu + tstep < ntl && interior(tile + tstep)) load_span(tile + tstep, l);

        // ---- phase B1: DFT over p1 (p = 10 p1 + p2), twiddle e^{-j 2 pi m1 p2 / 100}, in place ----
        // (OPT 32: the two DFT passes -- LDS round trips with little arithmetic between them -- run ahead of the other
        // workgroups' waves on the SIMD, which are mostly in the arithmetic-heavy march and epilogue)
        if (OPT & 256) setprio_dyn((p.dbg >> 16) & 15);
        else if (OPT & 32) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int sw = 0; sw < NSW; sw++) {
            const int bi = l + sw * NTH;
            // (the one wave with a second sweep is what the other three wait for at the barrier: ahead of the waves of the
            // other workgroups on its SIMD while it lasts)
            if (bi < NTASK) {
                const int brow = bi / 10, bp2 = bi - 10 * brow;
                cf *col = U + brow * UST + bp2;
                const cf *twp = s_tw + bp2;
                cf x[10];
#pragma unroll
                for (int k = 0; k < 10; k++) x[k] = col[10 * k];
                dft10(x);
                col[0] = x[0];
#pragma unroll
                for (int k = 1; k < 10; k++) col[10 * k] = cmulf(x[k], twp[10 * k]);
            }
        }
        if (PSL > 8) mark(9);
        __syncthreads();
        mark(2);

        // ---- phase B2: DFT over p2.  Channel rows: bin m = m1 + 10 m2 -> Y[row][m]; squelch rows stay in place ----
#pragma unroll
        for (int sw = 0; sw < NSW; sw++) {
            const uint32_t task = sw == 0 ? (b2task & 0xffffu) : (b2task >> 16);
            if (task != 0xffffu) {
                const int row = (int)(task >> 4), m1 = (int)(task & 15u);
                cf2 *src = (cf2 *)(U + row * UST + 10 * m1);
                cf x[10];
#pragma unroll
                for (int k = 0; k < 5; k++) { const cf2 t = src[k]; x[2 * k] = t.xy; x[2 * k + 1] = t.zw; }
                dft10(x);
                if (row < NT) {
                    cf *dst = Y + row * YST + m1;
#pragma unroll
                    for (int k = 0; k < 10; k++) dst[10 * k] = x[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 5; k++) { cf2 t; t.xy = x[2 * k]; t.zw = x[2 * k + 1]; src[k] = t; }
                }
            }
        }
        if (OPT & 256) setprio_dyn((p.dbg >> 20) & 15);
        else if (OPT & 32) __builtin_amdgcn_s_setprio((OPT & 128) ? 1 : 0);
        if (PSL > 8) mark(10);
        __syncthreads();
        mark(3);

        // ---- phase C: squelch stage-1 bins (fetched now, stored after the channel epilogue) ----
#pragma unroll
        for (int j = 0; j < NZT; j++) {
            const int i = l + j * NTH < p.nsel * NU ? l + j * NTH : p.nsel * NU - 1;
            (j == 0 ? zv0 : zv1) = U[(NT + i % NU) * UST + nz_pos[j]];
        }
        prev_u0 = nz_u0;
        // channel epilogue (run sums and angle tiles go to the channel rows of U, dead since pass 2; the squelch rows
        // read above lie behind them), lane (run, c): <= RUN consecutive instants, the previous instant's bin in registers
        if (tile >= 0 && e_on) {
            const int tl0 = 1 + e_chunk * RUN;
            const cf *yc = Y + e_pos;
            // instants of this run that exist: inside the tile and inside the stream
            const long long left = p.T - (t0 + tl0);
            int nval = NT - tl0 < RUN ? NT - tl0 : RUN;
            nval = left < nval ? (int)(left < 0 ? 0 : left) : nval;
            // head sum (first tail % TT instants of the tile): only the tile that holds the end of a window's last
            // partial block is ever asked for it (block_sum_kernel)
            const int hr = (p.tail % TT > 0 && tile % p.tiles_per_block == p.tail / TT) ? p.tail % TT : 0;   // block-uniform
            float sum = 0.f, head = 0.f;
            float *drow = s_d + (tl0 - 1) * 80 + e_c;
            float *dcolp = s_dc + e_c * TT + (tl0 - 1);
            // one instant per step, the bins read two steps ahead; fenced like the march (the unrolled run is
            // independent work the scheduler would otherwise overlap at the price of a hundred registers)
            cf yb = yc[(tl0 - 1) * YST], ya = yc[tl0 * YST];
            if (OPT & 8) {
                // lean form (rho = +-1, checked by the host): 29 vector instructions per instant instead of 41.
                //  * scalar multiply-adds with the conjugation in their sign modifiers (the packed form pays a swap
                //    and a sign flip per instant to build (y.im, -y.re));
                //  * the first product is an FMA onto +0, so an exact zero comes out as +0 without the two
                //    canonicalising additions demod_poly starts with ((0, 0) must give the angle 0);
                //  * the head sum (one tile per block asks for it) is formed behind the loop, from the bin rows;
                //  * reads one row ahead without a clamp: rows past the tile's last instant are the (finite or not,
                //    never used) rest of this LDS allocation.
                float c5 = kc.c[5];
                BTGPU_OPAQUE(c5);                        // (a vector register for good: two scalar operands need a move per use)
                const float rho1 = e_rho.x;
                auto instant = [&](const cf &yp, const cf &yc_, int k) {
                    const float bx = yp.x * rho1, by = yp.y * rho1;
                    sum = fmaf(yc_.x, yc_.x, sum);
                    sum = fmaf(yc_.y, yc_.y, sum);
                    float pr = fmaf(bx, yc_.x, 0.0f), pi = fmaf(bx, yc_.y, 0.0f);            // Y[t] conj(Y[t-1]) rho
                    pr = fmaf(by, yc_.y, pr);
                    pi = fmaf(-by, yc_.x, pi);
                    const float ang = demod_poly_pz(kc, c5, pr, pi);
                    drow[k * 80] = ang;
                    dcolp[k] = ang;
                };
                constexpr int LAST = TT - (CH - 1) * RUN;        // instants of the last run
                if ((OPT & 16) && LAST % 2 == 1 && RUN % 2 == 1 && t0 + NT <= p.T) {              // block-uniform: every instant of the tile is inside the stream
                    // no per-instant predicate, two instants at a time in lockstep (demod_poly_pz2): one instant is a chain
                    // of ~25 dependent instructions (the arctangent polynomial), and a wave that is alone on its SIMD between
                    // two barriers waits out every link of it
                    auto pair = [&](const cf &y0, const cf &y1, const cf &y2, int k) {     // instants k (y0 -> y1) and k + 1 (y1 -> y2)
                        const float bx0 = y0.x * rho1, by0 = y0.y * rho1, bx1 = y1.x * rho1, by1 = y1.y * rho1;
                        sum = fmaf(y1.x, y1.x, sum);
                        sum = fmaf(y1.y, y1.y, sum);
                        sum = fmaf(y2.x, y2.x, sum);
                        sum = fmaf(y2.y, y2.y, sum);
                        float pr0 = fmaf(bx0, y1.x, 0.0f), pi0 = fmaf(bx0, y1.y, 0.0f), pr1 = fmaf(bx1, y2.x, 0.0f), pi1 = fmaf(bx1, y2.y, 0.0f);
                        pr0 = fmaf(by0, y1.y, pr0); pr1 = fmaf(by1, y2.y, pr1);
                        pi0 = fmaf(-by0, y1.x, pi0); pi1 = fmaf(-by1, y2.x, pi1);
                        float a0, a1;
                        demod_poly_pz2(kc, c5, pr0, pi0, pr1, pi1, a0, a1);
                        drow[k * 80] = a0; dcolp[k] = a0;
                        drow[(k + 1) * 80] = a1; dcolp[k + 1] = a1;
                    };
                    // (LAST and RUN odd: pairs (0,1) .. (LAST-3, LAST-2); then LAST-1 alone, or (LAST-1, LAST) .. and RUN-1 alone)
#pragma unroll
                    for (int k = 0; k + 1 < LAST; k += 2) {
                        const cf y1 = yc[(tl0 + k + 1) * YST], y2 = yc[(tl0 + k + 2) * YST];
                        pair(yb, ya, y1, k);
                        yb = y1; ya = y2;                        // rows tl0 + k + 1, tl0 + k + 2: previous and current of instant k + 2
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (e_chunk < CH - 1) {
#pragma unroll
                        for (int k = LAST - 1; k + 1 < RUN; k += 2) {
                            const cf y1 = yc[(tl0 + k + 1) * YST], y2 = yc[(tl0 + k + 2) * YST];
                            pair(yb, ya, y1, k);
                            yb = y1; ya = y2;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        instant(yb, ya, RUN - 1);
                    } else instant(yb, ya, LAST - 1);
                } else {
#pragma unroll
                    for (int k = 0; k < RUN; k++) {
                        const cf yn = yc[(tl0 + k + 1) * YST];
                        if (k < nval) instant(yb, ya, k);
                        yb = ya; ya = yn;
                        if (!(OPT & 1)) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (hr > 0) {                                    // block-uniform, one tile in tiles_per_block
#pragma unroll 1
                    for (int k = 0; k < nval && tl0 + k - 1 < hr; k++) {
                        const cf y = yc[(tl0 + k) * YST];
                        head = fmaf(y.x, y.x, head);
                        head = fmaf(y.y, y.y, head);
                    }
                }
            } else {
#pragma unroll
            for (int k = 0; k < RUN; k++) {
                const int rn = tl0 + k + 1 < NT ? tl0 + k + 1 : NT - 1;
                const cf yn = yc[rn * YST];
                if (k < nval) {
                    const cf ybr = p.rho_real ? yb * e_rho.xx : cmulf(yb, mk(e_rho.x, -e_rho.y));   // conj(Y[t-1]) rho, conjugated
                    const float m = ya.x * ya.x + ya.y * ya.y;
                    sum += m;
                    if (tl0 + k - 1 < hr) head += m;
                    const cf pq = ybr.xx * ya + ybr.yy * mk(ya.y, -ya.x);                       // Y[t] conj(Y[t-1]) rho
                    const float ang = demod_poly(kc, pq.x, pq.y);
                    drow[k * 80] = ang;
                    dcolp[k] = ang;
                }
                yb = ya; ya = yn;
                if (!(OPT & 1)) __builtin_amdgcn_sched_barrier(0);
            }
            }
            s_part[(e_chunk * 80 + e_c) * 2 + 0] = sum;
            s_part[(e_chunk * 80 + e_c) * 2 + 1] = head;
            if (p.Z) {                                           // BTGPU_FLAG_DEBUG_Y: the de-rotated channel output
                const int period = p.rot_period;
                int ph = (int)((t0 + tl0) % period);
                for (int k = 0; k < nval; k++) {
                    const cf kr = ((const cf *)p.krot)[e_c * period + ph];
                    ((cf *)p.Z)[(size_t)e_c * p.zstride + (t0 + tl0 + k)] = cmulf(yc[(tl0 + k) * YST], kr);
                    ph = ph + 1 == period ? 0 : ph + 1;
                }
            }
        }
        if (PSL > 8) mark(11);
        __syncthreads();
        if (!(OPT & 256) && (OPT & 128)) __builtin_amdgcn_s_setprio(0);
        mark(4);                                         // angle tiles complete; Y (= the span region) is dead
    }
    if (prev_any) flush_prev(prev_tile, prev_u0, l0);
}

}  // namespace btgpu
