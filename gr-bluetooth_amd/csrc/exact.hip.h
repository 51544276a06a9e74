// exact.hip.h -- the reference's per-channel DDC + quadrature demod on the fp32 MATRIX pipe (gfx950, wave = 64 lanes).
//
// What it replaces: freq_xlating_fir_filter_ccf [EXT] as multi_block::channel_samples calls it (lib/multi_block.cc:180-205)
// followed by multi_block::demod (:158-168), for the (channel, time tile) pairs a bitmap names; the rows are written over the
// polyphase banks' demodulated stream in place, so that every consumer behind (window_kernel, finish_kernel) reads the
// reference's own arithmetic wherever a channel is busy.
//
// The decimating FIR as a dense contraction.  With the reversed taps t[j] and D = the decimation,
//     y[n] = sum_j t[j] x[nD + j]  =  sum_q G[q][n + q],      G[q][m] = sum_{r < D} t[qD + r] x[mD + r]
// -- a GEMM  G = T [2 QB x 2 D] * X [2 D x columns]  over the input reshaped into columns of D samples (no copy: column m is
// x[mD .. mD + D)), followed by a QB-term diagonal sum.  Rows of T: (q, re) = (tr, -ti) interleaved along K, (q, im) = (ti, tr);
// K runs over (r, re/im) = the interleaved floats of the input as they lie in memory.
//
// SUMMATION ORDER (the bit-exact contract with oracle/bt_oracle.c ddc_run and ddc_direct_kernel):
//     G[q] = fmaf chain over r ascending, from +0:  re: fmaf(tr, xr, .) then fmaf(-ti, xi, .);  im: fmaf(ti, xr, .) then fmaf(tr, xi, .)
//     y    = ((G[0] + G[1]) + G[2]) + ... ascending q
// v_mfma_f32_32x32x2_f32 is bit for bit that chain (k = 0 then k = 1, one rounding per product, no wider accumulator:
// scripts/ubench/exact_mfma.hip checks it on the device against per-lane fmaf).  VOLK's order in the reference is unspecified
// and GNU Radio is not in the image (parity of the float half is unpinned upstream): the order is this repository's to fix.
//
// One workgroup = one time tile of 128 columns = 115 outputs (the first is the demodulator's halo) + the diagonal sum's 13: four waves, wave w owns
// columns [32 w, 32 w + 32) -- its 32 D input samples go straight from the stream into D registers per lane and stay there as the B
// operand for EVERY channel the bitmap names for the tile; the channel's T (2 QB x 2 D floats, lane-major, 12.8 KB at D = 50) comes
// from the L1 / L2 as the A operand.  Per channel and wave: D MFMAs (64 cycles each) and nothing else on the critical path.
#pragma once
#include "kernels.hip.h"

namespace btgpu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kExWaves = 4, kExThreads = 64 * kExWaves;         // one wave per SIMD: a five-wave form ran at 45 % of the matrix pipe's rate -- the SIMD that
                                                                 // held two waves of every workgroup set the pace, whatever the occupancy (profiles/r06_e_*)
constexpr int kExCols = 32 * kExWaves;          // polyphase columns per tile
constexpr int kExQB = 14;                       // tap blocks: ceil(ntaps / D) (firdes: ntaps = 44 fs / (22 * 300 kHz) | 1 = 13.33 D + 1 with D = fs / 2 MHz)
constexpr int kExTile = 114;                    // new demodulated rows per full tile; a SLOT (1250 rows) is ten of those and one of 110 (kernels.hip.h exact_tile_of)
constexpr int kExOuts = kExTile + 1;            // outputs per tile; output 0 is the halo of the demodulator
static_assert(kExOuts + kExQB - 1 <= kExCols, "the tile's outputs and the diagonal sum's overhang fit the columns");   // 128 of 128
static_assert(kExOuts <= 2 * 63 + 1, "two waves hold a tile's outputs");
static_assert(kExTile == kExTileRows, "kernels.hip.h marks the bitmap in tiles of kExTileRows rows");
constexpr int kExRS = 36;                       // floats per COLUMN of G in LDS (column-major: a lane's four consecutive rows of a C/D register group leave in ONE
                                                // 16-byte write, a (q, re), (q, im) pair comes back in ONE 8-byte read at an immediate offset): 32 rows + 4, 16-byte aligned
constexpr int kExWords = kExBmWords;            // bitmap words per tile (<= 96 channels)

struct ExactParams {
    long long x_len;
    long long first0;             // x index of (grid row 0, tap 0): w0 + first_channel_sample
    long long G;                  // rows of the shared output grid
    const float *tapsA;           // [nch][64][exact_dpad(D)]: the A operand of lane l, step r -- row l & 31 = 2 q + (re: 0, im: 1), k = l >> 5 (exact_pack_taps)
    const float2 *rot; int Qr;    // de-rotation table [nch][Qr] by grid row (the windows' own rotators differ from it by an exact +-1: the demodulated rows are the same bits)
    const float *atan_tab; float gain;
    const uint32_t *bitmap;       // [ntiles][kExWords]: bit c of tile j = rows [kExTile j, kExTile (j + 1)) of channel c are recomputed
    int ntiles;
    unsigned int *stat;           // nullptr, or a counter of the (channel, tile) pairs computed
    float *d; int drow;           // time-major stream [G][drow]
    float *dcol;                  // nullptr, or the tile-blocked copy [G / 25][80][25]
    float2 *ydbg; long long ystride;   // diagnostics: nullptr, or the de-rotated y [nch][ystride]
    int nch;
    int dbg;                      // timing experiments (scripts/ubench/exact_mfma.hip): 1 no stores, 2 no demodulator
};

constexpr int exact_dpad(int D) { return (D + 3) / 4 * 4; }      // steps of a channel's A operand, padded to whole 16-byte groups
constexpr int kExGSize = kExCols * kExRS;                        // floats of one G buffer
// LDS: two G buffers + the arctangent table -- 37.9 KB: FOUR workgroups per CU.  (Neither operand goes through LDS: B comes straight
// from the stream into registers, once per tile; A from the L1 / L2, where the 13 KB of a channel's taps stay resident.)
inline size_t exact_lds_bytes(int) { return (size_t)2 * kExGSize * sizeof(float) + 260 * sizeof(float); }
inline int exact_ntiles(long long G) { return (int)((G + kExSlotRows - 1) / kExSlotRows) * kExSlotTiles; }

// tapsA from a direct-form bank's reversed taps [nch][ntp][2] (design.h FilterBank): out[((c * (dpad / 4) + i) * 64 + l) * 4 + e] = the A
// operand of step r = 4 i + e for lane l -- row l & 31 = 2 q + (re: 0, im: 1), k = l >> 5: (tr, -ti) for a re row, (ti, tr) for an im
// row; zero beyond the filter, in rows 2 QB .. 31 and in the pad.  One 16-byte load per lane and four steps, 1 KB contiguous per wave.
inline void exact_pack_taps(const float *taps, int nch, int ntp, int D, float *out /* [nch][dpad / 4][64][4] */)
{
    const int DP = exact_dpad(D);
    for (int c = 0; c < nch; c++)
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < DP; r++) {
                const int row = l & 31, kh = l >> 5, q = row >> 1, im = row & 1, j = q * D + r;
                float tr = 0.f, ti = 0.f;
                if (r < D && q < kExQB && j < ntp) { tr = taps[((size_t)c * ntp + j) * 2]; ti = taps[((size_t)c * ntp + j) * 2 + 1]; }
                out[(((size_t)c * (DP / 4) + r / 4) * 64 + l) * 4 + (r & 3)] = im ? (kh ? tr : ti) : (kh ? -ti : tr);
            }
}
inline size_t exact_taps_floats(int nch, int D) { return (size_t)nch * 64 * exact_dpad(D); }

// Choreography of one tile: every wave fetches its 32 columns straight into D registers per lane (the B operand for every channel
// of the tile; strided 4-byte loads, once per tile), then per channel ONE barrier:
//     D MFMAs, the A operand streaming in from the L1 / L2 three 16-byte loads ahead  ->  G to Gs[n & 1]  ->  barrier
//     ->  epilogue of channel n: diagonal sums, de-rotation, demodulator, stores  (while other waves are already in channel n + 1's MFMAs)
// F12 (round 6, scripts/ubench/mfma_overlap.hip, mfma_shadow.hip): v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate because it runs ON
// the SIMD's vector lanes: nothing else of that SIMD -- of the same wave or of another -- issues in its shadow, and a kernel's time
// is its matrix time PLUS every other instruction's issue time (workgroups with matrix waves and epilogue waves side by side, one to
// four workgroups per CU, operands through LDS or from the L1: all within 10 % of each other).  What is left is to issue FEW other
// instructions: per channel the epilogue runs on TWO waves with every lane busy (waves 0, 1 for even channels of the tile, 2, 3 for
// odd ones: lanes 0 .. 63 of the first take outputs -1 .. 62, of the second 62 .. 125; y[t - 1] is one shuffle away), its row
// indices are formed once per tile.
#ifndef BTGPU_EX_FULLA
#define BTGPU_EX_FULLA 1
#endif
// D = 50: the A operand as a rotating register copy -- a 16-byte group is consumed by its four matrix instructions and at once re-fetched
// for the tile's NEXT channel, every tap load a whole channel ahead of its use: 158 VGPRs, three workgroups per CU.  Against three groups
// ahead in 122 VGPRs and four workgroups (-DBTGPU_EX_FULLA=0, and what D <= 25 keeps): the kernel alone 2.18 -> 2.11 ms at nine channels
// per tile, 3.89 -> 3.72 at twenty, 1.21 -> 1.20 at four; in the bench's step 2.14 -> 2.04 ms, records identical
// (profiles/r06_zz_ubench_exact_full_a_ab.txt, r06_zz_exact_full_a_ab_bench.txt).
constexpr bool exact_full_a(int D) { return BTGPU_EX_FULLA && D > 25; }
template <int D>
__global__ __launch_bounds__(kExThreads, exact_full_a(D) ? 3 : 4) void exact_rows_kernel(ExactParams p, const float2 *__restrict__ x)
{
    HIP_DYNAMIC_SHARED(float2, lds)
    constexpr int NS = 32 * D;                                     // samples per wave
    constexpr int NI = exact_dpad(D) / 4;                          // 16-byte groups of a channel's A per lane
    float *Gs = (float *)lds;                                      // [2][kExGSize]
    float *atab = Gs + 2 * kExGSize;
    const int tid = (int)threadIdx.x, lane = tid & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    for (int i = tid; i < 257; i += kExThreads) atab[i] = p.atan_tab[i];
    // the epilogue's lane: output u of the tile (u = 0: the halo)
    const int u = (wave & 1) ? 62 + lane : lane - 1;
    const bool u_ok = u >= 0 && u < kExOuts;
    for (int tile = (int)blockIdx.x; tile < p.ntiles; tile += (int)gridDim.x) {
        uint32_t bm[kExWords];
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < kExWords; i++) { bm[i] = p.bitmap[(size_t)tile * kExWords + i]; any |= bm[i]; }
        if (!any) continue;                                        // uniform
        if (p.stat && tid == 0) { unsigned int n = 0; for (int i = 0; i < kExWords; i++) n += (unsigned int)__popc(bm[i]); atomicAdd(p.stat, n); }
        auto next_channel = [&]() {                                // uniform: the lowest channel left in bm, -1 when none
            int c = -1;
#pragma unroll
            for (int wi = kExWords - 1; wi >= 0; wi--) if (bm[wi]) c = 32 * wi + __ffs(bm[wi]) - 1;
#pragma unroll
            for (int wi = 0; wi < kExWords; wi++) if (c >= 0 && (c >> 5) == wi) bm[wi] &= bm[wi] - 1;   // (static indices: a run-time one put bm in scratch memory)
            return c;
        };
        const long long g0 = exact_tile_row0(tile) - 1;            // grid row of output 0
        const int nout = (int)(((tile + 1) % kExSlotTiles == 0 ? (long long)(tile / kExSlotTiles + 1) * kExSlotRows : exact_tile_row0(tile + 1)) - g0);   // outputs of this tile incl. the halo (115, or 111 for a slot's last)
        const long long sb = p.first0 + g0 * D + (long long)wave * NS;
        // ---- B: lane (column m = lane & 31, half kh = lane >> 5) holds the re (kh = 0) or im (kh = 1) parts of its column's D samples ----
        float B[D];
        {
            const long long s0 = sb + (long long)D * (lane & 31);
            if (sb >= 0 && sb + NS <= p.x_len) {                   // uniform: the wave's span lies inside the stream
                // (16-byte loads -- two samples, the lane keeps its half -- cost the texture unit a quarter of this form's work per tile and
                // measured the same: 2.09 ms against 2.07, profiles/r06_u_*)
                const float *xf = (const float *)(x + s0) + (lane >> 5);
#pragma unroll
                for (int r = 0; r < D; r++) B[r] = xf[2 * r];
            } else {
#pragma unroll
                for (int r = 0; r < D; r++) {
                    const long long a = s0 + r;
                    const float2 t = x[a < 0 ? 0 : (a < p.x_len ? a : p.x_len - 1)];
                    B[r] = (a >= 0 && a < p.x_len) ? ((lane >> 5) ? t.y : t.x) : 0.f;
                }
            }
        }
        const long long g = g0 + (u_ok ? u : 0);
        // this lane's row in the two streams and its phase in the rotator's period, once per tile (a 64-bit division per channel was
        // a third of the epilogue's instructions)
        const bool row_ok = u_ok && (wave & 1 ? lane >= 1 : lane >= 2) && u < nout && g >= 1 && g < p.G;      // (lane 0 of a wave, and u = 0, only feed their neighbour)
        const unsigned int gq = (unsigned int)(g > 0 ? g : 0);
        const unsigned int rot_i = gq % (unsigned int)p.Qr;
        float *drow_p = p.d + (size_t)gq * p.drow;
        float *dcol_p = p.dcol ? p.dcol + (size_t)(gq + 25u * 79u * (gq / 25u)) : nullptr;
        int c_cur = next_channel();
        // everything of a channel behind its matrix instructions: G to LDS, the barrier, the epilogue on this channel's two waves
        auto finish_channel = [&](const f32x16 &acc, int n, int c, float2 rt) {
            const bool my_turn = ((wave >> 1) & 1) == (n & 1);       // uniform: this wave is one of the channel's two epilogue waves
            float *Gn = Gs + (n & 1) * kExGSize;
            {
                // register group j of the C/D layout = rows 8 j + 4 (lane >> 5) + (0 .. 3) of column lane & 31: four 16-byte writes, no
                // predicate (rows 28 .. 31 are zero taps: written, never read)
                float4 *gw = (float4 *)(Gn + (32 * wave + (lane & 31)) * kExRS + 4 * (lane >> 5));
#pragma unroll
                for (int j = 0; j < 4; j++) gw[2 * j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
            }
            __syncthreads();
            // ---- epilogue of channel c ----
            if (my_turn) {
            float2 y = make_float2(0.f, 0.f);
            if (u_ok) {
                const float *gu = Gn + u * kExRS;                  // block q of output u: column u + q, rows 2 q (re), 2 q + 1 (im)
                float2 gq[kExQB];
#pragma unroll
                for (int q = 0; q < kExQB; q++) gq[q] = *(const float2 *)(gu + q * (kExRS + 2));
                float yr = gq[0].x, yi = gq[0].y;
#pragma unroll
                for (int q = 1; q < kExQB; q++) { yr = yr + gq[q].x; yi = yi + gq[q].y; }
                y.x = fmaf(-yi, rt.y, yr * rt.x);
                y.y = fmaf(yi, rt.x, yr * rt.y);
                if (p.ydbg && g >= 0 && g < p.G) p.ydbg[(size_t)c * p.ystride + g] = y;
            }
            float2 yp;                                             // y[u - 1]: the left neighbour's
            yp.x = __shfl_up(y.x, 1, 64); yp.y = __shfl_up(y.y, 1, 64);
            if (row_ok && !(p.dbg & 2)) {
                const float dv = demod_one(atab, p.gain, y, yp);
                if (!(p.dbg & 1)) {
                    drow_p[c] = dv;
                    if (dcol_p) dcol_p[25 * c] = dv;
                }
            }
            }
        };
        auto rot_of = [&](int n, int c) {
            float2 rt = make_float2(1.f, 0.f);
            if (((wave >> 1) & 1) == (n & 1) && u_ok && g >= 0) rt = p.rot[(size_t)c * p.Qr + rot_i];
            return rt;
        };
        if constexpr (exact_full_a(D)) {
            // a rotating copy of the whole A operand: group i is consumed by its four matrix instructions and at once re-fetched for the
            // NEXT channel -- every tap load is a whole channel (D matrix instructions, 64 cycles each) ahead of its use
            float4 q[NI];
            {
                const float4 *ap = (const float4 *)p.tapsA + (size_t)c_cur * NI * 64 + lane;
#pragma unroll
                for (int i = 0; i < NI; i++) q[i] = ap[i * 64];
            }
            __syncthreads();                                       // (the previous tile's last epilogue has read its G buffer)
            for (int n = 0; c_cur >= 0; n++) {
                const int c_nxt = next_channel();
                const float4 *an = (const float4 *)p.tapsA + (size_t)(c_nxt >= 0 ? c_nxt : c_cur) * NI * 64 + lane;
                const float2 rt = rot_of(n, c_cur);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const float4 a = q[i];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, B[4 * i], acc, 0, 0, 0);
                    if (4 * i + 1 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, B[4 * i + 1 < D ? 4 * i + 1 : 0], acc, 0, 0, 0);
                    if (4 * i + 2 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, B[4 * i + 2 < D ? 4 * i + 2 : 0], acc, 0, 0, 0);
                    if (4 * i + 3 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, B[4 * i + 3 < D ? 4 * i + 3 : 0], acc, 0, 0, 0);
                    q[i] = an[i * 64];                             // (after the tile's last channel: that channel's own taps again, unused)
                    __builtin_amdgcn_sched_barrier(0);             // (keep the re-fetch HERE: the scheduler sinks all of them behind the chain otherwise)
                }
                finish_channel(acc, n, c_cur, rt);
                c_cur = c_nxt;
            }
        } else {
        const float4 *ap = (const float4 *)p.tapsA + (size_t)c_cur * NI * 64 + lane;
        float4 q0 = ap[0], q1 = ap[NI > 1 ? 64 : 0], q2 = ap[NI > 2 ? 128 : 0];
        __syncthreads();                                           // (the previous tile's last epilogue has read its G buffer)
        for (int n = 0; c_cur >= 0; n++) {
            const int c_nxt = next_channel();
            const float2 rt = rot_of(n, c_cur);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const float4 a = q0;
                q0 = q1; q1 = q2;
                if (i + 3 < NI) q2 = ap[(i + 3) * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, B[4 * i], acc, 0, 0, 0);
                if (4 * i + 1 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, B[4 * i + 1 < D ? 4 * i + 1 : 0], acc, 0, 0, 0);
                if (4 * i + 2 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, B[4 * i + 2 < D ? 4 * i + 2 : 0], acc, 0, 0, 0);
                if (4 * i + 3 < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, B[4 * i + 3 < D ? 4 * i + 3 : 0], acc, 0, 0, 0);
            }
            if (c_nxt >= 0) {                                      // the next channel's first groups land under this one's epilogue
                ap = (const float4 *)p.tapsA + (size_t)c_nxt * NI * 64 + lane;
                q0 = ap[0]; q1 = ap[NI > 1 ? 64 : 0]; q2 = ap[NI > 2 ? 128 : 0];
            }
            finish_channel(acc, n, c_cur, rt);
            c_cur = c_nxt;
        }
        }
    }
}

typedef void (*ExactRowsKernel)(ExactParams, const float2 *);
// the instantiations: D = fs / 2 MHz for the rates the polyphase banks serve (4 .. 50 Msps: D = 2 .. 25; 100 Msps: D = 50)
// (Round 6 also built a vector-lane form for D <= 4 -- a lane = an output row, its 14 D samples in registers, the taps as scalar
// operands, no LDS and no barrier per channel -- bit-identical to this kernel on 164 M rows and SLOWER: 2.58 ms against 2.05 for
// 145 M rows at D = 4 (profiles/r06_t_ubench_exact.txt); removed.)
inline ExactRowsKernel exact_rows_pick(int D)
{
    switch (D) {
#define BTGPU_EX(n) case n: return exact_rows_kernel<n>;
        BTGPU_EX(2) BTGPU_EX(3) BTGPU_EX(4) BTGPU_EX(5) BTGPU_EX(6) BTGPU_EX(7) BTGPU_EX(8) BTGPU_EX(9) BTGPU_EX(10) BTGPU_EX(11) BTGPU_EX(12) BTGPU_EX(13)
        BTGPU_EX(14) BTGPU_EX(15) BTGPU_EX(16) BTGPU_EX(17) BTGPU_EX(18) BTGPU_EX(19) BTGPU_EX(20) BTGPU_EX(21) BTGPU_EX(22) BTGPU_EX(23) BTGPU_EX(24) BTGPU_EX(25)
        BTGPU_EX(50)
#undef BTGPU_EX
    }
    return nullptr;
}

}  // namespace btgpu
