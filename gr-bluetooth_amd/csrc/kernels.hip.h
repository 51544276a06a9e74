// kernels.hip.h -- CDNA4 (gfx950) kernels of the sniffer hot path.  Wave = 64 lanes.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit
// fmaf(), in the order DESIGN.md fixes, so that the DIRECT path is bit-exact against the
// CPU oracle kept under oracle/ (test infrastructure; never linked here).
//
// Data layout in HBM:
//   x      interleaved complex64 input stream segment, [n] float2
//   taps   [nch][ntp] float2, time-reversed complex band-pass taps, zero padded to 8
//   Y      [nch][ystride] float2   channel / noise DDC output on the shared output grid
//   d      [G][80] float           quadrature-demodulated stream (gain * atan2), TIME-major:
//                                  the window kernel's lanes are channels -> coalesced reads
//   P, Pt  [nch][nb]      double   |Y|^2 sums per slot-block / per block head (`tail` outs)
//   Q      [nch][S]       double   noise |Y|^2 sums per slot
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Optimiser fences that emit no instruction (what they are for: pfb100f.hip.h, window_kernel).  BTGPU_OPAQUE(x): the value of x
// is unknown to the optimiser from here on; BTGPU_AFTER(x, dep): x cannot be formed before dep exists.
#if defined(__HIP_DEVICE_COMPILE__)
#define BTGPU_AFTER(x, dep) asm volatile("" : "+v"(x) : "v"(dep))
#define BTGPU_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define BTGPU_OPAQUE(x) ((void)0)
#define BTGPU_AFTER(x, dep) ((void)(dep))                        /* host emulation of the kernels (tests/emu) */
#endif

namespace btgpu {

struct DeviceHit {            // 40 bytes, written by the window kernel
    uint32_t slot;            // batch-relative slot index
    int32_t  channel_idx;     // index into the visible channel range
    int32_t  offset;
    uint32_t lap;
    int32_t  ac_errors;
    int32_t  kind;
    double   snr;
    int32_t  nsym;            // len - sub - offset, filled by nsym_patch_kernel once len is known
    int32_t  sub;             // symbols the classic pass consumed before the LE pass (Q6), else 0
    int32_t  sym;             // index of the window's packed symbols (BTGPU_FLAG_SYMBOLS), else -1
    int32_t  pad_;
};

struct FinishRec {            // M&M state of a window that reported hits, handed to finish_kernel
    int32_t  w;               // window index k * nch + c
    uint32_t ii;
    int32_t  oo;
    float    mu, omega, last;
    int32_t  done;            // the window already ended inside the window kernel (len known)
    int32_t  pad_;
};

constexpr int kSymWords = 120;    // packed symbols kept per hit window (3840 >= ~3760 symbols)

// Exact confirmation (verify.hip.h): a window of the polyphase path that reported a classic hit, or whose channel energy
// shows a burst inside the detection span, is handed to the exact stage as one of these; its records then come from the
// direct-form (bit-exact) arithmetic over the first n_exact demodulated rows of the window.
struct VerifyTask {
    int32_t w;                // window index k * nch + c
    int32_t n_exact;          // demodulated rows [0, n_exact) of the window are exact by the time the second run reads them
    double  snr;              // the window's squelch figure (the second run goes beside the next batch, whose sums reuse P / Qn)
    int32_t emit_from;        // the first run has already emitted this window's classic records at offsets below this one (from exact rows)
    int32_t le_emit_from;     // ... and its access-address records at offsets below this one
};
constexpr int kVerRows = 1416;    // rows of a window the exact stage's clock recovery can reach in 693 symbols (693 * 2.01 + 8), rounded up
constexpr int kVerTile = 127;     // new demodulated rows per tile of verify_ddc_kernel (128 outputs, the first is the demod halo)

// ------------------------------------------------------------------------------------
// K1: direct-form decimating complex band-pass FIR bank (channel bank and noise bank).
//   y[c][g] = ( sum_j taps[c][j] * x[first + g*D + j] ) * rot[c][g]
// Summation order (bit-exact contract with the oracle's ddc_run and with exact.hip.h, which runs the same order on the matrix
// pipe): the taps in BLOCKS of D (the hop), each block an fmaf chain from +0 over r ascending -- re: fmaf(tr, xr, .) then
// fmaf(-ti, xi, .); im: fmaf(ti, xr, .) then fmaf(tr, xi, .) -- and the block sums added in ascending order; ntp is a whole
// number of blocks (zero taps behind the filter).
// One lane = one output instant, CPB channels per workgroup; the input span of the
// workgroup's tile is staged through LDS once per tap chunk and shared by the CPB
// channels; taps are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------
template <int CPB>
__global__ __launch_bounds__(256) void ddc_direct_kernel(
    const float2 *__restrict__ x, long long x_len, long long first, int D, int ntp, int JC,
    const float2 *__restrict__ taps, const float2 *__restrict__ rot, int Q,
    const double *__restrict__ rot_step_turns,   // used when Q == 0
    float2 *__restrict__ Y, long long G, long long ystride, int nch,
    int seg_len, long long seg_stride)           // seg_len > 0: segmented outputs (one segment per window)
{
    // Segmented addressing (sample rates where consecutive windows sit on different decimation phases,
    // 625 * sps not a multiple of D): output n = seg * seg_len + i reads x[first + seg * seg_stride + i * D + j]
    // and is de-rotated with phase index i -- every window restarts its rotator like the reference.
    HIP_DYNAMIC_SHARED(float2, tile)
    const int T = blockDim.x;
    const int tiles_per_seg = seg_len > 0 ? (seg_len + T - 1) / T : 0;
    const long long seg = seg_len > 0 ? (long long)(blockIdx.x / tiles_per_seg) : 0;
    const long long g0 = seg_len > 0 ? (long long)(blockIdx.x % tiles_per_seg) * T : (long long)blockIdx.x * T;
    first += seg * seg_stride;
    const int c0 = blockIdx.y * CPB;
    const int o = threadIdx.x;

    float gr[CPB], gi[CPB], yr[CPB], yi[CPB];      // the running block and the sum of the finished ones
#pragma unroll
    for (int cc = 0; cc < CPB; cc++) { gr[cc] = 0.f; gi[cc] = 0.f; yr[cc] = 0.f; yi[cc] = 0.f; }

    const float2 *tp[CPB];
#pragma unroll
    for (int cc = 0; cc < CPB; cc++) {
        int c = c0 + cc < nch ? c0 + cc : nch - 1;
        tp[cc] = taps + (size_t)c * ntp;
    }

    int r = 0;                                     // position inside the block (uniform)
    bool first_blk = true;
    for (int j0 = 0; j0 < ntp; j0 += JC) {
        const int jc = (ntp - j0) < JC ? (ntp - j0) : JC;
        const int need = (T - 1) * D + jc;
        const long long base = first + g0 * D + j0;
        __syncthreads();
        // unconditional (clamped) loads, four per lane in flight: a load under a lane-dependent
        // branch is waited for at the end of that branch, one memory round trip per element
        for (int s0 = o; s0 < need; s0 += 4 * T) {
            float2 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long a = base + s0 + k * T;
                const long long ac = a < 0 ? 0 : (a < x_len ? a : x_len - 1);
                v[k] = x[ac];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = s0 + k * T;
                const long long a = base + s;
                if (s < need) tile[s] = (a >= 0 && a < x_len) ? v[k] : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        const float2 *px = tile + o * D;
        for (int j = 0; j < jc; j++) {
            const float2 v = px[j];
#pragma unroll
            for (int cc = 0; cc < CPB; cc++) {
                const float2 t = tp[cc][j0 + j];
                gr[cc] = fmaf(t.x, v.x, gr[cc]);
                gr[cc] = fmaf(-t.y, v.y, gr[cc]);
                gi[cc] = fmaf(t.y, v.x, gi[cc]);
                gi[cc] = fmaf(t.x, v.y, gi[cc]);
            }
            if (++r == D) {                        // uniform: the block is complete
#pragma unroll
                for (int cc = 0; cc < CPB; cc++) {
                    yr[cc] = first_blk ? gr[cc] : yr[cc] + gr[cc];
                    yi[cc] = first_blk ? gi[cc] : yi[cc] + gi[cc];
                    gr[cc] = 0.f; gi[cc] = 0.f;
                }
                r = 0; first_blk = false;
            }
        }
    }

    const long long g = g0 + o;                      // index inside the segment (or on the shared grid)
    if (seg_len > 0 ? g >= seg_len : g >= G) return;
    const long long gout = seg_len > 0 ? seg * seg_len + g : g;
#pragma unroll
    for (int cc = 0; cc < CPB; cc++) {
        const int c = c0 + cc;
        if (c >= nch) break;
        float rr, ri;
        if (Q > 0) {
            const float2 rt = rot[(size_t)c * Q + (int)(g % Q)];
            rr = rt.x; ri = rt.y;
        } else {
            double t = rot_step_turns[c] * (double)g;
            t -= floor(t);
            double s, co;
            sincospi(2.0 * t, &s, &co);
            rr = (float)co; ri = (float)s;
        }
        float2 out;
        out.x = fmaf(-yi[cc], ri, yr[cc] * rr);
        out.y = fmaf(yi[cc], rr, yr[cc] * ri);
        Y[(size_t)c * ystride + gout] = out;
    }
}

// ------------------------------------------------------------------------------------
// gr::fast_atan2f [EXT GNU Radio 3.7], table in LDS/global (257 floats)
// ------------------------------------------------------------------------------------
// Straight-line form (no divergent branch: a wave whose lanes fall on both sides of |y| < |x| ran BOTH divisions, a third of the
// exact rows' epilogue): the same operations on the same operands as the reference's branches, so the same bits --
//   z = min / max by the reference's own comparison;  base = z or the table's interpolation;
//   |x| > |y|:  x >= 0: +-base,  else +-(pi - base);   otherwise: +-(pi/2 -+ base)       (the outer sign is y's; -(a - b) == b - a exactly)
__device__ __forceinline__ float fast_atan2f_dev(const float *__restrict__ tab, float y, float x)
{
    const float TAN_MAP_RES = 0.003921569f;
    const float TAN_MAP_SIZE = 255.0f;
    const float y_abs = fabsf(y), x_abs = fabsf(x);
    const bool lt = y_abs < x_abs;
    const float z = __fdiv_rn(lt ? y_abs : x_abs, lt ? x_abs : y_abs);       // (0 / 0 only where the result below is discarded)
    float alpha = z * TAN_MAP_SIZE;
    const int index = ((int)alpha) & 0xff;
    alpha -= (float)index;
    const float t0 = tab[index], t1 = tab[index + 1];
    const float base_angle = z < TAN_MAP_RES ? z : t0 + ((t1 - t0) * alpha);
    const bool xg = x_abs > y_abs, xp = x >= 0.0f;
    const float off = xg ? (xp ? 0.0f : 3.14159265358979323846f) : 1.57079632679489661923f;
    const float sb = (xg == xp) ? base_angle : -base_angle;                  // + base: (|x| > |y|, x >= 0) and (|x| <= |y|, x < 0)
    float angle = (xg && xp) ? base_angle : off + sb;
    angle = (y >= 0.0f) ? angle : -angle;
    return ((y_abs > 0.0f) || (x_abs > 0.0f)) ? angle : 0.0f;
}

__device__ __forceinline__ float demod_one(const float *__restrict__ atab, float gain, float2 a, float2 b)
{
    // a * conj(b)  (multi_block.cc:165-166)
    float pr = fmaf(a.y, b.y, a.x * b.x);
    float pi = fmaf(a.y, b.x, -(a.x * b.y));
    return gain * fast_atan2f_dev(atab, pi, pr);
}

// ------------------------------------------------------------------------------------
// K2: |Y|^2 sums per slot-block (and over the first `tail` outputs of the block).  One workgroup
// per (block b of `bs` outputs, channel).  Float mag^2 accumulated in double like
// multi_block.cc:206-218; fixed reduction order.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void energy_kernel(
    const float2 *__restrict__ Y, long long G, long long ystride, int bs, int tail,
    double *__restrict__ P, double *__restrict__ Pt, int nb, int nch, int nsum)
{
    // nsum: outputs of a block that count towards P (the channel bank sums whole blocks, nsum = bs; the
    // noise bank only the noise_out = 850 outputs of each slot the reference averages, multi_block.cc:269-286)
    __shared__ double red[2][4];
    const int c = blockIdx.y;
    const int b = blockIdx.x;
    const float2 *y = Y + (size_t)c * ystride;
    const long long gb = (long long)b * bs;
    double s_full = 0.0, s_tail = 0.0;
    for (int i = threadIdx.x; i < nsum; i += blockDim.x) {
        const long long g = gb + i;
        if (g >= G) break;
        const float2 v = y[g];
        const float m = (v.x * v.x) + (v.y * v.y);
        s_full += (double)m;
        if (i < tail) s_tail += (double)m;
    }
    // wave reduce (64 lanes) then across the 4 waves
    for (int off = 32; off > 0; off >>= 1) {
        s_full += __shfl_down(s_full, off, 64);
        s_tail += __shfl_down(s_tail, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s_full; red[1][wave] = s_tail; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        double a = 0.0, t = 0.0;
        for (int w = 0; w < nw; w++) { a += red[0][w]; t += red[1][w]; }
        P[(size_t)c * nb + b] = a;
        if (Pt) Pt[(size_t)c * nb + b] = t;
    }
}

// ------------------------------------------------------------------------------------
// K2b: quadrature demodulation of the channel streams Y[c][g] (multi_block::demod,
// lib/multi_block.cc:158-173) into the time-major stream d[g][drow] the window and finish kernels read.
// One workgroup = 64 consecutive grid points
// of every channel: wave w takes channels w, w + 4, ... with lane = grid point (coalesced reads
// along g), the values cross an LDS tile and leave as whole rows (a lane-per-grid-point store
// into d would touch 64 different rows per instruction).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void demod_rows_kernel(
    const float2 *__restrict__ Y, long long G, long long ystride, int nch,
    const float *__restrict__ atan_tab, float gain, float *__restrict__ d, int drow)
{
    __shared__ float atab[257];
    __shared__ float tile[64 * 81];
    for (int i = threadIdx.x; i < 257; i += blockDim.x) atab[i] = atan_tab[i];
    __syncthreads();
    const long long g0 = (long long)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long g = g0 + lane;
    const long long gc = g < G ? g : G - 1;
    for (int c = w; c < nch; c += 4) {
        const float2 *y = Y + (size_t)c * ystride;
        const float2 v = y[gc], vp = y[gc > 0 ? gc - 1 : 0];
        const float dv = (g > 0 && g < G) ? demod_one(atab, gain, v, vp) : 0.f;   // policy Q1: d[0] = 0
        tile[lane * 81 + c] = dv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * nch; i += blockDim.x) {
        const int r = i / nch, cc = i - r * nch;
        if (g0 + r < G) d[(size_t)(g0 + r) * drow + cc] = tile[r * 81 + cc];
    }
}

// ------------------------------------------------------------------------------------
// K3: per (slot, channel) window: squelch decision, M&M clock recovery, slicer and the
// access-code search, fused.  One lane per window (windows are independent under the
// windowed-reset policy).  Symbols are never stored: the 68-symbol correlator window is
// a shift register updated as each symbol is sliced, and offset c = s - 67 is tested the
// moment symbol s exists (classic_packet::sniff_ac, lib/packet_impl.cc:247-268).  A
// candidate is committed when one more symbol exists (c < len - 68) and c < 625
// (lib/multi_sniffer_impl.cc:108-127); the next search resumes at c + 68.
// ------------------------------------------------------------------------------------
struct WindowParams {
    int nch, S;
    int qn_stride;              // slots per channel row of Qn (= S of the batch; the exact stage's launch has its own S)
    int outs_per_slot;          // grid points per slot (slot / decim)
    int ddc_out, noise_out;
    int blocks_per_window, tail;
    int nb;                     // number of energy blocks per channel
    long long ystride;
    double target_snr;
    float gain_mu, mu0, omega_relative_limit, omega0, gain_omega, omega_mid;
    int mode;                   // BTGPU_MODE_*
    int max_hits;
    uint64_t a0_lo; uint32_t a0_hi;
    int le;                     // run the le_packet::sniff_aa pass (multi_sniffer, BTGPU_FLAG_LE)
    int syms;                   // export the packed symbols of hit windows (BTGPU_FLAG_SYMBOLS)
    int low_channel;
    int btbb;                   // multi_LAP: libbtbb-style search (BTGPU_CORRELATOR_BTBB)
    const uint64_t *btbb_pcol;  // [24] parity column of each LAP bit (device memory)
    int fin_prio;               // wave priority of finish_kernel (0..3)
    int want_len;               // 0 (BTGPU_FLAG_NO_NSYM): windows that outlast the detection span are not continued, nsym = -1
    // deferred squelch: the window kernel runs BESIDE squelch stage 2 instead of behind it -- every window is taken through clock
    // recovery and search (at the reference's default threshold they all pass anyway), the SNR decision is applied
    // where records leave: the exact stage's run skips a task that fails it, nsym_patch_kernel drops the other records
    int deferred;
    const double *snr_arr;      // [S * nch] 10 log10(E_on / E_off) per window, from squelch_kernel (deferred mode)
    // exact confirmation of the polyphase path's records (verify.hip.h)
    int verify;                 // 1, 2: a classic hit whose rows are not exact yet (exact.hip.h has not covered them: bm1) is not emitted; the window
                                //    is listed for the second run, window_kernel<LAY, true>, behind a second launch of exact.hip.h (bm2).  1: presence_kernel has marked bm1
    int rows_per_slot;          // stream rows from one slot's first row to the next slot's (outs_per_slot; second run: kVerRows)
    const double *ptile;        // [nch][ptile_stride] |Y|^2 sums per tile of tile_outs outputs (the polyphase banks' by-product)
    int ptile_stride, tile_outs, tiles_per_slot;
    VerifyTask *vtasks; unsigned int *vcount;   // vcount: 0 tasks reserved, 1 (channel, tile) pairs marked, 2 windows turned away (list full), 3 busy windows (presence)
    int vcap;                   // capacity of vtasks
    uint32_t *bm1, *bm2;        // exact rows' bitmaps [bm_tiles][kExBmWords]: marked by presence_kernel / by the first run's uncovered hits
    int bm_tiles;
    float burst_abs;            // presence: a W-tile sum above burst_abs * (the quietest aligned W-tile block of the span) ...
    int burst_w;                // ... W = tiles per ~50 us (set_verify_flagging, bank_launch.h)
    float burst_abs1;           // ... or above burst_abs1 * (the quietest single tile), whichever is lower
    float burst_abs2;           // ... or above burst_abs2 * chan_floor[channel]: the channel's quietest tile of the whole batch (channel_floor_kernel)
    const float *chan_floor;    // [81], nullptr: not computed
    float burst_abs_hot, burst_hot;   // as multiples of that threshold: the threshold beside a neighbour channel whose W-tile sum exceeds burst_hot x it
    int span_extra;             // symbols behind an access code that stay exact as well (the 54-symbol header + margin)
    int dbg_stop;               // diagnostics (BTGPU_WIN_STOP): 1 = stop before phase 1, 2 = after it, 3 = after the classic search
};

__device__ __forceinline__ int popc5min(uint32_t v, uint32_t a, uint32_t b)
{
    int da = __popc(v ^ a), db = __popc(v ^ b);
    return da < db ? da : db;
}

constexpr int kWinThreads = 256;     // lane = (slot, channel), slots per workgroup * channels <= 256
constexpr int kWinRows = 38;         // demod rows staged per chunk and slot (default of a layout)
constexpr int kWinRowsSmall = 29;    // the smallest C79 tile that still holds phase 2's tables: 39.8 KB, four workgroups per CU --
                                     // measured no faster than 38 rows / three (profiles/r03_m_*): the kernel is bound by VALU issue, not occupancy
// Window-kernel layouts, by channel count.  NSL slots per workgroup, rows of NCP4 float4 (the used
// columns of the 80-float rows of d) at an LDS row stride of RS floats.
//   <3, 96, 20>  41..80 channels (C79): 3 x 79 lanes fill four waves to 93 %, one wave per SIMD (2 x 79 in
//                three waves and 4 x 79 in five both leave the SIMDs unevenly loaded: 1.4-1.6x slower);
//                RS is a multiple of 32, so lane c always reads bank c mod 32 whatever its input index
//                (with the stream's own 80 the lanes c, c + 16 collide whenever their indices differ
//                by an odd number); slot s starts 16 s floats into its bank row so that the wave
//                holding the end of one slot and the start of the next still reads 32 different banks
//   <6, 40, 10> <12, 20, 5> <32, 8, 2> <64, 4, 1>   narrower captures (20, 8, 4, 2 Msps): compact rows,
//                as many slots as fill the 256 lanes; the per-slot pad of 8 floats spreads the slots
//                over the four 8-bank groups
// row stride (floats) of the time-major stream d for a capture of nch channels = 4 * NCP4 of its layout
inline int win_drow(int nch) { return nch > 40 ? 80 : nch > 20 ? 40 : nch > 8 ? 20 : nch > 4 ? 8 : 4; }

template <int NSL, int RS, int NCP4, int ROWS = kWinRows>
struct WinLayout {
    static constexpr int kSlots = NSL, kRowStride = RS, kVecPerRow = NCP4;
    static constexpr int kRows = ROWS;                       // demod rows staged per chunk and slot
    static constexpr int kTileFloats = ROWS * RS + (RS % 32 == 0 ? 16 : 8);
};
constexpr int kMmseStride = 12;      // floats per interpolator row in LDS: 16-byte slot 3 imu mod 16 instead of
                                     // 2 imu mod 16 (eight classes for the sixteen lanes of a 16-byte read group)
constexpr int kDetectSyms = 693;     // 625 search offsets + 68-symbol access code
constexpr int kBurstAhead = 2;       // the neighbour channels' energy sums run this many tiles ahead of the channel's own
constexpr int kChanFloorAll = 80;     // channel_floor_kernel: out[80] = the quietest tile of ANY channel
constexpr int kBurstFront = 7;       // tiles in front of a window that the burst scan reads (57 tiles of the span + 7 = one wave's 64 lanes)
constexpr int kBitWords = 24;        // 32-bit words of sliced symbols kept per lane (>= 693 + 99 bits)

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | (c & (a | b)); }

constexpr int kSymbolsShortAcDev = 68;   // SYMBOLS_PER_BASIC_RATE_SHORTENED_ACCESS_CODE

// classic_packet::sniff_ac (lib/packet_impl.cc:247-268) with check_ac (:471-510) over the offsets
// [resume, limit) of one lane's packed symbols (word j of the lane at mybits[j * kWinThreads], LDS),
// 32 offsets per pass: the gate PREAMBLE_DISTANCE[5 bits] + BARKER_DISTANCE[7 bits] <= 2 -- both
// tables are "min Hamming distance to a pattern or its complement" -- is evaluated bit-sliced with
// full adders over shifted copies of the stream; survivors get the 68-bit compare against the
// affine access code AC(0) ^ cols(LAP) (three 256-entry LUTs) with popcount < 7.  A hit moves the
// search on by `step` symbols (68: the loop of lib/multi_sniffer_impl.cc:107-127; 1: every
// qualifying offset); first_only stops at the first one (multi_LAP).  Shared by window_kernel's
// phase 2 and scan_symbols_kernel, which runs it on captured symbol streams.
template <class Emit, class AcHi>
__device__ __forceinline__ void search_classic(const uint32_t *mybits, int limit, int step, bool first_only,
                                               uint64_t a0_lo, uint32_t a0_hi, const uint64_t *ac_lo,
                                               const AcHi *ac_hi, int &resume, int &nhits, Emit emit)
{
    uint32_t r0 = mybits[0], r1 = mybits[kWinThreads], r2 = mybits[2 * kWinThreads], r3;
    for (int b = 0; b * 32 < limit; b++) {
        r3 = mybits[(b + 3) * kWinThreads];
        // W_k: bit j = symbol (32 b + j + k)
        const uint32_t w0 = r0;
        const uint32_t w1 = __funnelshift_r(r0, r1, 1), w2 = __funnelshift_r(r0, r1, 2);
        const uint32_t w3 = __funnelshift_r(r0, r1, 3), w4 = __funnelshift_r(r0, r1, 4);
        const uint32_t v61 = __funnelshift_r(r1, r2, 29), v62 = __funnelshift_r(r1, r2, 30);
        const uint32_t v63 = __funnelshift_r(r1, r2, 31), v64 = r2;
        const uint32_t v65 = __funnelshift_r(r2, r3, 1), v66 = __funnelshift_r(r2, r3, 2);
        const uint32_t v67 = __funnelshift_r(r2, r3, 3);
        // preamble: distance to 0x0a (symbols 0,1,0,1,0) or its complement
        const uint32_t m0 = w0, m1 = ~w1, m2 = w2, m3 = ~w3, m4 = w4;
        const uint32_t s1 = xor3(m0, m1, m2), c1 = maj3(m0, m1, m2);
        const uint32_t s2 = m3 ^ m4, c2 = m3 & m4;
        const uint32_t a0 = s1 ^ s2, c3 = s1 & s2;
        const uint32_t a1 = xor3(c1, c2, c3), a2 = maj3(c1, c2, c3);
        const uint32_t p1 = ~a2 & a1;                                  // min(d, 5 - d), bit 1
        const uint32_t p0 = (~a2 & ~a1 & a0) | (a2 & ~a0);             //               bit 0
        // Barker + LAP msb: distance to 0x27 (symbols 1,1,1,0,0,1,0) or its complement
        const uint32_t n0 = ~v61, n1 = ~v62, n2 = ~v63, n3 = v64, n4 = v65, n5 = ~v66, n6 = v67;
        const uint32_t t1 = xor3(n0, n1, n2), u1 = maj3(n0, n1, n2);
        const uint32_t t2 = xor3(n3, n4, n5), u2 = maj3(n3, n4, n5);
        const uint32_t d0 = xor3(t1, t2, n6), u3 = maj3(t1, t2, n6);
        const uint32_t d1 = xor3(u1, u2, u3), d2 = maj3(u1, u2, u3);
        const uint32_t b0 = d0 ^ d2, b1 = d1 ^ d2;                     // min(d, 7 - d) in 0..3
        // PREAMBLE_DISTANCE + BARKER_DISTANCE <= 2
        uint32_t ok = (~p1 & ~p0 & ~(b1 & b0)) | (~p1 & p0 & ~b1) | (p1 & ~p0 & ~b1 & ~b0);
        while (ok) {
            const int j = __ffs(ok) - 1;
            ok &= ok - 1;
            const int cpos = 32 * b + j;
            if (cpos < resume || cpos >= limit) continue;
            // 68-bit window at offset cpos
            const uint32_t x0 = j ? ((r0 >> j) | (r1 << (32 - j))) : r0;
            const uint32_t x1 = j ? ((r1 >> j) | (r2 << (32 - j))) : r1;
            const uint32_t x2 = j ? ((r2 >> j) | (r3 << (32 - j))) : r2;
            const uint64_t wlo = ((uint64_t)x1 << 32) | x0;
            const uint32_t whi = x2 & 0xf;
            const uint32_t lap = (uint32_t)(wlo >> 38) & 0xffffff;
            const uint64_t elo = a0_lo ^ ac_lo[lap & 0xff] ^ ac_lo[256 + ((lap >> 8) & 0xff)] ^ ac_lo[512 + (lap >> 16)];
            const uint32_t ehi = a0_hi ^ (uint32_t)ac_hi[lap & 0xff] ^ (uint32_t)ac_hi[256 + ((lap >> 8) & 0xff)] ^ (uint32_t)ac_hi[512 + (lap >> 16)];
            const int err = __popcll(elo ^ wlo) + __popc((ehi ^ whi) & 0xf);
            if (err < 7) {
                emit(cpos, lap, err);
                nhits++;
                resume = cpos + step;
                if (first_only) limit = 0;                             // multi_LAP: first hit only
            }
        }
        r0 = r1; r1 = r2; r2 = r3;
    }
}

// One workgroup per kWinSlots consecutive slots, one lane per (slot, channel) window (nch <= 79).
//
// Phase 1 -- clock recovery.  The demodulated stream is time-major [g][80], so the rows a slot's
// windows need are shared by all its lanes: they are staged through LDS in chunks of kWinRows
// rows (16-byte copies, LDS row stride 96 so that lane c always reads bank c mod 32) and the
// strictly sequential M&M recursion of each lane (multi_block::mm_cr, lib/multi_block.cc:128-155;
// windowed reset) runs out of LDS.  A lane leaves a chunk with its input index just past the last
// usable row, so the chunk schedule is static: the rows of chunk n + 1 are fetched into registers
// (unconditional loads) before the recursion over chunk n starts.  Only the first 693 symbols can
// hold a reportable access code (lib/multi_sniffer_impl.cc:108-126), so the recursion stops there;
// sliced symbols are packed one bit each, leave through a small HBM scratch and come back into
// the dead tile for phase 2 -- the LDS footprint (51 KB) lets three workgroups of four waves
// share a CU, one wave of each per SIMD, which keeps every window of a 2048-slot batch resident.
//
// Phase 2 -- access-code search (classic_packet::sniff_ac, lib/packet_impl.cc:247-268) on the
// packed bits, 32 offsets at a time: the 5-bit preamble and 7-bit Barker distance LUTs are
// "min Hamming distance to a pattern or its complement", evaluated bit-sliced with full adders
// over shifted copies of the stream; surviving candidates (7.7 % on noise) get the 68-bit
// compare against the affine access code AC(0) ^ cols(LAP) with popcount < 7
// (check_ac, lib/packet_impl.cc:471-510).  Hits are taken greedily with resume at c + 68 and
// limit = min(len - 68, 625), which is what the reference's while (limit >= 0) loop does.
// One Mueller & Mueller update (multi_block::mm_cr, lib/multi_block.cc:128-155; clock_recovery_mm_ff [EXT]) after
// the interpolator produced `out`: the reference's float operations in its order, one rounding each.
//   mm_val = slice(last) * out - slice(out) * last      slice(x) = x < 0 ? -1 : +1
// Both products are exact (+-x), so the value is the ONE rounding of their difference: the sign of `last` is
// XORed into `out` and vice versa (two full-rate bit operations each instead of v_bfi + v_mul), then one subtract.
// Returns floor(mu) as an integer: the input samples to advance by.
__device__ __forceinline__ int mm_update(float out, float &last, float &omega, float &mu, const WindowParams &p)
{
    const uint32_t ob = __float_as_uint(out), lb = __float_as_uint(last);
    const float t1 = __uint_as_float(ob ^ (lb & 0x80000000u));      // slice(last) * out
    const float t2 = __uint_as_float(lb ^ (ob & 0x80000000u));      // slice(out) * last
    const float mm_val = t1 - t2;
    last = out;
    omega = omega + (p.gain_omega * mm_val);
    {
        const float xx = omega - p.omega_mid;                        // branchless_clip [EXT], roundings kept
        const float x1 = fabsf(xx + p.omega_relative_limit) - fabsf(xx - p.omega_relative_limit);
        omega = fmaf(0.5f, x1, p.omega_mid);                         // 0.5 * x1 is exact
    }
    mu = mu + (omega + (p.gain_mu * mm_val));
    const float fl = floorf(mu);
    mu = mu - fl;
    return (int)fl;
}

// Sliced symbols, one bit each (1 = the sample is not negative), 32 per word, symbol n in bit n & 31.  Inside
// the recursion a word is collected with ONE instruction per symbol: the sample's sign bit is shifted in from
// the right (v_alignbit), which leaves the word bit-reversed and inverted until it is flushed.  (The interpolator
// sum starts from +0 and is never -0, so "sign bit clear" is "x >= 0"; the compare + select + shift + or it
// replaces cost five instructions, one of them a v_cndmask on vcc.)
__device__ __forceinline__ uint32_t sym_push(uint32_t acc, float out)
{
    return __builtin_amdgcn_alignbit(acc, __float_as_uint(out), 31);     // (acc << 1) | sign(out)
}
// the collected word in the stored format; n = symbols in it (1..32)
__device__ __forceinline__ uint32_t sym_word(uint32_t acc, int n)
{
    return (~__brev(acc)) >> (32 - n);
}

// rows the clock recovery can reach within `span` symbols: at most omega_mid + omega_relative_limit input rows per symbol, + the
// 8-tap interpolator
__device__ __forceinline__ int ver_rows_of(const WindowParams &p, int span)
{
    int rows = (int)((float)span * (p.omega_mid + p.omega_relative_limit)) + 12;
    const int cap_rows = p.ddc_out < kVerRows ? p.ddc_out : kVerRows;
    return rows > cap_rows ? cap_rows : rows;
}

// ---- exact rows: which (channel, time tile) pairs are recomputed through the reference's own arithmetic (exact.hip.h) ----
// bitmap[tile][kExWords]: bit c of tile t.  Tiles never straddle a slot: a slot's 1250 rows (at every rate with a shared grid) are
// eleven tiles -- ten of 114 rows and one of 110 -- and a batch begins with a slot, so which rows are exact does not depend on how the
// stream is cut into batches or ranks, nor on what the first slot is called.  (114 + the diagonal sum's 13 + the demodulator's halo
// = 128 polyphase columns = four waves, one per SIMD.)
constexpr int kExTileRows = 114;     // = exact.hip.h kExTile
constexpr int kExSlotRows = 1250, kExSlotTiles = 11;
constexpr int kExBmWords = 3;        // bitmap words per tile (<= 96 channels)
__host__ __device__ __forceinline__ int exact_tile_of(long long g)         // the tile that holds grid row g
{
    const long long k = g / kExSlotRows;
    const int j = (int)(g - k * kExSlotRows) / kExTileRows;
    return (int)k * kExSlotTiles + (j < kExSlotTiles ? j : kExSlotTiles - 1);
}
__host__ __device__ __forceinline__ long long exact_tile_row0(int t)      // its first row; exact_tile_row0(t + 1) is one past its last
{
    const int k = t / kExSlotTiles, j = t - k * kExSlotTiles;
    return (long long)k * kExSlotRows + (long long)j * kExTileRows;      // (j = 11 of slot k = row 1254 > 1250: never asked -- t + 1 of a slot's last tile is j = 0 of the next)
}
__device__ __forceinline__ void exact_mark(uint32_t *bm, const uint32_t *skip, int ntiles, long long g_lo, long long g_hi, int c, unsigned int *stat)
{
    int t0 = exact_tile_of(g_lo), t1 = exact_tile_of(g_hi - 1);
    if (t1 >= ntiles) t1 = ntiles - 1;
    const uint32_t bit = 1u << (c & 31);
    int n = 0;
    for (int t = t0; t <= t1; t++) {
        const size_t i = (size_t)t * kExBmWords + (c >> 5);
        if (skip && (skip[i] & bit)) continue;
        if (!(atomicOr(&bm[i], bit) & bit)) n++;
    }
    if (stat && n) atomicAdd(stat, (unsigned int)n);
}
// rows of window (k, c), from its first row on, that the bitmap covers without a gap
__device__ __forceinline__ int exact_covered_rows(const uint32_t *bm, int ntiles, long long g_lo, int c)
{
    int t = exact_tile_of(g_lo);
    const uint32_t bit = 1u << (c & 31);
    while (t < ntiles && (bm[(size_t)t * kExBmWords + (c >> 5)] & bit)) t++;
    const long long cov = exact_tile_row0(t) - g_lo;
    return cov > 0 ? (cov > 0x3fffffff ? 0x3fffffff : (int)cov) : 0;
}

// PRESENCE (round 6).  The reference runs one arithmetic over every window and needs no edge, no gap and no
// step in level to find an access code (lib/multi_sniffer_impl.cc:87-127).  The product's default front end is a tolerance path
// (polyphase banks), so every window whose detection span holds ENERGY -- the channel's 50-us sum over 2.0 x the mean noise,
// anywhere in the span: a packet that begins there, one that began before the window and is still on the air, one that
// continues another with neither gap nor step -- gets its rows, from the window's first row to 80 symbols past the last busy
// tile, recomputed by exact.hip.h before the window kernel reads them.  Rounds 4-5 selected on RISING energy (a dozen tuned
// constants; each round's generator found the next class of packet that shows no rise): gone.  What presence does not see is a
// packet under the threshold itself (~2-3 dB over the noise; beside a neighbour channel >= 20 dB over the noise: 3.0 x, ~4.5 dB)
// -- sensitivity, the one statement left.
// Launched like the window kernel (lane = (slot, channel)); the tile sums of a slot's channels cross LDS.
template <class LAY>
__global__ __launch_bounds__(kWinThreads) void presence_kernel(WindowParams p)
{
    constexpr int kWinSlots = LAY::kSlots, kTileFloats = LAY::kTileFloats;
    __shared__ __attribute__((aligned(16))) float tile[kWinSlots * kTileFloats];
    __shared__ unsigned int s_cnt[2];
    const int nch = p.nch;
    const int sl = (int)threadIdx.x / nch, cq = (int)threadIdx.x - sl * nch, kq = blockIdx.x * kWinSlots + sl;
    const bool lane_ok = sl < kWinSlots && kq < p.S;
    if (threadIdx.x == 0) { s_cnt[0] = 0u; s_cnt[1] = 0u; }
    const int TT = p.tile_outs;
    const int ntm = (2 * kDetectSyms + 16 + TT - 1) / TT;          // tiles of the detection span
    constexpr int NF = kBurstFront;                                // tiles in front of the window that the noise estimate and the first sums read
    const int NTW = ntm + NF, PW = NTW | 1;                        // odd pitch: no bank conflicts
    int spp = (kWinSlots * kTileFloats) / (nch * PW);
    spp = spp < 1 ? 1 : (spp > kWinSlots ? kWinSlots : spp);
    float *et = tile;                                              // [spp][nch][PW]: tile t0 - NF + jj of (slot, channel) at et[(sp * nch + cc) * PW + jj]; -1 = no such tile
    for (int s0 = 0; s0 < kWinSlots; s0 += spp) {
        if (blockIdx.x * kWinSlots + s0 >= p.S) break;             // uniform
        __syncthreads();
        for (int pr0 = (int)threadIdx.x >> 6; pr0 < spp * nch; pr0 += 4 * (kWinThreads / 64)) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pr = pr0 + u * (kWinThreads / 64);
                const int prc = pr < spp * nch ? pr : pr0;
                const int sp = prc / nch, cc = prc - sp * nch;
                const int ks = blockIdx.x * kWinSlots + s0 + sp;
                const int jj = (int)threadIdx.x & 63;
                const int t = ks * p.tiles_per_slot - NF + jj;
                const bool in = s0 + sp < kWinSlots && ks < p.S && t >= 0 && t < p.ptile_stride;
                v[u] = (float)p.ptile[(size_t)cc * p.ptile_stride + (in ? t : 0)];
                v[u] = in ? v[u] : -1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pr = pr0 + u * (kWinThreads / 64);
                const int jj = (int)threadIdx.x & 63;
                if (pr < spp * nch && jj < NTW) et[pr * PW + jj] = v[u];
            }
        }
        __syncthreads();
        if (sl < s0 || sl >= s0 + spp || !lane_ok) continue;       // this lane's slot is not in this pass
        const int t0 = kq * p.tiles_per_slot;
        const float *pe = et + ((sl - s0) * nch + cq) * PW + NF;   // pe[j]: tile j of this window's span, j = -NF .. nt - 1
        int nt = ntm;
        if (nt > p.ptile_stride - t0) nt = p.ptile_stride - t0;
        const int W = p.burst_w;
        const int jl = t0 + nt == p.ptile_stride ? nt - 1 : nt;    // the batch's last tile may be a partial one
        // ---- the noise references: the span's quietest aligned W-tile block (the host folds its known shortfall under the mean into
        // burst_abs), the span's quietest single tile (1.6 x: noise alone never prefers it), the channel's quietest tile of the whole
        // batch and 1.5 x the quietest of any channel's (channel_floor_kernel) -- a span that a packet fills from end to end holds no
        // quiet tile of its own.  The LOWEST threshold decides: a false "busy" costs rows, a false "quiet" a record.
        float bmin = 3.0e38f;
        bool some = false;                                         // any tile with signal at all (GNU Radio's leading zeros are none)
        for (int jb = -NF; jb + W <= jl; jb += W) {
            float sb = 0.f;
            bool ok = true;
            for (int u = 0; u < W; u++) { const float e = pe[jb + u]; ok = ok && e > 0.f; some = some || e > 0.f; sb += e; }
            bmin = (ok && sb < bmin) ? sb : bmin;
        }
        float mn1 = 3.0e38f;
        {
            float ep = pe[-NF];
            for (int jx = -NF + 1; jx < jl; jx++) { const float e = pe[jx]; mn1 = (e > 0.f && ep > 0.f && e < mn1) ? e : mn1; ep = e; }
        }
        int last_busy = -1;
        if (bmin > 1.0e38f) {
            // no whole block of the span holds signal (a stream that begins inside the span): nothing to compare with -- exact to the end
            if (some) last_busy = nt - 1;
        } else {
            const float thr_b = p.burst_abs * bmin, thr_1 = p.burst_abs1 * mn1;
            const float cf_ = p.chan_floor ? fminf(p.chan_floor[cq], 1.5f * p.chan_floor[kChanFloorAll]) : 3.0e38f;
            const float thr_2 = cf_ < 1.0e38f ? p.burst_abs2 * cf_ : 3.0e38f;
            float thr = thr_1 < thr_b ? thr_1 : thr_b;
            thr = thr_2 < thr ? thr_2 : thr;
            const float thr_n = thr * p.burst_abs_hot, hot = thr * p.burst_hot;      // (both as multiples of the threshold)
            const bool has_l = cq > 0, has_r = cq + 1 < nch;
            float s_cur = 0.f, sl_ = 0.f, sr_ = 0.f;               // the W-tile sums of the channel and of its two neighbours (those kBurstAhead tiles ahead)
            for (int u = 0; u < W; u++) {                          // position j = -1
                const int ic = -1 - u;
                const float ec = ic >= -NF ? pe[ic] : -1.f;
                s_cur += ec > 0.f ? ec : 0.f;
                const int in_ = ic + kBurstAhead;
                const float el = (has_l && in_ >= -NF && in_ < ntm) ? pe[in_ - PW] : 0.f, er = (has_r && in_ >= -NF && in_ < ntm) ? pe[in_ + PW] : 0.f;
                sl_ += el > 0.f ? el : 0.f; sr_ += er > 0.f ? er : 0.f;
            }
            for (int jx = 0; jx < nt; jx++) {
                const int io = jx - W;
                const float en = pe[jx], eo = io >= -NF ? pe[io] : -1.f;
                s_cur += (en > 0.f ? en : 0.f) - (eo > 0.f ? eo : 0.f);
                {
                    // (two tiles AHEAD of this channel's sum: a transmitter that switches on splatters into the neighbour channels
                    // in its first microseconds, when its own W-tile sum has hardly begun to rise)
                    const int jn = jx + kBurstAhead, jo = io + kBurstAhead;
                    const float ln = (has_l && jn < ntm) ? pe[jn - PW] : 0.f, lo = (has_l && jo >= -NF && jo < ntm) ? pe[jo - PW] : 0.f;
                    const float rn = (has_r && jn < ntm) ? pe[jn + PW] : 0.f, ro = (has_r && jo >= -NF && jo < ntm) ? pe[jo + PW] : 0.f;
                    sl_ += (ln > 0.f ? ln : 0.f) - (lo > 0.f ? lo : 0.f);
                    sr_ += (rn > 0.f ? rn : 0.f) - (ro > 0.f ? ro : 0.f);
                }
                // Beside a neighbour channel that carries a packet >= 20 dB over the noise the threshold is 3 x the mean noise
                // instead of 2 x: the channel filter passes -36 .. -20 dB of that packet per tile, i.e. about the noise level
                // and up, and with the low threshold every strong packet would make busy windows of its two neighbour channels.
                // A statement about SENSITIVITY: beside such a neighbour a packet is taken from ~4.5 dB over the noise.
                const float thr_j = (sl_ > hot || sr_ > hot) ? thr_n : thr;
                if (s_cur > thr_j) last_busy = jx;
            }
        }
        if (last_busy >= 0) {
            // to 80 symbols behind the last busy tile: the end of an access code that starts in it, its header (54 + 4 symbols)
            const int rows = ver_rows_of(p, ((last_busy + 1) * TT) / 2 + 80);
            const long long g_lo = (long long)kq * p.outs_per_slot;
            exact_mark(p.bm1, nullptr, p.bm_tiles, g_lo, g_lo + rows, cq, &s_cnt[1]);
            atomicAdd(&s_cnt[0], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[0]) { atomicAdd(&p.vcount[3], s_cnt[0]); atomicAdd(&p.vcount[1], s_cnt[1]); }   // (statistics: busy windows, pairs marked)
}

// BTGPU_FLAG_EXACT_ALL: no selection -- every (channel, tile) pair is marked
__global__ void exact_mark_all_kernel(uint32_t *__restrict__ bm, int ntiles, int nch)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= ntiles * kExBmWords) return;
    const int w = i % kExBmWords, lo = 32 * w;
    bm[i] = nch >= lo + 32 ? 0xffffffffu : (nch > lo ? (1u << (nch - lo)) - 1u : 0u);
}

// The quietest tile of every channel over the whole batch (with a predecessor that holds signal too, like the scan's own minimum),
// and in out[kChanFloorAll] the quietest of all channels: presence's noise reference of last resort.  gridDim.y = channels,
// gridDim.x workgroups share a channel's tiles and meet in an atomic minimum on the float's bits (positive floats order like their
// bit patterns); out[] preset to 0x7f7f7f7f (3.4e38) by the caller.  ntiles: the FULL tiles (a batch's last tile may be a partial one).
__global__ __launch_bounds__(256) void channel_floor_kernel(const double *__restrict__ ptile, int stride, int ntiles, float *__restrict__ out)
{
    __shared__ float red[256];
    const double *pt = ptile + (size_t)blockIdx.y * stride;
    float mn = 3.0e38f;
    for (int t = 1 + (int)(blockIdx.x * 256 + threadIdx.x); t < ntiles; t += 256 * (int)gridDim.x) {
        const float e = (float)pt[t], ep = (float)pt[t - 1];
        mn = (e > 0.f && ep > 0.f && e < mn) ? e : mn;
    }
    red[threadIdx.x] = mn;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] = red[threadIdx.x + sft] < red[threadIdx.x] ? red[threadIdx.x + sft] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] < 3.0e38f) {
        atomicMin((unsigned int *)out + blockIdx.y, __float_as_uint(red[0]));
        atomicMin((unsigned int *)out + kChanFloorAll, __float_as_uint(red[0]));
    }
}

// VER = true: the exact stage's run (verify.hip.h).  Lanes are the tasks of p.vtasks, 79 (nch) per pseudo-slot, and the stream
// `d` is the task buffer dxt[pseudo-slot][kVerRows][row]: exact demodulated rows in front, the polyphase path's behind them.
// Squelch, records, window lengths and symbol export address the task's REAL window (slot, channel).
template <class LAY, bool VER = false>
__global__ __launch_bounds__(kWinThreads) void window_kernel(
    WindowParams p, const float *__restrict__ d, long long d_rows, const double *__restrict__ P,
    const double *__restrict__ Pt, const double *__restrict__ Qn,
    const float *__restrict__ mmse_g, const uint64_t *__restrict__ ac_lo_g,
    const uint32_t *__restrict__ ac_hi_g,
    double *__restrict__ e_on_out, double *__restrict__ e_off_out, double *__restrict__ snr_out,
    int *__restrict__ win_len, DeviceHit *__restrict__ hits, unsigned int *__restrict__ hit_count,
    FinishRec *__restrict__ fin, unsigned int *__restrict__ fin_count,
    const uint8_t *__restrict__ le_hdr_g, const uint16_t *__restrict__ le_whiten_g,
    const int8_t *__restrict__ le_index_g, int *__restrict__ win_fin, uint32_t *__restrict__ symbits,
    uint32_t *winbits)
{
    constexpr int kWinSlots = LAY::kSlots, kWinRowStride = LAY::kRowStride, kTileFloats = LAY::kTileFloats;
    constexpr int kVecPerRow = LAY::kVecPerRow;
    constexpr int kRows = LAY::kRows;
    __shared__ __attribute__((aligned(16))) float mmse[129 * kMmseStride];
    // slot s's rows start 16 s floats into their bank row: the wave that holds the last lanes of one
    // slot and the first of the next then still reads 32 different banks
    __shared__ __attribute__((aligned(16))) float tile[kWinSlots * kTileFloats];
    // After phase 1 the demod rows are dead and the same LDS holds the sliced symbols of every lane, the
    // access-code tables of phase 2 (the 4 high bits of the 68 as bytes) and the LE header table.  With
    // 29 rows per chunk (BTGPU_WIN_ROWS=29, C79 layout) the footprint is 39.8 KB, four workgroups per CU
    // instead of three -- and the kernel takes the same 0.31 ms: 130 M vector instructions, many of
    // them half-rate compares / selects / conversions, fill its SIMDs' issue slots at either occupancy.
    static_assert(kWinSlots * kTileFloats * 4 >= kBitWords * kWinThreads * 4 + 3 * 256 * 9 + 1024, "phase-2 tables must fit the dead tile");
    uint32_t *bits = (uint32_t *)tile;                                        // [kBitWords][kWinThreads]
    uint64_t *ac_lo = (uint64_t *)(tile + kBitWords * kWinThreads);           // [3][256]
    uint8_t *ac_hi = (uint8_t *)(ac_lo + 3 * 256);                            // [3][256]  (bits 64..67 of a column)
    uint8_t *le_hdr = ac_hi + 3 * 256;                                        // [4][256]  (loaded with the tables, phase 2)
    __shared__ int s_live[2];
    for (int i = threadIdx.x; i < 129 * 8; i += blockDim.x) mmse[(i >> 3) * kMmseStride + (i & 7)] = mmse_g[i];

    const int nch = p.nch;
    const int sl = (int)threadIdx.x / nch;                       // slot of this lane inside the workgroup
    const int c = (int)threadIdx.x - sl * nch;                   // column of the stream rows this lane reads
    const int k = blockIdx.x * kWinSlots + sl;                   // slot (VER: pseudo-slot) whose rows this lane reads
    // the window the lane stands for: (k, c) itself, or the task's
    int kq = k, cq = c;
    int emit_from = 0, le_emit_from = 0;             // VER: classic / access-address records below these offsets left in the first run
    bool lane_ok = sl < kWinSlots && k < p.S;
    if (VER) {
        unsigned int ntask = p.vcount[0];
        if (ntask > (unsigned int)p.vcap) ntask = (unsigned int)p.vcap;
        if ((unsigned int)(blockIdx.x * kWinSlots * nch) >= ntask) return;          // uniform: no task for this workgroup
        const unsigned int q = (unsigned int)(k * nch + c);
        lane_ok = sl < kWinSlots && q < ntask;
        const int wr = lane_ok ? p.vtasks[q].w : 0;
        kq = wr / nch; cq = wr - kq * nch;
        emit_from = lane_ok ? p.vtasks[q].emit_from : 0;
        le_emit_from = lane_ok ? p.vtasks[q].le_emit_from : 0;
    }
    const long long w = (long long)kq * nch + cq;
    int nmax = 0;                                    // symbols this lane will produce in phase 1
    double snr = 0.0;
    if (VER && lane_ok) {                            // a task passed the squelch in the first run -- or, deferred, is judged here
        snr = p.deferred ? p.snr_arr[w] : p.vtasks[k * nch + c].snr;
        win_len[w] = -1;
        if (!p.deferred || snr >= p.target_snr) nmax = kDetectSyms;
    }
    if (!VER && lane_ok && p.deferred) {             // deferred squelch: every window runs, the decision comes with the records
        win_len[w] = -1;
        nmax = kDetectSyms;
    }

    // ---- squelch: multi_block::channel_samples energy + check_snr (multi_block.cc:206-293) ----
    if (!VER && lane_ok && !p.deferred) {
        double e_on = 0.0;
        for (int j = 0; j < p.blocks_per_window; j++) e_on += P[(size_t)cq * p.nb + kq + j];
        if (p.tail > 0) e_on += Pt[(size_t)cq * p.nb + kq + p.blocks_per_window];
        e_on /= (double)p.ddc_out;
        const double e_off = Qn[(size_t)cq * p.qn_stride + kq] / (double)p.noise_out;
        snr = 10.0 * log10(e_on / e_off);
        e_on_out[w] = e_on; e_off_out[w] = e_off; snr_out[w] = snr;
        win_len[w] = -1;
        if (snr >= p.target_snr) nmax = kDetectSyms;
    }

    // ---- exact rows (exact.hip.h) ----
    // The clock-recovery loop quantises its phase to 1/128 sample, so a 1e-6 difference of the polyphase stream parts the two
    // trajectories in ~10 % of the windows within 150 symbols; in noise that is invisible, across a burst it can move the access
    // code by a symbol, change an error count or lose the packet on one side.  So the rows of every BUSY window have been
    // recomputed by the reference's own arithmetic before this kernel runs (presence_kernel -> bm1 -> exact_rows_kernel, in place):
    // cov_rows of this window, from its first row on.  A classic hit inside them is final.  One that reaches beyond them (a
    // packet under the presence threshold, a hit born from noise) is a claim: the window is listed, its rows are marked (bm2) and
    // recomputed, and the second run -- this kernel with VER = true -- emits what the first has not.
    int vslot = -1, vspan = 0;
    int cov_rows = 0, first_uncov = -1;
    if (!VER && p.verify && lane_ok && nmax != 0) cov_rows = exact_covered_rows(p.bm1, p.bm_tiles, (long long)kq * p.outs_per_slot, cq);
    auto ver_rows = [&](int span) { return ver_rows_of(p, span); };

    if (p.dbg_stop == 1) return;
    // ---- phase 1: M&M ----
    const int demod_n = p.ddc_out - 1;
    const unsigned int ni = (unsigned int)(demod_n - 8);
    if (nmax > demod_n) nmax = demod_n;
    float mu = p.mu0, omega = p.omega0, last = 0.f;
    unsigned int ii = 0;
    int oo = 0;
    uint32_t cur = 0;
    // sliced symbols leave phase 1 through a per-workgroup scratch in HBM ([word][lane], one
    // coalesced store per 32 symbols) and come back into the dead tile for phase 2
    uint32_t *gbits = winbits + (size_t)blockIdx.x * (kBitWords * kWinThreads) + threadIdx.x;
    const float *mytile = tile + (sl < kWinSlots ? sl : 0) * kTileFloats;

    // Chunk schedule.  A lane leaves a chunk with its input index just past the chunk's last usable
    // row (index + 8 rows must be staged), so the next chunk always starts kRows - 7 rows further:
    // the schedule is static and the rows of chunk n + 1 are fetched into registers before the
    // recursion over chunk n starts -- the HBM round trip hides under the dependent arithmetic.
    // Every load is unconditional (clamped address): a load under a lane-dependent branch would be
    // waited for at the end of that branch, nine round trips per chunk.
    constexpr int kAdv = kRows - 7;
    constexpr int kVec = kRows * kVecPerRow;                  // float4 per chunk and slot (the used part of the 80-float rows)
    constexpr int kTot = kVec * kWinSlots;
    constexpr int kPer = (kTot + kWinThreads - 1) / kWinThreads;
    float4 v[kPer];
    // Per-thread constants of the staging copy, computed once: byte offset of element j from the
    // workgroup's first row (chunk 0) and its float4 index in the LDS tile; bit 31 marks row 0 of a
    // window (policy Q1: demod_out[0] = 0).  Per chunk an element then costs one add and one clamp.
    // The clamp only keeps the address inside the stream: rows past the end belong to no window of
    // this batch (every window's ddc_out rows exist) and are never consumed.
    constexpr int kRowBytes = kVecPerRow * 16;
    uint32_t goff[kPer], loff[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const int i = (int)threadIdx.x + j * kWinThreads;
        const int ic = i < kTot ? i : kTot - 1;
        const int s = ic / kVec, iv = ic - s * kVec;
        const int r = iv / kVecPerRow, q4 = iv - r * kVecPerRow;
        goff[j] = (uint32_t)((s * p.rows_per_slot + r) * kRowBytes + q4 * 16);
        loff[j] = (uint32_t)(s * (kTileFloats / 4) + r * (kWinRowStride / 4) + q4) | (r == 0 ? 0x80000000u : 0u);
    }
    const long long wg_row0 = (long long)blockIdx.x * kWinSlots * p.rows_per_slot;
    const char *wgb = (const char *)(d + (size_t)wg_row0 * (kVecPerRow * 4));
    const long long span = (d_rows - wg_row0) * kRowBytes - 16;             // last float4 of the stream (d_rows > wg_row0)
    const uint32_t max_off = span > 0xFFFFFFF0LL ? 0xFFFFFFF0u : (uint32_t)span;
    auto fetch = [&](int base) {
        const uint32_t boff = (uint32_t)base * (uint32_t)kRowBytes;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const uint32_t o = goff[j] + boff;
            v[j] = *(const float4 *)(wgb + (o < max_off ? o : max_off));
        }
    };
    int base = 0;
    fetch(0);
    for (int it = 0;; it++) {
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int i = (int)threadIdx.x + j * kWinThreads;
            const bool zero = base == 0 && (loff[j] >> 31);
            const float4 t = make_float4(zero ? 0.f : v[j].x, zero ? 0.f : v[j].y, zero ? 0.f : v[j].z, zero ? 0.f : v[j].w);
            if (i < kTot) ((float4 *)tile)[loff[j] & 0x7fffffffu] = t;
        }
        if (threadIdx.x == 0) s_live[it & 1] = 0;
        __syncthreads();
        fetch(base + kAdv);
        unsigned int lim = (unsigned int)(base + kRows - 8);
        if (lim > ni - 1) lim = ni - 1;                          // while (ii < ni) of the reference
        // byte offset of this lane's column in row `base` of the tile (24-bit arithmetic: one v_mad_u32_u24 per symbol)
        const uint32_t colb = (uint32_t)(((int)(mytile - tile) + c - base * kWinRowStride) * 4);
        while (ii <= lim && oo < nmax) {
            // interpolate: sum_q T[imu][7-q] * in[ii+q], q ascending
            // mu = x - floor(x) lies in [0, 1] (1.0 when x is a tiny negative number): imu in 0..128, no clamp needed
            // rintf(mu * 128) through the mantissa: mu * 128 + 1.5 * 2^23 is rounded to the nearest integer (ties to even, as
            // rintf does) and carries it in its low bits -- 0x4B400000 + imu, of which the 24-bit multiply takes 0x400000 + imu:
            // one FMA and one multiply-add instead of multiply, round, convert, multiply
            const uint32_t tb24 = __float_as_uint(fmaf(mu, 128.0f, 12582912.0f));
            const uint32_t toff = __umul24(tb24, (uint32_t)(kMmseStride * 4)) - 0x400000u * (uint32_t)(kMmseStride * 4);
            // the interpolator row (two 16-byte reads) and eight 4-byte sample reads with immediate offsets, written out: left
            // to the compiler the sample reads are paired into ds_read2_b32, whose offsets reach 1020 bytes while the rows
            // are 4 * kWinRowStride apart -- three address additions per symbol at the same number of LDS cycles, in a loop
            // that is bound by vector-instruction issue
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f ta, tb;                                                        // T[4..7], T[0..3]
            float x0, x1, x2, x3, x4, x5, x6, x7;
            const uint32_t ioff = __umul24(ii, (uint32_t)(kWinRowStride * 4)) + colb;
#if defined(__HIP_DEVICE_COMPILE__)
            {
                constexpr int RB = kWinRowStride * 4;
                const uint32_t ia = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)tile + ioff;
                const uint32_t tadr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)mmse + toff;
                asm volatile("ds_read_b128 %0, %11 offset:16\n\tds_read_b128 %1, %11\n\t"
                             "ds_read_b32 %2, %10\n\tds_read_b32 %3, %10 offset:%12\n\tds_read_b32 %4, %10 offset:%13\n\t"
                             "ds_read_b32 %5, %10 offset:%14\n\tds_read_b32 %6, %10 offset:%15\n\tds_read_b32 %7, %10 offset:%16\n\t"
                             "ds_read_b32 %8, %10 offset:%17\n\tds_read_b32 %9, %10 offset:%18\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ta), "=&v"(tb), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                             : "v"(ia), "v"(tadr), "i"(RB), "i"(2 * RB), "i"(3 * RB), "i"(4 * RB), "i"(5 * RB), "i"(6 * RB), "i"(7 * RB)
                             : "memory");
            }
#else
            {
                const char *trow = (const char *)mmse + toff;
                const float *in = (const float *)((const char *)tile + ioff);
                ta = *(const v4f *)(trow + 16); tb = *(const v4f *)trow;
                x0 = in[0]; x1 = in[kWinRowStride]; x2 = in[2 * kWinRowStride]; x3 = in[3 * kWinRowStride];
                x4 = in[4 * kWinRowStride]; x5 = in[5 * kWinRowStride]; x6 = in[6 * kWinRowStride]; x7 = in[7 * kWinRowStride];
            }
#endif
            float acc = 0.f;
            acc = fmaf(ta.w, x0, acc);
            acc = fmaf(ta.z, x1, acc);
            acc = fmaf(ta.y, x2, acc);
            acc = fmaf(ta.x, x3, acc);
            acc = fmaf(tb.w, x4, acc);
            acc = fmaf(tb.z, x5, acc);
            acc = fmaf(tb.y, x6, acc);
            acc = fmaf(tb.x, x7, acc);
            const float out = acc;
            ii += (unsigned int)mm_update(out, last, omega, mu, p);
            // slicer: one bit per symbol
            cur = sym_push(cur, out);
            if ((oo & 31) == 31) {
                int o2 = oo;
                BTGPU_OPAQUE(o2);                                // (the word index is formed here, once per 32 symbols, not carried through the loop)
                gbits[(o2 >> 5) * kWinThreads] = sym_word(cur, 32);
            }
            oo++;
        }
        if (oo < nmax && ii < ni) s_live[it & 1] = 1;            // this lane needs another chunk
        __syncthreads();
        if (!s_live[it & 1]) break;                              // uniform
        base += kAdv;
    }
    if (oo & 31) gbits[(oo >> 5) * kWinThreads] = sym_word(cur, oo & 31);
    if (p.dbg_stop == 2) return;
    {
        // every lane reads back the words it wrote itself (all loads in flight together)
        const int nw = (oo + 31) >> 5;
        uint32_t wv[kBitWords];
#pragma unroll
        for (int j = 0; j < kBitWords; j++) wv[j] = gbits[(j < nw ? j : 0) * kWinThreads];
        uint64_t tl[4]; uint32_t th[4];
        const uint32_t lh = p.le ? ((const uint32_t *)le_hdr_g)[threadIdx.x] : 0u;     // 1024 bytes: one word per lane
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = (int)threadIdx.x + j * kWinThreads;
            tl[j] = ac_lo_g[i < 768 ? i : 767]; th[j] = ac_hi_g[i < 768 ? i : 767];
        }
#pragma unroll
        for (int j = 0; j < kBitWords; j++) bits[j * kWinThreads + threadIdx.x] = j < nw ? wv[j] : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = (int)threadIdx.x + j * kWinThreads;
            if (i < 768) { ac_lo[i] = tl[j]; ac_hi[i] = (uint8_t)th[j]; }
        }
        if (p.le) ((uint32_t *)le_hdr)[threadIdx.x] = lh;
    }
    __syncthreads();
    if (nmax == 0) return;
    uint32_t *mybits = bits + threadIdx.x;
    // the search below is lane-private: every lane reads only its own words

    // ---- phase 2: access-code search over offsets [0, limit) ----
    const int len1 = oo;                                         // 693, or the whole window if shorter
    int limit = len1 - 68 < 625 ? len1 - 68 : 625;
    int resume = 0, nhits = 0;
    auto emit_classic = [&](int cpos, uint32_t lap, int err) {
        if (!VER && p.verify) {
            // the rows this record stands on: to the end of its access code (+ the header the host will read)
            const int span_ = cpos + 72 + p.span_extra + 8;
            if (first_uncov >= 0 || ver_rows(span_) > cov_rows) {
                if (first_uncov < 0) {
                    const unsigned int s_ = atomicAdd(&p.vcount[0], 1u);
                    if (s_ < (unsigned int)p.vcap) { vslot = (int)s_; first_uncov = cpos; }
                    else atomicAdd(&p.vcount[2], 1u);             // list full: this window keeps the polyphase path's records
                }
                if (vslot >= 0) { vspan = span_ > vspan ? span_ : vspan; return; }
            }
        }
        if (VER && cpos < emit_from) return;                      // (the first run emitted it, from the same exact rows)
        const unsigned int slot_h = atomicAdd(hit_count, 1u);
        if (slot_h < (unsigned int)p.max_hits) {
            DeviceHit h;
            h.slot = (uint32_t)kq; h.channel_idx = cq; h.offset = cpos;
            h.lap = lap; h.ac_errors = err; h.kind = 0; h.snr = snr; h.nsym = -1; h.sub = 0; h.sym = -1; h.pad_ = 0;
            hits[slot_h] = h;
        }
    };
    if (p.btbb) {
        // [EXT libbtbb, unpinned] btbb_find_ac(symbols, latest_ac, LAP_ANY, 1, &pkt) as multi_LAP calls it
        // (lib/multi_LAP_impl.cc:93): per offset the 64-symbol sync word; gate on the 7-bit Barker field
        // (sync bits 57..63, distance <= 1, then replaced by the nearest valid pattern); zero BCH(64,30)
        // syndrome or a single-bit error over sync bits 0..57 corrected; first hit wins.  The
        // syndrome is formed against the access code regenerated from the received LAP (systematic
        // code: syndrome = received parity ^ re-encoded parity): weight 1 = a parity bit in error,
        // equal to the parity column of LAP bit k = that LAP bit in error.
        uint32_t q0 = mybits[0], q1 = mybits[kWinThreads], q2 = mybits[2 * kWinThreads];
        for (int b = 0; b * 32 < limit && nhits == 0; b++) {
            const uint32_t v57 = __funnelshift_r(q1, q2, 25), v58 = __funnelshift_r(q1, q2, 26);
            const uint32_t v59 = __funnelshift_r(q1, q2, 27), v60 = __funnelshift_r(q1, q2, 28);
            const uint32_t v61 = __funnelshift_r(q1, q2, 29), v62 = __funnelshift_r(q1, q2, 30);
            const uint32_t v63 = __funnelshift_r(q1, q2, 31);
            // mismatches against 0x27 (field bit j = sync bit 57 + j): 1,1,1,0,0,1,0
            const uint32_t n0 = ~v57, n1 = ~v58, n2 = ~v59, n3 = v60, n4 = v61, n5 = ~v62, n6 = v63;
            const uint32_t t1 = xor3(n0, n1, n2), u1 = maj3(n0, n1, n2);
            const uint32_t t2 = xor3(n3, n4, n5), u2 = maj3(n3, n4, n5);
            const uint32_t u3 = maj3(t1, t2, n6);
            const uint32_t d1 = xor3(u1, u2, u3), d2 = maj3(u1, u2, u3);   // d = d0 + 2 d1 + 4 d2
            uint32_t ok = ~(d1 ^ d2);                                  // d in {0, 1, 6, 7}: min(d, 7 - d) <= 1
            while (ok && nhits == 0) {
                const int j = __ffs(ok) - 1;
                ok &= ok - 1;
                const int cpos = 32 * b + j;
                if (cpos >= limit) break;
                const uint32_t x0 = j ? ((q0 >> j) | (q1 << (32 - j))) : q0;
                const uint32_t x1 = j ? ((q1 >> j) | (q2 << (32 - j))) : q1;
                uint64_t sw = ((uint64_t)x1 << 32) | x0;
                const uint32_t top7 = (uint32_t)(sw >> 57);
                const uint64_t fixed = __popc(top7 ^ 0x27u) <= 1 ? 0x27ull : 0x58ull;
                sw = (sw & 0x01ffffffffffffffull) | (fixed << 57);
                uint32_t lap = (uint32_t)(sw >> 34) & 0xffffff;
                const uint64_t elo = p.a0_lo ^ ac_lo[lap & 0xff] ^ ac_lo[256 + ((lap >> 8) & 0xff)] ^ ac_lo[512 + (lap >> 16)];
                const uint32_t ehi = p.a0_hi ^ ac_hi[lap & 0xff] ^ ac_hi[256 + ((lap >> 8) & 0xff)] ^ ac_hi[512 + (lap >> 16)];
                const uint64_t expect = (elo >> 4) | ((uint64_t)(ehi & 0xf) << 60);
                const uint64_t synd = sw ^ expect;                         // parity bits only
                int err = -1;
                if (synd == 0) err = 0;
                else if ((synd & (synd - 1)) == 0) err = 1;
                else {
                    for (int kb = 0; kb < 24; kb++)
                        if (synd == p.btbb_pcol[kb]) { lap ^= 1u << kb; err = 1; break; }
                }
                if (err >= 0) {
                    emit_classic(cpos, lap, err);
                    nhits++;
                    resume = cpos + 68;
                }
            }
            q0 = q1; q1 = q2; q2 = mybits[(b + 3) * kWinThreads];
        }
        limit = 0;                                                 // skip the in-tree search below
    }
    search_classic(mybits, limit, kSymbolsShortAcDev, p.mode == 0, p.a0_lo, p.a0_hi, ac_lo, ac_hi, resume, nhits, emit_classic);
    if (p.dbg_stop == 3) return;
    if (vslot >= 0) {
        // the rest of this window's records come from the second run: mark the rows they stand on (what presence has not marked)
        const int rows = ver_rows(vspan);
        const long long g_lo = (long long)kq * p.outs_per_slot;
        exact_mark(p.bm2, p.bm1, p.bm_tiles, g_lo, g_lo + rows, cq, &p.vcount[1]);
        VerifyTask t_;
        t_.w = (int32_t)w; t_.n_exact = rows; t_.snr = snr; t_.emit_from = first_uncov; t_.le_emit_from = 0;
        p.vtasks[vslot] = t_;
        return;
    }
    // ---- LE pass: le_packet::sniff_aa (lib/packet_impl.cc:1452-1527) with the loop of
    // lib/multi_sniffer_impl.cc:129-149.  `len` keeps what the classic pass left (Q6): the search
    // limit is min(len' - 68, 625) with len' = len - (last classic hit + 68); records carry
    // sub = len - len' so that nsym = len' - offset.
    const int le_index = (p.le && p.mode != 0) ? (int)le_index_g[p.low_channel + cq] : -1;
    if (le_index >= 0) {
        const int sub = nhits > 0 ? resume : 0;                  // resume = last classic hit + 68
        // phase 1 produced min(len, 693) symbols; len' - 68 >= 625 whenever the window was
        // truncated at 693 (len >= 1875 in sniffer geometry), otherwise len is known exactly
        int le_limit = (ii >= ni || oo >= demod_n) ? (len1 - sub - 68) : 625;
        if (le_limit > 625) le_limit = 625;
        const bool access = le_index >= 37;
        const uint8_t *phl = le_hdr + (access ? 0 : 512), *phm = phl + 256;
        const uint32_t wmask = le_whiten_g[le_index];
        int le_resume = 0;
        int le_uncov = -1;                                       // first run: the first access-address hit that stands on rows exact.hip.h has not covered
        uint32_t q0 = mybits[0], q1 = mybits[kWinThreads], q2;
        for (int b = 0; b * 32 < le_limit; b++) {
            q2 = mybits[(b + 2) * kWinThreads];
            // preamble + first AA bit (9 symbols): distance to 0x0aa or its complement 0x155
            uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;         // bit-sliced mismatch count vs 0x0aa
            for (int i = 0; i < 9; i++) {
                uint32_t wv = i ? __funnelshift_r(q0, q1, i) : q0;
                if (i & 1) wv = ~wv;                                 // 0x0aa: symbols 0,1,0,1,0,1,0,1,0
                const uint32_t c0 = cnt0 & wv; cnt0 ^= wv;
                const uint32_t c1 = cnt1 & c0; cnt1 ^= c0;
                const uint32_t c2 = cnt2 & c1; cnt2 ^= c1;
                cnt3 ^= c2;
            }
            // d = mismatches (0..9); min(d, 9 - d) <= maxd  (maxd = 2 on access channels, else 0)
            uint32_t cand;
            {
                const uint32_t is0 = ~(cnt0 | cnt1 | cnt2 | cnt3);
                const uint32_t is9 = cnt3 & ~cnt2 & ~cnt1 & cnt0;
                cand = is0 | is9;
                if (access) {
                    const uint32_t le2 = ~cnt3 & ~cnt2 & ~(cnt1 & cnt0);                 // d <= 2
                    const uint32_t ge7 = (cnt3) | (~cnt3 & cnt2 & cnt1 & cnt0);           // d >= 7
                    cand |= le2 | ge7;
                }
            }
            while (cand) {
                const int j = __ffs(cand) - 1;
                cand &= cand - 1;
                const int cpos = 32 * b + j;
                if (cpos < le_resume || cpos >= le_limit) continue;
                const uint32_t x0 = j ? ((q0 >> j) | (q1 << (32 - j))) : q0;      // symbols cpos .. +31
                const uint32_t x1 = j ? ((q1 >> j) | (q2 << (32 - j))) : q1;      // symbols cpos+32 .. +63
                const uint32_t pre = x0 & 0x1ff;
                int dist = __popc(pre ^ 0x0aa);
                dist = dist < 9 - dist ? dist : 9 - dist;
                const uint32_t aa = (x0 >> 8) | (x1 << 24);                        // symbols cpos+8 .. +39
                const uint32_t hdr = ((x1 >> 8) & 0xffff) ^ wmask;                 // symbols cpos+40 .. +55, de-whitened
                dist += phl[hdr & 0xff] + phm[hdr >> 8];
                int maxd = 0;
                if (access) { dist += __popc(aa ^ 0x8e89bed6u); maxd = 2; }
                if (dist <= maxd) {
                    // An access-address hit outside the exact rows -- in practice one born from noise symbols: a real advert's air time
                    // is busy -- is a claim like a classic one: the window goes to the second run, which settles it on exact rows
                    // (the records of the product's side then are the reference's wherever the product saw anything at all; what only
                    // the reference's noise shows stays on its side: ~1 % of the noise-born records, bench.py parity.aa_records).
                    bool emit = !(VER && cpos < le_emit_from);
                    if (!VER && p.verify) {
                        const int span_ = cpos + 40 + 16 + 8;        // preamble + address, the header bits the test reads, margin
                        if (le_uncov >= 0 || ver_rows(span_) > cov_rows) {
                            if (le_uncov < 0) {
                                const unsigned int s_ = atomicAdd(&p.vcount[0], 1u);
                                if (s_ < (unsigned int)p.vcap) { vslot = (int)s_; le_uncov = cpos; }
                                else atomicAdd(&p.vcount[2], 1u);
                            }
                            if (vslot >= 0) { vspan = span_ > vspan ? span_ : vspan; emit = false; }
                        }
                    }
                    const unsigned int slot_h = emit ? atomicAdd(hit_count, 1u) : 0xffffffffu;
                    if (slot_h < (unsigned int)p.max_hits) {
                        DeviceHit h;
                        h.slot = (uint32_t)kq; h.channel_idx = cq; h.offset = cpos;
                        h.lap = aa; h.ac_errors = 0; h.kind = 1; h.snr = snr; h.nsym = -1; h.sub = sub; h.sym = -1; h.pad_ = 0;
                        hits[slot_h] = h;
                    }
                    nhits++;
                    le_resume = cpos + 40;
                }
            }
            q0 = q1; q1 = q2;
        }
        if (vslot >= 0) {
            // (reserved by the LE pass: every classic record of this window has left already -- all stood on exact rows)
            const int rows = ver_rows(vspan);
            const long long g_lo = (long long)kq * p.outs_per_slot;
            exact_mark(p.bm2, p.bm1, p.bm_tiles, g_lo, g_lo + rows, cq, &p.vcount[1]);
            VerifyTask t_;
            t_.w = (int32_t)w; t_.n_exact = rows; t_.snr = snr; t_.emit_from = 0x3fffffff; t_.le_emit_from = le_uncov;
            p.vtasks[vslot] = t_;
            return;
        }
    }
    if (nhits > 0) {
        const bool ended = ii >= ni || oo >= demod_n;                  // the window ended inside phase 1
        if (ended) win_len[w] = oo;
        if ((!ended && p.want_len) || p.syms) {
            // the handlers need len = symbols in the whole window (and, with BTGPU_FLAG_SYMBOLS, the
            // symbols): hand the M&M state to finish_kernel (dense waves of hit windows)
            const unsigned int f = atomicAdd(fin_count, 1u);
            FinishRec r;
            r.w = (int32_t)w; r.ii = ii; r.oo = oo; r.mu = mu; r.omega = omega; r.last = last;
            r.done = ended ? 1 : 0;
            r.pad_ = 0;
            fin[f] = r;
            if (p.syms) {
                win_fin[w] = (int)f;
                uint32_t *dst = symbits + (size_t)f * kSymWords;
                const int nw = (oo + 31) >> 5;
                for (int j = 0; j < kSymWords; j++) dst[j] = j < nw ? mybits[j * kWinThreads] : 0u;
            }
        }
    }
}

constexpr int kFinRows = 16;       // rows per refill
constexpr int kFinRing = 32;       // rows resident per lane
constexpr int kFinSlab = kFinRing + 8 + 1;   // + the first eight slots again behind the ring (the 8-tap window never wraps),
                                             // + 1: an odd lane pitch keeps equal offsets of different lanes on different banks
constexpr int kFinLanes = 48;      // windows per workgroup.  48 * 41 * 4 + 4128 bytes = 12 KB of LDS: what a CU has left
                                   // beside three bank workgroups, so the tail does not push one of them off the CU
template <bool SYMS>
__global__ __launch_bounds__(64) void finish_kernel(
    WindowParams p, const float *__restrict__ d, int drow, long long d_rows,
    const float *__restrict__ mmse_g, const FinishRec *__restrict__ fin,
    const unsigned int *__restrict__ fin_count, int *__restrict__ win_len, uint32_t *__restrict__ symbits,
    const float *__restrict__ dcol = nullptr)
{
    constexpr unsigned int RING = kFinRing, MASK = RING - 1;
    __shared__ __attribute__((aligned(16))) float mmse[129 * 8];
    __shared__ float slab[kFinLanes * kFinSlab];
    const unsigned int n = *fin_count;
    if (blockIdx.x * kFinLanes >= n) return;                     // uniform: nothing for this workgroup
    const unsigned int stride = gridDim.x * kFinLanes;           // records beyond the grid: next round
    // a few dozen strictly sequential waves next to the throughput kernels of the following batch:
    // give them the highest wave issue priority, they use a fraction of a percent of the issue slots
    if (p.fin_prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p.fin_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.fin_prio == 1) __builtin_amdgcn_s_setprio(1);
    for (int i = threadIdx.x; i < 129 * 8; i += blockDim.x) mmse[i] = mmse_g[i];
    __syncthreads();
  for (unsigned int f = blockIdx.x * kFinLanes + threadIdx.x; f < n; f += stride) {
    const FinishRec r = fin[f];
    if (r.done) continue;
    const int k = r.w / p.nch, c = r.w - k * p.nch;
    const int demod_n = p.ddc_out - 1;
    const unsigned int ni = (unsigned int)(demod_n - 8);
    const long long row0 = (long long)k * p.outs_per_slot;
    const float *col = d + (size_t)row0 * drow + c;              // this window's demod samples: every drow-th float of the stream
    const unsigned int nvalid = (unsigned int)((d_rows - row0) < p.ddc_out ? (d_rows - row0) : p.ddc_out);
    float mu = r.mu, omega = r.omega, last = r.last;
    unsigned int ii = r.ii;
    int oo = r.oo;
    float *my = slab + threadIdx.x * kFinSlab;
    uint32_t *sb = SYMS ? symbits + (size_t)f * kSymWords : nullptr;
    // partially filled word left by the window kernel, back into the collecting format (sym_push)
    uint32_t cur = (SYMS && (oo & 31)) ? __brev(~sb[oo >> 5]) >> (32 - (oo & 31)) : 0u;
    uint32_t pend_w = 0u;
    int pend_i = -1;                                             // a completed symbol word waiting to be stored
    // rows [hi - RING, hi) are resident, row q in slot q & 31 (slots 0..7 also at 32..39); refills are whole
    // 16-row blocks, so a block is either slots 0..15 (guard copy of its first half) or 16..31
    unsigned int hi = ii & ~(unsigned int)(kFinRows - 1);
    auto fetch = [&](float *v) {                            // rows [hi, hi + 16): unconditional loads, values selected later
        if (p.dbg_stop == 9) {                                   // timing experiment (BTGPU_WIN_STOP=9): no stream traffic
#pragma unroll
            for (int j = 0; j < kFinRows; j++) v[j] = 0.01f * (float)((hi + j) & 7) - 0.03f;
            return;
        }
        if (dcol) {
            // the 100-bin bank's tile-blocked copy [tile][80][25]: sample q of channel c sits at
            // q + 25 (79 (q / 25) + c) -- sixteen consecutive samples share one or two cache lines
#pragma unroll
            for (int j = 0; j < kFinRows; j++) {
                const unsigned int idx = hi + j;
                const unsigned int q = (unsigned int)row0 + (idx < nvalid ? idx : nvalid - 1);
                const unsigned int tq = (unsigned int)(((unsigned long long)q * 0x51EB851Full) >> 35);   // q / 25 (v_mul_hi_u32)
                v[j] = dcol[(size_t)(q + 25u * (79u * tq + (unsigned int)c))];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < kFinRows; j++) { const unsigned int idx = hi + j; v[j] = col[(size_t)(idx < nvalid ? idx : nvalid - 1) * drow]; }
    };
    auto put = [&](const float *v) {                             // rows [hi, hi + 16) -> ring; rows past the stream read as 0
        const unsigned int s0 = hi & MASK;                       // 0 or 16
#pragma unroll
        for (int j = 0; j < kFinRows; j++) my[s0 + j] = hi + j < nvalid ? v[j] : 0.f;
        if (s0 == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) my[RING + j] = hi + j < nvalid ? v[j] : 0.f;
        }
        hi += kFinRows;
    };
    {
        float v0[kFinRows], v1[kFinRows];
        fetch(v0); put(v0);
        fetch(v1); put(v1);
    }
    while (ii < ni && oo < demod_n) {
        // issue the loads of the next block now; they land while the steps below run
        float v[kFinRows];
        fetch(v);
        // consume every step whose 8-tap window lies inside the resident rows [.., hi)
        while (ii + 8 <= hi && ii < ni && oo < demod_n) {
            // rintf(mu * 128) through the mantissa (window_kernel): 0x4B400000 + imu, shifted to the 32-byte row
            const uint32_t tb24 = __float_as_uint(fmaf(mu, 128.0f, 12582912.0f));
            const char *trow = (const char *)mmse + ((tb24 << 5) - (0x4B400000u << 5));
            const float4 ta = *(const float4 *)(trow + 16);
            const float4 tb = *(const float4 *)trow;
            const float *in = my + (ii & MASK);
            float acc = 0.f;
            acc = fmaf(ta.w, in[0], acc);
            acc = fmaf(ta.z, in[1], acc);
            acc = fmaf(ta.y, in[2], acc);
            acc = fmaf(ta.x, in[3], acc);
            acc = fmaf(tb.w, in[4], acc);
            acc = fmaf(tb.z, in[5], acc);
            acc = fmaf(tb.y, in[6], acc);
            acc = fmaf(tb.x, in[7], acc);
            const float out = acc;
            ii += (unsigned int)mm_update(out, last, omega, mu, p);
            if (SYMS) {
                cur = sym_push(cur, out);
                // a completed word is stored BEHIND the refill below: vector-memory operations retire in order, so a store
                // issued here sits in front of the loads of `fetch` and the wait for those loads (put) waited for the store's
                // round trip to HBM as well -- the symbol export cost a quarter of this kernel's time (0.69 -> 0.86 ms alone)
                if ((oo & 31) == 31) { pend_w = sym_word(cur, 32); pend_i = oo >> 5; }
            }
            oo++;
        }
        // here ii + 8 > hi (or the window is done): the block of slots that holds rows [hi - RING, hi - RING + 16)
        // lies below ii and takes rows [hi, hi + 16)
        put(v);
        if (SYMS && pend_i >= 0) {                                   // (at most one word completes per refill: ~8 symbols per 16 rows)
            if (pend_i < kSymWords) sb[pend_i] = pend_w;
            pend_i = -1;
        }
    }
    if (SYMS && pend_i >= 0 && pend_i < kSymWords) sb[pend_i] = pend_w;
    if (SYMS && (oo & 31) && (oo >> 5) < kSymWords) sb[oo >> 5] = sym_word(cur, oo & 31);
    win_len[r.w] = oo;
  }
}

// nsym = len - offset for every hit record, once finish_kernel has produced the window lengths
__global__ void nsym_patch_kernel(DeviceHit *__restrict__ hits, const unsigned int *__restrict__ hit_count,
                                  int max_hits, const int *__restrict__ win_len, int nch,
                                  const int *__restrict__ win_fin, const double *__restrict__ snr_arr = nullptr,
                                  double target_snr = 0.0)
{
    unsigned int n = *hit_count;
    if (n > (unsigned int)max_hits) n = (unsigned int)max_hits;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const size_t w = (size_t)hits[i].slot * nch + hits[i].channel_idx;
        const int len = win_len[w];
        hits[i].nsym = len < 0 ? -1 : len - hits[i].sub - hits[i].offset;     // len < 0: BTGPU_FLAG_NO_NSYM, window not continued
        if (win_fin) hits[i].sym = win_fin[w];
        if (snr_arr) {                                                        // deferred squelch: the window's SNR, and its verdict
            const double s = snr_arr[w];
            hits[i].snr = s;
            if (!(s >= target_snr)) hits[i].kind = -1;                         // the reference never looked into this window: no record
        }
    }
}

// multi_block::channel_samples' energy + check_snr's ratio (lib/multi_block.cc:206-293) per (slot, channel) window, from the
// block sums P / Pt and the per-slot noise sums Qn: what window_kernel computes in line when the squelch is not deferred.
__global__ __launch_bounds__(256) void squelch_kernel(WindowParams p, const double *__restrict__ P, const double *__restrict__ Pt,
                                                      const double *__restrict__ Qn, double *__restrict__ e_on_out,
                                                      double *__restrict__ e_off_out, double *__restrict__ snr_out)
{
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (long long)p.S * p.nch) return;
    const int k = (int)(w / p.nch), c = (int)(w - (long long)k * p.nch);
    double e_on = 0.0;
    for (int j = 0; j < p.blocks_per_window; j++) e_on += P[(size_t)c * p.nb + k + j];
    if (p.tail > 0) e_on += Pt[(size_t)c * p.nb + k + p.blocks_per_window];
    e_on /= (double)p.ddc_out;
    const double e_off = Qn[(size_t)c * p.qn_stride + k] / (double)p.noise_out;
    e_on_out[w] = e_on; e_off_out[w] = e_off; snr_out[w] = 10.0 * log10(e_on / e_off);
}

// ------------------------------------------------------------------------------------
// The same search on a captured SYMBOL stream (one bit per symbol, LSB first in 32-bit words --
// samples/channel37.dem of the reference, bit-packed): lane = a 625-offset chunk of the stream,
// its 693 symbols aligned into the LDS layout of window_kernel's phase 2.  Records carry the
// chunk index in `slot` and the offset inside the chunk; the caller adds 625 * slot.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWinThreads) void scan_symbols_kernel(
    const uint32_t *__restrict__ words, unsigned long long nsym, int step, uint64_t a0_lo, uint32_t a0_hi,
    const uint64_t *__restrict__ ac_lo_g, const uint32_t *__restrict__ ac_hi_g,
    DeviceHit *__restrict__ hits, unsigned int *__restrict__ hit_count, int max_hits)
{
    __shared__ uint32_t bits[kBitWords * kWinThreads];
    __shared__ uint64_t ac_lo[3 * 256];
    __shared__ uint32_t ac_hi[3 * 256];
    for (int i = threadIdx.x; i < 768; i += kWinThreads) { ac_lo[i] = ac_lo_g[i]; ac_hi[i] = ac_hi_g[i]; }
    const unsigned long long chunk = (unsigned long long)blockIdx.x * kWinThreads + threadIdx.x;
    const unsigned long long s0 = chunk * 625ull;                       // first symbol of this lane's chunk
    const unsigned long long nwords = (nsym + 31) / 32;
    const unsigned long long w0 = s0 >> 5;
    const int sh = (int)(s0 & 31);
    for (int j = 0; j < kBitWords; j++) {
        const unsigned long long a = w0 + j, b = a + 1;
        const uint32_t lo = a < nwords ? words[a] : 0u, hi = b < nwords ? words[b] : 0u;
        bits[j * kWinThreads + threadIdx.x] = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    }
    __syncthreads();
    if (s0 >= nsym) return;
    const unsigned long long left = nsym - s0;
    const int len1 = left < (unsigned long long)kDetectSyms ? (int)left : kDetectSyms;
    // a stream scan may use the last 68 symbols too (no "one more symbol" rule as in the block's loop)
    const int limit = len1 - 67 < 625 ? len1 - 67 : 625;
    int resume = 0, nhits = 0;
    search_classic(bits + threadIdx.x, limit, step, false, a0_lo, a0_hi, ac_lo, ac_hi, resume, nhits,
                   [&](int cpos, uint32_t lap, int err) {
                       const unsigned int slot_h = atomicAdd(hit_count, 1u);
                       if (slot_h < (unsigned int)max_hits) {
                           DeviceHit h;
                           h.slot = (uint32_t)chunk; h.channel_idx = 0; h.offset = cpos;
                           h.lap = lap; h.ac_errors = err; h.kind = 0; h.snr = 0.0; h.nsym = -1; h.sub = 0; h.sym = -1; h.pad_ = 0;
                           hits[slot_h] = h;
                       }
                   });
}

// ------------------------------------------------------------------------------------
// Header sweep: what basic_rate_piconet::UAP_from_header asks of a packet for
// each of the 64 candidate clocks -- classic_packet::try_clock (lib/packet_impl.cc:1046-1063): the
// 18 header bits out of the 1/3-rate FEC (unfec13, :367-383), unwhitened with the clock's sequence
// (:513-526), the UAP that makes the HEC come out right (UAP_from_hec, :597-609) and the packet
// type.  One wave per classic hit, lane = clock.  `to_header` = symbols from hit.offset to the
// header (72 for the in-tree correlator's offset, 68 for libbtbb's).
// ------------------------------------------------------------------------------------
struct HeaderRec { uint8_t uap[64]; uint8_t type[64]; int32_t fec13_ok; int32_t pad_; };

__global__ __launch_bounds__(64) void header_sweep_kernel(
    const DeviceHit *__restrict__ hits, const unsigned int *__restrict__ hit_count, int max_hits,
    const uint32_t *__restrict__ symbits, const uint32_t *__restrict__ wh18, int to_header,
    HeaderRec *__restrict__ out)
{
    unsigned int n = *hit_count;
    if (n > (unsigned int)max_hits) n = (unsigned int)max_hits;
    const int lane = threadIdx.x;
    const uint32_t mask = wh18[lane];
    for (unsigned int i = blockIdx.x; i < n; i += gridDim.x) {
        const DeviceHit h = hits[i];
        HeaderRec *o = &out[i];
        if (h.kind != 0 || h.sym < 0) {
            o->uap[lane] = 0; o->type[lane] = 0;
            if (lane == 0) { o->fec13_ok = 0; o->pad_ = 0; }
            continue;
        }
        const uint32_t *row = symbits + (size_t)h.sym * kSymWords;
        // lanes 0..17: majority vote of header bit `lane`
        uint32_t bit = 0, dis = 0;
        if (lane < 18) {
            const int b0 = h.offset + to_header + 3 * lane;
            uint32_t v[3];
            for (int j = 0; j < 3; j++) {
                const int b = b0 + j;
                v[j] = (b >> 5) < kSymWords ? (row[b >> 5] >> (b & 31)) & 1u : 0u;
            }
            bit = (v[0] & v[1]) | (v[1] & v[2]) | (v[2] & v[0]);
            dis = (v[0] ^ v[1]) | (v[1] ^ v[2]) | (v[2] ^ v[0]);
        }
        const uint32_t hdr = (uint32_t)(__ballot(bit != 0) & 0x3ffffull);
        const int be = __popcll(__ballot(dis != 0));
        const uint32_t plain = hdr ^ mask;
        const uint32_t data = plain & 0x3ffu;
        uint32_t hec = (plain >> 10) & 0xffu;
        for (int k = 9; k >= 0; k--) {
            if (hec & 0x80u) hec ^= 0x65u;
            hec = ((hec << 1) | (((hec >> 7) ^ (data >> k)) & 1u)) & 0xffu;
        }
        o->uap[lane] = (uint8_t)(__brev(hec) >> 24);
        o->type[lane] = (uint8_t)((plain >> 3) & 0xfu);
        if (lane == 0) { o->fec13_ok = be < 18 / 4 ? 1 : 0; o->pad_ = 0; }
    }
}

// A batch's records leave for the host through THIS kernel: it knows the counts, the host does not when it enqueues the batch.  It
// writes the first min(count, cap) records of each kind -- the packed symbols of the hit windows, the header sweeps, the hit records
// themselves -- straight into page-locked host memory (device-visible), 16 bytes per lane.  Before: fixed-size copies sized for 8192
// hit windows and a second, synchronous copy for the rest -- which, at C8's 11 068 hit windows per batch, waited behind every
// kernel queued on the stream its hardware queue was shared with: 20 ms in every fifth call (profiles/r06_s_c8_*).
__global__ __launch_bounds__(256) void records_out_kernel(const unsigned int *__restrict__ counts /* [0] records, [1] hit windows */, int max_hits,
                                                          const uint4 *__restrict__ symbits, uint4 *__restrict__ h_sym, unsigned int cap_sym,
                                                          const uint2 *__restrict__ hdr, uint2 *__restrict__ h_hdr, unsigned int cap_hdr,
                                                          const uint4 *__restrict__ hits, uint4 *__restrict__ h_hits, unsigned int cap_hits)
{
    static_assert(kSymWords % 4 == 0 && sizeof(HeaderRec) % 8 == 0 && sizeof(DeviceHit) % 16 == 0, "whole 16- / 8-byte pieces per record");
    unsigned int n = counts[0];
    n = n > (unsigned int)max_hits ? (unsigned int)max_hits : n;
    const unsigned int nfin = counts[1];
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h_sym) {
        const size_t m = (size_t)(nfin < cap_sym ? nfin : cap_sym) * (kSymWords / 4);
        for (size_t i = i0; i < m; i += stride) h_sym[i] = symbits[i];
    }
    if (h_hdr) {
        const size_t m = (size_t)(n < cap_hdr ? n : cap_hdr) * (sizeof(HeaderRec) / 8);
        for (size_t i = i0; i < m; i += stride) h_hdr[i] = hdr[i];
    }
    if (h_hits) {
        const size_t m = (size_t)(n < cap_hits ? n : cap_hits) * (sizeof(DeviceHit) / 16);
        for (size_t i = i0; i < m; i += stride) h_hits[i] = hits[i];
    }
}

}  // namespace btgpu