// verify.hip.h -- exact confirmation of the polyphase (FAST) path's records.  gfx950, wave = 64 lanes.
//
// The polyphase channelizer's demodulated stream equals the per-channel direct-form DDC's to ~1e-6, and the reference's
// clock recovery (multi_block::mm_cr, lib/multi_block.cc:128-155) quantises its phase to 1/128 sample: the two
// trajectories part somewhere in most windows.  Over noise nobody can tell; across a burst the access code can come out
// a symbol earlier or later, with a different error count, or -- a carrier offset that puts one symbol level near zero --
// be found on one side only.  So every window that can carry a record of a real packet is RE-RUN through the arithmetic
// of the bit-exact path: the windows window_kernel hands over (a classic hit, or burst energy inside the detection span)
// get
//   verify_ddc_kernel   the reference's per-channel DDC (freq_xlating_fir_filter_ccf [EXT], lib/multi_block.cc:180-205) in
//                       ddc_direct_kernel's summation order, rotator, quadrature demod (multi_block::demod, :158-168) --
//                       only for the channel and the rows of the window the detection can reach
//   verify_fill_kernel  those rows, continued by the polyphase path's own rows, as the task's column of the time-major
//                       task stream dxt[pseudo-slot][kVerRows][drow]
//   window_kernel<LAY, true>   squelch figure carried over, clock recovery + slicer + access-code / LE search on that
//                       stream: the window's records (kernels.hip.h)
// and finish_kernel continues them as before.  Cost: 4 * ntp multiply-adds per recomputed row -- 2.2 M for the ~420 rows
// in front of a packet 85 symbols into its window -- per packet in the capture, not per window.
#pragma once
#include "kernels.hip.h"

namespace btgpu {

constexpr int kVerThreads = 512;      // 8 waves: wave l sums the taps j = l (mod 8) -- the 8 partial sums of the summation order
constexpr int kVerPre = 16;           // 8-byte loads per thread that fetch a tile's input span (8192 samples at most: D <= 50)
constexpr int kVerOuts = 128;         // outputs per tile: lane i of every wave takes outputs 2 i and 2 i + 1 (they share input samples)

struct VerifyParams {
    long long x_len;
    long long first0;                 // x index of (window 0, output 0, tap 0): w0 + first_channel_sample
    int D, ntp, slot;                 // decimation, padded filter length (multiple of 8), samples per slot
    uint32_t inv2d;                   // 2^32 / (2 D) + 1: n / (2 D) = mulhi(n, inv2d) for the n of a tile
    int mp, F;                        // tapsv: [nch][8][mp] complex, class-major copy of the reversed taps, F zeros in front and behind
    const float2 *rot; int Q;         // de-rotation table [nch][Q] by window-local output index (Q = 0: rot_step_turns)
    const double *rot_step_turns;
    const float *atan_tab; float gain;
    const VerifyTask *tasks; const unsigned int *vcount; int vcap;
    const uint32_t *tiles; const unsigned int *tcount; unsigned int tiles_cap;   // tile lists by channel: tiles[c * tiles_cap ..], tcount[c] entries
    int dx_stride;                    // floats per task in the output (kVerRows; the long tasks of BTGPU_FLAG_EXACT_PAYLOAD: whole windows)
    const unsigned int *tstart;       // nullptr, or [nch]: entries of each list that an earlier launch of this batch has taken
    int nch;
};
constexpr int kVerMaxTiles = (kVerRows + kVerTile - 1) / kVerTile;     // tiles of a task at most (12)
// class-major, zero-padded copy of a direct-form bank's taps for verify_ddc_kernel: out[(c * 8 + l) * mp + F + m] = taps[c][l + 8 m]
// (+ 16 zeros behind: the march runs in blocks of four steps and fetches the next block's taps while it works on the current one)
inline void verify_tap_shape(int D, int ntp, int &mp, int &F) { F = (D + 7) / 8; mp = ntp / 8 + 2 * F + 16; }   // (look-ahead room)

// LDS words (float2) of one tile: the padded input span, reused for the partial sums
inline int verify_span(int D, int ntp) { return (kVerOuts - 1) * D + ntp + D + 80; }
inline size_t verify_lds_bytes(int D, int ntp)
{
    const int ns = verify_span(D, ntp);
    int words = ns + ns / (2 * D) + 2;
    if (words < 8 * kVerOuts) words = 8 * kVerOuts;
    return (size_t)words * sizeof(float2) + (size_t)(kVerOuts + 1) * sizeof(float2) + 260 * sizeof(float);
}

// One tile = 128 consecutive outputs t = 127 j - 1 + u of one task (u = 0 is the halo the demodulator needs).
// Summation order of ddc_direct_kernel / the oracle (bit-exact contract): partial l takes the taps j = l, l + 8, ... ascending,
// four fmaf per complex multiply-add; ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)).  Wave l forms partial l of all 128 outputs: its
// taps are wave-uniform (scalar loads), lane i holds the two outputs u = 2 i, 2 i + 1, whose windows overlap by ntp - D samples:
// the sample that meets tap j of the first meets tap j - D of the second, so one LDS read feeds eight multiply-adds.  Lanes are
// 2 D samples apart; one pad word per 2 D samples makes that 2 (2 D + 1) dwords -- an odd multiple of two -- so the 32 lanes of a
// 64-bit read pass hit 32 different bank pairs.  (Taps outside the filter are exact zeros: an accumulator that has seen a
// sample is never -0, and adding +-0 leaves it as it is.)
__device__ __forceinline__ uint32_t ver_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }

// x: the batch's input; tapsv: VerifyParams; dx: [vcap][kVerRows] exact demodulated rows of each task
// DT, NTPT > 0: the decimation and the padded filter length at compile time (50, 672: the 100 Msps bank).  The march is then
// unrolled and every LDS read carries its sample offset AND its pad words as an immediate: with run-time D the pad-word
// bookkeeping (a compare and a select or two per step, on the one scalar unit a CU's four SIMDs share) made this kernel
// SCALAR-bound -- 157 M scalar against 105 M vector wave-instructions per launch, of which 70 M are the multiply-adds
// (profiles/r04_q_c79_pmc_sq.txt).  <0, 0> is the form for every other rate.
template <int DT, int NTPT>
__global__ __launch_bounds__(kVerThreads, 4) void verify_ddc_kernel(VerifyParams p, const float2 *__restrict__ x,
                                                                 const float2 *__restrict__ tapsv, float *__restrict__ dx)
{
    HIP_DYNAMIC_SHARED(float2, lds)
    const int D = DT > 0 ? DT : p.D, ntp = NTPT > 0 ? NTPT : p.ntp;
    const int ns = (kVerOuts - 1) * D + ntp + D + 80;
    int words = ns + ns / (2 * D) + 2;
    if (words < 8 * kVerOuts) words = 8 * kVerOuts;
    float2 *ys = lds + words;                                     // [kVerOuts + 1]
    float *atab = (float *)(ys + kVerOuts + 1);                   // [257]
    // Work items: the entries (task | tile << 24) of ONE CHANNEL's list (the window kernel files a task's tiles under its
    // channel).  A workgroup stays with channel blockIdx.x % nch: the class rows of that channel's taps -- 5.8 KB at 100 Msps,
    // read once per tile -- then stay in the CU's scalar cache; with the channel changing from tile to tile every block of the
    // march waited ~600 cycles for its taps to come from the L2 (0.42 ms per launch where the arithmetic needs 0.12).
    const int wg_per_ch = (int)gridDim.x / p.nch;
    if ((int)blockIdx.x >= wg_per_ch * p.nch) return;
    const int my_c = (int)blockIdx.x % p.nch;
    const unsigned int kstep = (unsigned int)wg_per_ch;
    const uint32_t *tlist = p.tiles + (size_t)my_c * p.tiles_cap;
    unsigned int ntiles = p.tcount[my_c];
    if (ntiles > p.tiles_cap) ntiles = p.tiles_cap;
    if (p.tstart) {                                             // second launch of a batch: the entries listed since the first one
        const unsigned int t0 = p.tstart[my_c] < ntiles ? p.tstart[my_c] : ntiles;
        tlist += t0; ntiles -= t0;
    }
    for (int i = threadIdx.x; i < 257; i += kVerThreads) atab[i] = p.atan_tab[i];
#if defined(__HIP_DEVICE_COMPILE__)
    const int l = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#else
    const int l = (int)threadIdx.x >> 6;
#endif
    const int lane = (int)threadIdx.x & 63;
    // The input span of a workgroup's NEXT tile is requested (kVerPre 8-byte loads per thread, registers) before the march over
    // the current one and written to LDS behind it: staging a tile costs four memory round trips -- more than the march itself.
    // The list entry and the task of an item are fetched a whole item earlier still: entry -> task -> address -> samples were
    // three dependent trips to memory in front of every prefetch.
    float2 pv[kVerPre];
    long long sb_next = 0;
    uint32_t e_cur = 0, e_nxt = 0;                              // entries of the current item and of the one being prefetched
    int w_cur = 0, w_nxt = 0, nx_cur = 0, nx_nxt = 0;           // their tasks: window, exact rows
    auto fetch = [&](unsigned int it) {                         // -> (e_nxt, w_nxt, nx_nxt)
        e_nxt = tlist[it];
        const VerifyTask tk_ = p.tasks[e_nxt & 0xffffffu];
        w_nxt = tk_.w; nx_nxt = tk_.n_exact;
    };
    auto issue = [&]() {                                        // the samples of (e_nxt, w_nxt)
        const unsigned int jt_ = e_nxt >> 24;
        const int k_ = w_nxt / p.nch;
        sb_next = p.first0 + (long long)k_ * p.slot + (long long)(kVerTile * (int)jt_ - 1) * D;
        if (sb_next >= 0 && sb_next + (long long)kVerPre * kVerThreads <= p.x_len) {   // uniform: the span lies inside the stream
            const float2 *xb = x + sb_next;                                 // (scalar base + one 32-bit lane offset: no 64-bit address per load)
#pragma unroll
            for (int r = 0; r < kVerPre; r++) pv[r] = xb[(int)threadIdx.x + r * kVerThreads];
        } else {
#pragma unroll
            for (int r = 0; r < kVerPre; r++) {                             // (static indices: a rolled loop would put pv in scratch memory)
                const long long a = sb_next + (int)threadIdx.x + r * kVerThreads;
                const long long ac = a < 0 ? 0 : (a < p.x_len ? a : p.x_len - 1);   // clamped; the value is replaced by 0 at the store
                pv[r] = x[ac];
            }
        }
    };
    unsigned int item = (unsigned int)blockIdx.x / (unsigned int)p.nch;
    if (item < ntiles) {
        fetch(item); issue();
        e_cur = e_nxt; w_cur = w_nxt; nx_cur = nx_nxt;
        if (item + kstep < ntiles) fetch(item + kstep);
    }
    while (item < ntiles) {
        const uint32_t e = e_cur;
        const int q = (int)(e & 0xffffffu), jt = (int)(e >> 24);
        const int n_exact = nx_cur;
        const int k = w_cur / p.nch, c = w_cur - k * p.nch;
        const int t_first = kVerTile * jt - 1;                     // output index of u = 0
        const long long sb = sb_next;
        __syncthreads();                                           // the previous item's partial sums are consumed
        // ---- the staged input span: sample n of the tile at word n + n / (2 D) ----
#pragma unroll
        for (int r = 0; r < kVerPre; r++) {
            const int n = (int)threadIdx.x + r * kVerThreads;
            const long long a = sb + n;
            if (n < ns) lds[n + (int)ver_mulhi((uint32_t)n, p.inv2d)] = (a >= 0 && a < p.x_len) ? pv[r] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        const unsigned int next = item + kstep;
        if (next < ntiles) {
            issue();                                               // in flight under the march
            e_cur = e_nxt; w_cur = w_nxt; nx_cur = nx_nxt;
            if (next + kstep < ntiles) fetch(next + kstep);
        }
        // ---- the march: step m meets tap l + 8 m of output 2 i and tap l + 8 m - D of output 2 i + 1 ----
        const int lc = ((l - D) % 8 + 8) % 8;                       // class of the second output's tap
        const int sh = (D - l + lc) / 8;                            // its step lag
        const float2 *t0 = tapsv + ((size_t)c * 8 + l) * p.mp + p.F;
        const float2 *t1 = tapsv + ((size_t)c * 8 + lc) * p.mp + p.F - sh;
        const int steps = (ntp / 8 + (D + 7) / 8 + 3) & ~3;         // whole blocks of four (the extra steps meet zero taps)
        float ar0 = 0.f, ai0 = 0.f, ar1 = 0.f, ai1 = 0.f;
        const int lbase = 2 * D * lane + lane;                      // word of the lane's first sample (sample 2 D lane, its pad words)
        int xw = l + l / (2 * D), rem = l % (2 * D);                // wave-uniform: word offset l + 8 m + (its pad words), (l + 8 m) mod 2 D
        const int twoD = 2 * D;
        if (DT > 0) {
            // compile-time geometry: sample offset x = l + 8 m sits x / (2 D) pad words further -- (8 m) / (2 D) of them known
            // here, one more where (8 m) mod (2 D) + l reaches 2 D (only the steps whose remainder is within 7 of 2 D can)
            constexpr int D2 = 2 * (DT > 0 ? DT : 1);
            constexpr int STEPS = ((NTPT > 0 ? NTPT : 8) / 8 + ((DT > 0 ? DT : 1) + 7) / 8 + 3) & ~3;
            const float2 *zb = lds + lbase + l;                     // (l < 8 <= 2 D: no pad word in front of the class offset)
            auto sample = [&](int mm) {
                const int x8 = 8 * mm, r8 = x8 % D2;
                int off = x8 + x8 / D2;
                if (r8 + 7 >= D2) off += (r8 + l >= D2) ? 1 : 0;
                return zb[off];
            };
            float2 vv[4], nv[4];                                    // the block's four samples, and the next block's: read a block ahead
#pragma unroll
            for (int u = 0; u < 4; u++) vv[u] = sample(u);
#pragma unroll
            for (int m = 0; m < STEPS; m += 4) {
                float2 a[4], b[4];                                  // wave-uniform: scalar loads
#pragma unroll
                for (int u = 0; u < 4; u++) { a[u] = t0[m + u]; b[u] = t1[m + u]; }
                if (m + 4 < STEPS) {
#pragma unroll
                    for (int u = 0; u < 4; u++) nv[u] = sample(m + 4 + u);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float2 v = vv[u];
                    ar0 = fmaf(a[u].x, v.x, ar0);
                    ar0 = fmaf(-a[u].y, v.y, ar0);
                    ai0 = fmaf(a[u].x, v.y, ai0);
                    ai0 = fmaf(a[u].y, v.x, ai0);
                    ar1 = fmaf(b[u].x, v.x, ar1);
                    ar1 = fmaf(-b[u].y, v.y, ar1);
                    ai1 = fmaf(b[u].x, v.y, ai1);
                    ai1 = fmaf(b[u].y, v.x, ai1);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) vv[u] = nv[u];
            }
        } else
        for (int m = 0; m < steps; m += 4) {
            float2 a[4], b[4];                                      // wave-uniform: scalar loads
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = t0[m + u]; b[u] = t1[m + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float2 v = lds[lbase + xw];
                xw += 8; rem += 8;
                if (rem >= twoD) { rem -= twoD; xw += 1; }
                if (rem >= twoD) { rem -= twoD; xw += 1; }          // (2 D >= 4: at most two pad words per eight samples)
                ar0 = fmaf(a[u].x, v.x, ar0);
                ar0 = fmaf(-a[u].y, v.y, ar0);
                ai0 = fmaf(a[u].x, v.y, ai0);
                ai0 = fmaf(a[u].y, v.x, ai0);
                ar1 = fmaf(b[u].x, v.x, ar1);
                ar1 = fmaf(-b[u].y, v.y, ar1);
                ai1 = fmaf(b[u].x, v.y, ai1);
                ai1 = fmaf(b[u].y, v.x, ai1);
            }
        }
        __syncthreads();                                           // every wave is done with the samples
        lds[l * kVerOuts + 2 * lane] = make_float2(ar0, ai0);
        lds[lc * kVerOuts + 2 * lane + 1] = make_float2(ar1, ai1);   // (the second output's partial is class lc, not l: the combine below sums by class)
        __syncthreads();
        if (threadIdx.x < kVerOuts) {
            const int u = (int)threadIdx.x;
            float2 pz[8];
#pragma unroll
            for (int j = 0; j < 8; j++) pz[j] = lds[j * kVerOuts + u];
            const float yr = ((pz[0].x + pz[1].x) + (pz[2].x + pz[3].x)) + ((pz[4].x + pz[5].x) + (pz[6].x + pz[7].x));
            const float yi = ((pz[0].y + pz[1].y) + (pz[2].y + pz[3].y)) + ((pz[4].y + pz[5].y) + (pz[6].y + pz[7].y));
            const int t = t_first + u;                              // window-local output index: the rotator restarts per window
            float rr = 1.f, ri = 0.f;
            if (t >= 0) {
                if (p.Q > 0) {
                    const float2 r = p.rot[(size_t)c * p.Q + (t % p.Q)];
                    rr = r.x; ri = r.y;
                } else {
                    double tt = p.rot_step_turns[c] * (double)t;
                    tt -= floor(tt);
                    double sn, co;
                    sincospi(2.0 * tt, &sn, &co);
                    rr = (float)co; ri = (float)sn;
                }
            }
            float2 out;
            out.x = fmaf(-yi, ri, yr * rr);
            out.y = fmaf(yi, rr, yr * ri);
            ys[u] = out;
        }
        __syncthreads();
        if (threadIdx.x >= 1 && threadIdx.x < kVerOuts) {
            const int u = (int)threadIdx.x, t = t_first + u;
            if (t >= 1 && t < n_exact) dx[(size_t)q * (size_t)p.dx_stride + t] = demod_one(atab, p.gain, ys[u], ys[u - 1]);
        }
        item = next;
    }
}

// The same work for the SMALL decimations (8 Msps: D = 4, 56 taps; 20 Msps: D = 10, 136): a tile's march is 12-20 steps there, and
// eight waves that meet at four barriers to share them spend their time on the tile's fixed costs (55 000 tiles per C8 batch:
// 0.63 ms for 1.6 G multiply-adds).  Here ONE WAVE owns a tile: lane i forms all eight partial sums of outputs 2 i and 2 i + 1 in
// registers (class by class, the taps wave-uniform as before, the same LDS read feeding both outputs), combines them, de-rotates;
// only the demodulator's neighbour crosses LDS.  Four waves = four tiles per workgroup pass, two barriers (the waves share
// nothing; the barriers order each wave's own LDS traffic, and keep the CPU emulator's lane-by-lane run faithful).
constexpr int kVerSmallWaves = 4;
inline int verify_small_ns(int D, int ntp) { return (kVerOuts - 2) * D + ntp + 8 * ((D + 7) / 8); }     // samples a tile's march touches
inline size_t verify_small_lds_bytes(int D, int ntp)
{
    const int ns = verify_small_ns(D, ntp);
    return (size_t)kVerSmallWaves * (ns + ns / (2 * D) + 1 + kVerOuts) * sizeof(float2) + 260 * sizeof(float);
}
template <int DT, int NTPT>
__global__ __launch_bounds__(64 * kVerSmallWaves, 4) void verify_ddc_small_kernel(VerifyParams p, const float2 *__restrict__ x,
                                                                             const float2 *__restrict__ tapsv, float *__restrict__ dx)
{
    HIP_DYNAMIC_SHARED(float2, lds)
    constexpr int D = DT, D2 = 2 * DT, F = (DT + 7) / 8, M = NTPT / 8 + F;
    constexpr int ns = (kVerOuts - 2) * D + NTPT + 8 * F;
    constexpr int W = ns + ns / D2 + 1;                           // words of a wave's staged span
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#else
    const int wave = (int)threadIdx.x >> 6;
#endif
    const int lane = (int)threadIdx.x & 63;
    float2 *smp = lds + wave * W;
    float2 *ys = lds + kVerSmallWaves * W + wave * kVerOuts;      // [kVerOuts]: the wave's de-rotated outputs
    float *atab = (float *)(lds + kVerSmallWaves * (W + kVerOuts));
    const int wg_per_ch = (int)gridDim.x / p.nch;
    if ((int)blockIdx.x >= wg_per_ch * p.nch) return;
    const int c = (int)blockIdx.x % p.nch;                        // a workgroup stays with one channel (verify_ddc_kernel)
    const uint32_t *tlist = p.tiles + (size_t)c * p.tiles_cap;
    unsigned int ntiles = p.tcount[c];
    if (ntiles > p.tiles_cap) ntiles = p.tiles_cap;
    if (p.tstart) {                                             // (second launch of a batch: see verify_ddc_kernel)
        const unsigned int t0 = p.tstart[c] < ntiles ? p.tstart[c] : ntiles;
        tlist += t0; ntiles -= t0;
    }
    for (int i = threadIdx.x; i < 257; i += 64 * kVerSmallWaves) atab[i] = p.atan_tab[i];
    const int lbase = D2 * lane + lane;                           // word of the lane's first sample (one pad word per 2 D samples)
    for (unsigned int base = ((unsigned int)blockIdx.x / (unsigned int)p.nch) * kVerSmallWaves; base < ntiles;
         base += (unsigned int)wg_per_ch * kVerSmallWaves) {
        const unsigned int it = base + (unsigned int)wave;
        const bool act = it < ntiles;                             // wave-uniform
        int q = 0, n_exact = 0, t_first = 0;
        if (act) {
            const uint32_t e = tlist[it];
            q = (int)(e & 0xffffffu);
            const int jt = (int)(e >> 24);
            const VerifyTask tk = p.tasks[q];
            n_exact = tk.n_exact;
            t_first = kVerTile * jt - 1;
            const long long sb = p.first0 + (long long)(tk.w / p.nch) * p.slot + (long long)t_first * D;
            constexpr int NR = (ns + 63) / 64;
            float2 pv[NR];
            if (sb >= 0 && sb + 64 * NR <= p.x_len) {             // wave-uniform: the span lies inside the stream -- all loads in flight together
                const float2 *xb = x + sb;
#pragma unroll
                for (int r = 0; r < NR; r++) pv[r] = xb[lane + 64 * r];
            } else {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const long long a = sb + lane + 64 * r;
                    const float2 v = x[a < 0 ? 0 : (a < p.x_len ? a : p.x_len - 1)];
                    pv[r] = (a >= 0 && a < p.x_len) ? v : make_float2(0.f, 0.f);
                }
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int n = lane + 64 * r;
                if (n < ns) smp[n + n / D2] = pv[r];
            }
        }
        __syncthreads();
        float2 y0 = make_float2(0.f, 0.f), y1 = y0;
        if (act) {
            // The eight partial sums of the two outputs, in registers.  The class loop is a REAL loop (unrolled, the scheduler issues
            // all 8 M reads and 16 M tap loads up front and spills both register files); which register a finished sum goes to is a
            // wave-uniform switch.  Sample 2 i D + l + 8 m meets tap l + 8 m of output 2 i and tap l + 8 m - D of output 2 i + 1
            // (class lc, step lag sh).
            float2 pa0 = y0, pa1 = y0, pa2 = y0, pa3 = y0, pa4 = y0, pa5 = y0, pa6 = y0, pa7 = y0;     // output 2 i
            float2 pb0 = y0, pb1 = y0, pb2 = y0, pb3 = y0, pb4 = y0, pb5 = y0, pb6 = y0, pb7 = y0;     // output 2 i + 1
#pragma unroll 1
            for (int l = 0; l < 8; l++) {
                const int lc = (l + 8 * F - D) & 7, sh = (D - l + lc) / 8;
                const float2 *t0 = tapsv + ((size_t)c * 8 + l) * p.mp + p.F;
                const float2 *t1 = tapsv + ((size_t)c * 8 + lc) * p.mp + p.F - sh;
                float ar0 = 0.f, ai0 = 0.f, ar1 = 0.f, ai1 = 0.f;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int xo = l + 8 * m;
                    const float2 v = smp[lbase + xo + xo / D2];
                    const float2 a = t0[m], b = t1[m];
                    ar0 = fmaf(a.x, v.x, ar0);
                    ar0 = fmaf(-a.y, v.y, ar0);
                    ai0 = fmaf(a.x, v.y, ai0);
                    ai0 = fmaf(a.y, v.x, ai0);
                    ar1 = fmaf(b.x, v.x, ar1);
                    ar1 = fmaf(-b.y, v.y, ar1);
                    ai1 = fmaf(b.x, v.y, ai1);
                    ai1 = fmaf(b.y, v.x, ai1);
                }
                const float2 sa = make_float2(ar0, ai0), sb2 = make_float2(ar1, ai1);
                switch (l) { case 0: pa0 = sa; break; case 1: pa1 = sa; break; case 2: pa2 = sa; break; case 3: pa3 = sa; break;
                             case 4: pa4 = sa; break; case 5: pa5 = sa; break; case 6: pa6 = sa; break; default: pa7 = sa; break; }
                switch (lc) { case 0: pb0 = sb2; break; case 1: pb1 = sb2; break; case 2: pb2 = sb2; break; case 3: pb3 = sb2; break;
                              case 4: pb4 = sb2; break; case 5: pb5 = sb2; break; case 6: pb6 = sb2; break; default: pb7 = sb2; break; }
            }
            const float ar0[8] = {pa0.x, pa1.x, pa2.x, pa3.x, pa4.x, pa5.x, pa6.x, pa7.x}, ai0[8] = {pa0.y, pa1.y, pa2.y, pa3.y, pa4.y, pa5.y, pa6.y, pa7.y};
            const float ar1[8] = {pb0.x, pb1.x, pb2.x, pb3.x, pb4.x, pb5.x, pb6.x, pb7.x}, ai1[8] = {pb0.y, pb1.y, pb2.y, pb3.y, pb4.y, pb5.y, pb6.y, pb7.y};
            auto finish = [&](const float *ar, const float *ai, int t) {
                const float yr = ((ar[0] + ar[1]) + (ar[2] + ar[3])) + ((ar[4] + ar[5]) + (ar[6] + ar[7]));
                const float yi = ((ai[0] + ai[1]) + (ai[2] + ai[3])) + ((ai[4] + ai[5]) + (ai[6] + ai[7]));
                float rr = 1.f, ri = 0.f;                         // window-local output index: the rotator restarts per window
                if (t >= 0) {
                    if (p.Q > 0) {
                        const float2 r = p.rot[(size_t)c * p.Q + (t % p.Q)];
                        rr = r.x; ri = r.y;
                    } else {
                        double tt = p.rot_step_turns[c] * (double)t;
                        tt -= floor(tt);
                        double sn, co;
                        sincospi(2.0 * tt, &sn, &co);
                        rr = (float)co; ri = (float)sn;
                    }
                }
                float2 out;
                out.x = fmaf(-yi, ri, yr * rr);
                out.y = fmaf(yi, rr, yr * ri);
                return out;
            };
            y0 = finish(ar0, ai0, t_first + 2 * lane);
            y1 = finish(ar1, ai1, t_first + 2 * lane + 1);
            ys[2 * lane] = y0; ys[2 * lane + 1] = y1;
        }
        __syncthreads();
        if (act) {
            const int ta = t_first + 2 * lane, tb = ta + 1;
            if (lane >= 1 && ta >= 1 && ta < n_exact) dx[(size_t)q * (size_t)p.dx_stride + ta] = demod_one(atab, p.gain, y0, ys[2 * lane - 1]);
            if (tb >= 1 && tb < n_exact) dx[(size_t)q * (size_t)p.dx_stride + tb] = demod_one(atab, p.gain, y1, y0);
        }
    }
}

typedef void (*VerifyDdcKernel)(VerifyParams, const float2 *, const float2 *, float *);
struct VerifyDdcLaunch { VerifyDdcKernel kern; int threads; size_t lds; };
inline VerifyDdcLaunch verify_ddc_pick(int D, int ntp)
{
    if (D == 50 && ntp == 672) return {verify_ddc_kernel<50, 672>, kVerThreads, verify_lds_bytes(D, ntp)};       // 100 Msps
    if (D == 4 && ntp == 56) return {verify_ddc_small_kernel<4, 56>, 64 * kVerSmallWaves, verify_small_lds_bytes(D, ntp)};      // 8 Msps
    if (D == 10 && ntp == 136) return {verify_ddc_small_kernel<10, 136>, 64 * kVerSmallWaves, verify_small_lds_bytes(D, ntp)};  // 20 Msps
    return {verify_ddc_kernel<0, 0>, kVerThreads, verify_lds_bytes(D, ntp)};
}

// The task stream the exact stage's window_kernel reads: dxt[(pseudo-slot * kVerRows + row) * drow + column], task q in
// column q % nch of pseudo-slot q / nch.  Rows [1, n_exact) are the exact ones (row 0 is zeroed by the consumer, policy Q1),
// the rest the polyphase path's -- from the 100-bin bank's tile-blocked copy dcol[tile][80][25] where it exists, else the
// strided column of d -- so the clock recovery can run on behind the exact span.  One workgroup = 64 rows of one pseudo-slot;
// the values cross an LDS tile and leave as whole rows.
struct VerifyFillParams {
    const VerifyTask *tasks; const unsigned int *vcount; int vcap;
    const float *dx;                  // [vcap][kVerRows]
    const float *d; const float *dcol; int drow; long long d_rows;
    int nch, outs_per_slot, rows;     // rows of a task that are filled (min(ddc_out, kVerRows))
    float *dxt;
};
__global__ __launch_bounds__(256) void verify_fill_kernel(VerifyFillParams p)
{
    __shared__ float tile[64 * 81];
    unsigned int ntask = p.vcount[0];
    if (ntask > (unsigned int)p.vcap) ntask = (unsigned int)p.vcap;
    const int nps = (int)((ntask + (unsigned int)p.nch - 1u) / (unsigned int)p.nch);
    const int nrb = (p.rows + 63) / 64;
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    for (int item = blockIdx.x; item < nps * nrb; item += gridDim.x) {
        const int ps = item / nrb, rb = item - ps * nrb;
        const int r = rb * 64 + lane;                              // row this lane fetches
        __syncthreads();
        for (int col = wv; col < p.nch; col += 4) {
            const unsigned int q = (unsigned int)(ps * p.nch + col);
            float v = 0.f;
            if (q < ntask && r < p.rows) {
                const VerifyTask tk = p.tasks[q];
                if (r < tk.n_exact) v = r >= 1 ? p.dx[(size_t)q * kVerRows + r] : 0.f;
                else {
                    const int k = tk.w / p.nch, c = tk.w - k * p.nch;
                    long long g = (long long)k * p.outs_per_slot + r;
                    if (g >= p.d_rows) g = p.d_rows - 1;
                    if (p.dcol) {
                        const unsigned int gq = (unsigned int)g, tq = gq / 25u;
                        v = p.dcol[(size_t)(gq + 25u * (79u * tq + (unsigned int)c))];
                    } else v = p.d[(size_t)g * p.drow + c];
                }
            }
            tile[lane * 81 + col] = v;
        }
        __syncthreads();
        const int nrows = p.rows - rb * 64 < 64 ? p.rows - rb * 64 : 64;
        for (int i = threadIdx.x; i < nrows * p.nch; i += 256) {
            const int rr = i / p.nch, cc = i - rr * p.nch;
            p.dxt[((size_t)ps * kVerRows + rb * 64 + rr) * p.drow + cc] = tile[rr * 81 + cc];
        }
    }
}

}  // namespace btgpu
