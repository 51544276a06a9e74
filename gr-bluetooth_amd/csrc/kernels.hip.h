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

namespace btgpu {

struct DeviceHit {            // 32 bytes, written by the window kernel
    uint32_t slot;            // batch-relative slot index
    int32_t  channel_idx;     // index into the visible channel range
    int32_t  offset;
    uint32_t lap;
    int32_t  ac_errors;
    int32_t  kind;
    double   snr;
};

struct FinishRec {            // M&M state of a window that reported hits, handed to finish_kernel
    int32_t  w;               // window index k * nch + c
    uint32_t ii;
    int32_t  oo;
    float    mu, omega, last;
};

// ------------------------------------------------------------------------------------
// K1: direct-form decimating complex band-pass FIR bank (channel bank and noise bank).
//   y[c][g] = ( sum_j taps[c][j] * x[first + g*D + j] ) * rot[c][g]
// Summation order (bit-exact contract with the oracle): 8 partial sums, partial l takes
// j = l, l+8, l+16, ... ascending, four fmaf per complex MAC; combined as
// ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)).
// One lane = one output instant, CPB channels per workgroup; the input span of the
// workgroup's tile is staged through LDS once per tap chunk and shared by the CPB
// channels; taps are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------
template <int CPB>
__global__ __launch_bounds__(256) void ddc_direct_kernel(
    const float2 *__restrict__ x, long long x_len, long long first, int D, int ntp, int JC,
    const float2 *__restrict__ taps, const float2 *__restrict__ rot, int Q,
    const double *__restrict__ rot_step_turns,   // used when Q == 0
    float2 *__restrict__ Y, long long G, long long ystride, int nch)
{
    extern __shared__ float2 tile[];
    const int T = blockDim.x;
    const long long g0 = (long long)blockIdx.x * T;
    const int c0 = blockIdx.y * CPB;
    const int o = threadIdx.x;

    float ar[CPB][8], ai[CPB][8];
#pragma unroll
    for (int cc = 0; cc < CPB; cc++)
#pragma unroll
        for (int l = 0; l < 8; l++) { ar[cc][l] = 0.f; ai[cc][l] = 0.f; }

    const float2 *tp[CPB];
#pragma unroll
    for (int cc = 0; cc < CPB; cc++) {
        int c = c0 + cc < nch ? c0 + cc : nch - 1;
        tp[cc] = taps + (size_t)c * ntp;
    }

    for (int j0 = 0; j0 < ntp; j0 += JC) {
        const int jc = (ntp - j0) < JC ? (ntp - j0) : JC;
        const int need = (T - 1) * D + jc;
        const long long base = first + g0 * D + j0;
        __syncthreads();
        for (int s = o; s < need; s += T) {
            long long a = base + s;
            float2 v = make_float2(0.f, 0.f);
            if (a >= 0 && a < x_len) v = x[a];
            tile[s] = v;
        }
        __syncthreads();
        const float2 *px = tile + o * D;
        for (int j = 0; j < jc; j += 8) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const float2 v = px[j + l];
#pragma unroll
                for (int cc = 0; cc < CPB; cc++) {
                    const float2 t = tp[cc][j0 + j + l];
                    ar[cc][l] = fmaf(t.x, v.x, ar[cc][l]);
                    ar[cc][l] = fmaf(-t.y, v.y, ar[cc][l]);
                    ai[cc][l] = fmaf(t.x, v.y, ai[cc][l]);
                    ai[cc][l] = fmaf(t.y, v.x, ai[cc][l]);
                }
            }
        }
    }

    const long long g = g0 + o;
    if (g >= G) return;
#pragma unroll
    for (int cc = 0; cc < CPB; cc++) {
        const int c = c0 + cc;
        if (c >= nch) break;
        float yr = ((ar[cc][0] + ar[cc][1]) + (ar[cc][2] + ar[cc][3])) +
                   ((ar[cc][4] + ar[cc][5]) + (ar[cc][6] + ar[cc][7]));
        float yi = ((ai[cc][0] + ai[cc][1]) + (ai[cc][2] + ai[cc][3])) +
                   ((ai[cc][4] + ai[cc][5]) + (ai[cc][6] + ai[cc][7]));
        float rr, ri;
        if (Q > 0) {
            const float2 r = rot[(size_t)c * Q + (int)(g % Q)];
            rr = r.x; ri = r.y;
        } else {
            double t = rot_step_turns[c] * (double)g;
            t -= floor(t);
            double s, co;
            sincospi(2.0 * t, &s, &co);
            rr = (float)co; ri = (float)s;
        }
        float2 out;
        out.x = fmaf(-yi, ri, yr * rr);
        out.y = fmaf(yi, rr, yr * ri);
        Y[(size_t)c * ystride + g] = out;
    }
}

// ------------------------------------------------------------------------------------
// gr::fast_atan2f [EXT GNU Radio 3.7], table in LDS/global (257 floats)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2f_dev(const float *__restrict__ tab, float y, float x)
{
    const float TAN_MAP_RES = 0.003921569f;
    const float TAN_MAP_SIZE = 255.0f;
    float y_abs = fabsf(y), x_abs = fabsf(x), z, base_angle, angle;
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    if (y_abs < x_abs) z = __fdiv_rn(y_abs, x_abs); else z = __fdiv_rn(x_abs, y_abs);
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

__device__ __forceinline__ float demod_one(const float *__restrict__ atab, float gain, float2 a, float2 b)
{
    // a * conj(b)  (multi_block.cc:165-166)
    float pr = fmaf(a.y, b.y, a.x * b.x);
    float pi = fmaf(a.y, b.x, -(a.x * b.y));
    return gain * fast_atan2f_dev(atab, pi, pr);
}

// ------------------------------------------------------------------------------------
// K2: demodulate the channel stream and reduce |Y|^2 per slot-block.
// One workgroup per (block b of `bs` outputs, channel).  DEMOD=false: energy only
// (noise bank).  Sums are float mag^2 accumulated in double like multi_block.cc:206-218.
// ------------------------------------------------------------------------------------
template <bool DEMOD>
__global__ __launch_bounds__(256) void demod_energy_kernel(
    const float2 *__restrict__ Y, long long G, long long ystride, int bs, int tail,
    const float *__restrict__ atan_tab, float gain, float *__restrict__ d,
    double *__restrict__ P, double *__restrict__ Pt, int nb, int nch, float *__restrict__ d2,
    long long d2stride)
{
    __shared__ float atab[257];
    __shared__ double red[2][4];
    const int c = blockIdx.y;
    const int b = blockIdx.x;
    if (DEMOD) {
        for (int i = threadIdx.x; i < 257; i += blockDim.x) atab[i] = atan_tab[i];
        __syncthreads();
    }
    const float2 *y = Y + (size_t)c * ystride;
    const long long gb = (long long)b * bs;
    double s_full = 0.0, s_tail = 0.0;
    for (int i = threadIdx.x; i < bs; i += blockDim.x) {
        const long long g = gb + i;
        if (g >= G) break;
        const float2 v = y[g];
        const float m = (v.x * v.x) + (v.y * v.y);
        s_full += (double)m;
        if (i < tail) s_tail += (double)m;
        if (DEMOD) {
            float dv = 0.f;
            if (g > 0) dv = demod_one(atab, gain, v, y[g - 1]);
            d[(size_t)g * 80 + c] = dv;
            if (d2) d2[(size_t)c * d2stride + g] = dv;
        }
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
    int outs_per_slot;          // grid points per slot (slot / decim)
    int ddc_out, noise_out;
    int blocks_per_window, tail;
    int nb;                     // number of energy blocks per channel
    long long ystride;
    int dstride;                // row stride of d in floats (nch padded to a multiple of 4)
    double target_snr;
    float gain_mu, mu0, omega_relative_limit, omega0, gain_omega, omega_mid;
    int mode;                   // BTGPU_MODE_*
    int max_hits;
    uint64_t a0_lo; uint32_t a0_hi;
};

__device__ __forceinline__ int popc5min(uint32_t v, uint32_t a, uint32_t b)
{
    int da = __popc(v ^ a), db = __popc(v ^ b);
    return da < db ? da : db;
}

constexpr int kWinThreads = 128;     // >= 79 visible channels; lane = channel
constexpr int kWinRows = 64;         // demod rows staged per chunk

// One workgroup per slot k, one lane per channel c (nch <= 79 < 128).  The demodulated stream
// is time-major [g][nch], so the rows a slot's windows need are shared by all its lanes: they
// are staged through LDS in chunks of kWinRows rows with fully coalesced loads, and the
// strictly sequential M&M recursion of each lane then runs out of LDS instead of paying a
// global-memory round trip per symbol.  Lanes drift apart by a few samples only (omega is
// clipped to 2 +- 0.005), so a chunk starts at the minimum input index over the live lanes.
// Lanes stop after the 625-offset search range unless they committed a hit; those continue
// to the end of the window to obtain `len` (the handlers' symbol count, nsym = len - offset).
__global__ __launch_bounds__(kWinThreads) void window_kernel(
    WindowParams p, const float *__restrict__ d, long long d_rows, const double *__restrict__ P,
    const double *__restrict__ Pt, const double *__restrict__ Qn,
    const float *__restrict__ mmse_g, const uint64_t *__restrict__ ac_lo_g,
    const uint32_t *__restrict__ ac_hi_g,
    double *__restrict__ e_on_out, double *__restrict__ e_off_out, double *__restrict__ snr_out,
    int *__restrict__ win_len, DeviceHit *__restrict__ hits, unsigned int *__restrict__ hit_count,
    FinishRec *__restrict__ fin, unsigned int *__restrict__ fin_count)
{
    __shared__ float mmse[129 * 8];
    __shared__ uint64_t ac_lo[3 * 256];
    __shared__ uint32_t ac_hi[3 * 256];
    __shared__ __attribute__((aligned(16))) float tile[kWinRows * 80];
    __shared__ int s_min;
    for (int i = threadIdx.x; i < 129 * 8; i += blockDim.x) mmse[i] = mmse_g[i];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) { ac_lo[i] = ac_lo_g[i]; ac_hi[i] = ac_hi_g[i]; }

    const int k = blockIdx.x;
    const int c = threadIdx.x;
    const int nch = p.nch;
    const long long w = (long long)k * nch + c;
    bool active = c < nch;
    double snr = 0.0;

    // ---- squelch: multi_block::channel_samples energy + check_snr (multi_block.cc:206-293) ----
    if (active) {
        double e_on = 0.0;
        for (int j = 0; j < p.blocks_per_window; j++) e_on += P[(size_t)c * p.nb + k + j];
        if (p.tail > 0) e_on += Pt[(size_t)c * p.nb + k + p.blocks_per_window];
        e_on /= (double)p.ddc_out;
        const double e_off = Qn[(size_t)c * p.S + k] / (double)p.noise_out;
        snr = 10.0 * log10(e_on / e_off);
        e_on_out[w] = e_on; e_off_out[w] = e_off; snr_out[w] = snr;
        win_len[w] = -1;
        if (!(snr >= p.target_snr)) active = false;
    }

    // ---- M&M (multi_block.cc:128-155), windowed reset ----
    const int demod_n = p.ddc_out - 1;
    const unsigned int ni = (unsigned int)(demod_n - 8);
    const long long row0 = (long long)k * p.outs_per_slot;       // global row of window index 0
    float mu = p.mu0, omega = p.omega0, last = 0.f;
    unsigned int ii = 0;
    int oo = 0;
    uint64_t wlo = 0; uint32_t whi = 0;      // correlator window: bit i = symbol (s-67+i)
    int pending = -1, resume = 0, nhits = 0;
    uint32_t pend_lap = 0; int pend_err = 0;
    bool searching = true;

    for (;;) {
        // chunk base = min input index over the live lanes
        if (threadIdx.x == 0) s_min = 0x7fffffff;
        __syncthreads();
        if (active) atomicMin(&s_min, (int)ii);
        __syncthreads();
        const int base = s_min;
        if (base == 0x7fffffff) break;                           // no live lane left (uniform)
        {
            // rows [base, base + kWinRows) of this slot's windows are one contiguous range of the
            // time-major stream (row stride dstride = 80 floats): straight 16-byte copy, all
            // loads of a lane issued before the first LDS store.
            constexpr int kVec = kWinRows * 80 / 4;              // float4 per chunk
            constexpr int kPer = (kVec + kWinThreads - 1) / kWinThreads;
            const long long r_first = row0 + base;
            const float4 *src = (const float4 *)(d + (size_t)r_first * 80);
            long long rows_ok = d_rows - r_first;                // rows that exist in the buffer
            const long long win_ok = (long long)p.ddc_out - base;
            if (win_ok < rows_ok) rows_ok = win_ok;
            const int vec_ok = rows_ok <= 0 ? 0 : (rows_ok >= kWinRows ? kVec : (int)rows_ok * 20);
            float4 v[kPer];
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int i = threadIdx.x + j * kWinThreads;
                v[j] = (i < vec_ok) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int i = threadIdx.x + j * kWinThreads;
                if (i < kVec) ((float4 *)tile)[i] = v[j];
            }
            if (base == 0 && threadIdx.x < 80) tile[threadIdx.x] = 0.f;   // policy Q1: demod_out[0] = 0
        }
        __syncthreads();
        const unsigned int lim = (unsigned int)(base + kWinRows - 8);
        while (active && ii <= lim) {
            if (!(ii < ni && oo < demod_n)) {                    // input exhausted: window done
                if (nhits > 0) win_len[w] = oo;
                active = false;
                break;
            }
            // interpolate: sum_q T[imu][7-q] * in[ii+q], q ascending
            int imu = (int)rintf(mu * 128.0f);
            imu = imu < 0 ? 0 : (imu > 128 ? 128 : imu);
            const float *t = &mmse[imu * 8];
            const float *in = &tile[(ii - (unsigned int)base) * 80 + c];
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) acc = fmaf(t[7 - q], in[q * 80], acc);
            const float out = acc;
            const float s_last = (last < 0) ? -1.0f : 1.0f;
            const float s_out = (out < 0) ? -1.0f : 1.0f;
            const float mm_val = s_last * out - s_out * last;
            last = out;
            omega = omega + (p.gain_omega * mm_val);
            {
                const float xx = omega - p.omega_mid;
                float x1 = fabsf(xx + p.omega_relative_limit);
                const float x2 = fabsf(xx - p.omega_relative_limit);
                x1 -= x2;
                omega = p.omega_mid + 0.5f * x1;
            }
            mu = mu + (omega + (p.gain_mu * mm_val));
            const float fl = floorf(mu);
            ii += (unsigned int)(int)fl;
            mu = mu - fl;

            // ---- slicer + streaming access-code search ----
            const uint32_t sym = (out < 0) ? 0u : 1u;
            const int s = oo;
            oo++;
            if (searching) {
                wlo = (wlo >> 1) | ((uint64_t)(whi & 1u) << 63);
                whi = (whi >> 1) | (sym << 3);
                if (pending >= 0) {          // one more symbol exists: pending < len - 68
                    const unsigned int slot_h = atomicAdd(hit_count, 1u);
                    if (slot_h < (unsigned int)p.max_hits) {
                        DeviceHit h;
                        h.slot = (uint32_t)k; h.channel_idx = c; h.offset = pending;
                        h.lap = pend_lap; h.ac_errors = pend_err; h.kind = 0; h.snr = snr;
                        hits[slot_h] = h;
                    }
                    nhits++;
                    resume = pending + 68;
                    pending = -1;
                    if (p.mode == 0) searching = false;      // multi_LAP: first hit only
                }
                const int cpos = s - 67;
                if (searching && cpos >= resume && cpos < 625) {
                    const uint32_t pre = (uint32_t)wlo & 0x1f;
                    const uint32_t bar = ((uint32_t)(wlo >> 61) | (whi << 3)) & 0x7f;
                    const int gate = popc5min(pre, 0x0a, 0x15) + popc5min(bar, 0x27, 0x58);
                    if (gate <= 2) {
                        const uint32_t lap = (uint32_t)(wlo >> 38) & 0xffffff;
                        const uint64_t elo = p.a0_lo ^ ac_lo[lap & 0xff] ^ ac_lo[256 + ((lap >> 8) & 0xff)] ^
                                             ac_lo[512 + (lap >> 16)];
                        const uint32_t ehi = p.a0_hi ^ ac_hi[lap & 0xff] ^ ac_hi[256 + ((lap >> 8) & 0xff)] ^
                                             ac_hi[512 + (lap >> 16)];
                        const int err = __popcll(elo ^ wlo) + __popc((ehi ^ whi) & 0xf);
                        if (err < 7) { pending = cpos; pend_lap = lap; pend_err = err; }
                    }
                }
                if (cpos >= 625 && pending < 0) searching = false;
            }
            if (!searching && nhits == 0) active = false;    // search range exhausted without a hit
            if (!searching && nhits > 0 && p.mode != 0) {
                // multi_sniffer: the handlers need len = symbols in the whole window.  Hand the
                // M&M state to finish_kernel (dense waves of hit windows) instead of keeping this
                // slot's workgroup alive for 5x longer with one or two live lanes.
                const unsigned int f = atomicAdd(fin_count, 1u);
                FinishRec r;
                r.w = (int32_t)w; r.ii = ii; r.oo = oo; r.mu = mu; r.omega = omega; r.last = last;
                fin[f] = r;
                active = false;
            }
        }
    }
}

// Continue the M&M recursion of the windows that reported hits to the end of their window and
// store len.  One lane per window; each lane stages its own column of the time-major stream
// (row stride 80 floats) into a private LDS slab, kFinRows rows at a time, all loads of a chunk
// in flight together.  No cross-lane data => no barriers.
constexpr int kFinRows = 32;
__global__ __launch_bounds__(64) void finish_kernel(
    WindowParams p, const float *__restrict__ d2, long long d2stride, long long d_rows,
    const float *__restrict__ mmse_g, const FinishRec *__restrict__ fin,
    const unsigned int *__restrict__ fin_count, int *__restrict__ win_len)
{
    __shared__ float mmse[129 * 8];
    __shared__ float slab[64 * (2 * kFinRows + 1)];
    const unsigned int n = *fin_count;
    if (blockIdx.x * blockDim.x >= n) return;                    // uniform: nothing for this workgroup
    for (int i = threadIdx.x; i < 129 * 8; i += blockDim.x) mmse[i] = mmse_g[i];
    __syncthreads();
    const unsigned int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const FinishRec r = fin[f];
    const int k = r.w / p.nch, c = r.w - k * p.nch;
    const int demod_n = p.ddc_out - 1;
    const unsigned int ni = (unsigned int)(demod_n - 8);
    const long long row0 = (long long)k * p.outs_per_slot;
    const float *col = d2 + (size_t)c * d2stride + row0;         // this window's demod samples, contiguous
    const unsigned int nvalid = (unsigned int)((d_rows - row0) < p.ddc_out ? (d_rows - row0) : p.ddc_out);
    float mu = r.mu, omega = r.omega, last = r.last;
    unsigned int ii = r.ii;
    int oo = r.oo;
    // private ring of 2*kFinRows samples: rows [lo, lo + 2*kFinRows) live at my[row & (2*kFinRows-1)]
    float *my = slab + threadIdx.x * (2 * kFinRows + 1);
    constexpr unsigned int RING = 2 * kFinRows, MASK = RING - 1;
    unsigned int hi = ii;                                        // rows [.., hi) are resident
    {
        float v[RING];
#pragma unroll
        for (unsigned int j = 0; j < RING; j++) { const unsigned int idx = hi + j; v[j] = idx < nvalid ? col[idx] : 0.f; }
#pragma unroll
        for (unsigned int j = 0; j < RING; j++) my[(hi + j) & MASK] = v[j];
        hi += RING;
    }
    while (ii < ni && oo < demod_n) {
        // issue the loads of the next kFinRows rows now; they land while the steps below run
        float v[kFinRows];
#pragma unroll
        for (int j = 0; j < kFinRows; j++) { const unsigned int idx = hi + j; v[j] = idx < nvalid ? col[idx] : 0.f; }
        // consume every step whose 8-tap window lies inside the resident rows [.., hi)
        while (ii + 8 <= hi && ii < ni && oo < demod_n) {
            int imu = (int)rintf(mu * 128.0f);
            imu = imu < 0 ? 0 : (imu > 128 ? 128 : imu);
            const float *t = &mmse[imu * 8];
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) acc = fmaf(t[7 - q], my[(ii + q) & MASK], acc);
            const float out = acc;
            const float s_last = (last < 0) ? -1.0f : 1.0f;
            const float s_out = (out < 0) ? -1.0f : 1.0f;
            const float mm_val = s_last * out - s_out * last;
            last = out;
            omega = omega + (p.gain_omega * mm_val);
            {
                const float xx = omega - p.omega_mid;
                float x1 = fabsf(xx + p.omega_relative_limit);
                const float x2 = fabsf(xx - p.omega_relative_limit);
                x1 -= x2;
                omega = p.omega_mid + 0.5f * x1;
            }
            mu = mu + (omega + (p.gain_mu * mm_val));
            const float fl = floorf(mu);
            ii += (unsigned int)(int)fl;
            mu = mu - fl;
            oo++;
        }
        // here ii + 8 > hi (or the window is done): the ring slots of rows [hi-RING, hi-RING+kFinRows)
        // are all below ii and can take rows [hi, hi + kFinRows)
#pragma unroll
        for (int j = 0; j < kFinRows; j++) my[(hi + j) & MASK] = v[j];
        hi += kFinRows;
    }
    win_len[r.w] = oo;
}

}  // namespace btgpu