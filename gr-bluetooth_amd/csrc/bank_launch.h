// bank_launch.h -- geometry and parameters of one launch of the polyphase bank kernels
// (pfb100.hip.h), shared by the runtime (btgpu.hip) and by the host-emulation tests
// (tests/emu), so that the tile / halo / noise-grid arithmetic that is checked on the CPU is the
// very code that runs in production.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "design.h"
#include "pfb100.hip.h"
#include "pfb100f.hip.h"
#include "pfbm.hip.h"
#include "verify.hip.h"
#include "exact.hip.h"

namespace btgpu {

constexpr int kBankThreads = 256;            // lanes per tile of the non-fused banks
constexpr int kBankThreadsWide = 512;        // fused channel + noise bank: eight waves per tile
constexpr int kBankThreadsF = 320;           // pfb100f_kernel: five waves per run of tiles (every DFT pass in one sweep)
// which kernel runs the fused C79 bank (launch_channel_bank)
enum BankVariant { kBankLegacy = 0, kBankLegacyWide = 1, kBankRun256 = 2, kBankRun320 = 3,
                   kBankRun256d = 6,        // run256 as of profiles/r03_j_* (OPT 7: no lean epilogue, no wave priorities)
                   kBankRun256e = 7,        // the default without the wave priorities (OPT 31)
                   kBankRun256a = 8,        // ... with the priorities read from BTGPU_PFB_DBG (OPT 31 + 256)
                   kBankRun512 = 9,         // round 5: eight waves per tile, channel and squelch branches on different waves, <= 80 VGPRs
                   kBankRun512r = 10 };     // ... in <= 128 VGPRs (two workgroups = 16 waves per CU)
constexpr int kBankNT = 26;                 // channel instants per tile (25 new + 1 halo for the demod)
constexpr int kNoiseNT = 10;                // instants per tile of the stand-alone noise stage 1

// launch geometry of ddc_direct_kernel: T outputs per workgroup, the taps in chunks of JC through 64 KB of LDS
struct LaunchShape {
    int T = 0;        // lanes (= outputs) per workgroup
    int JC = 0;       // taps per LDS chunk
    size_t lds = 0;
};

inline bool pick_shape(int D, int ntp, LaunchShape &s)
{
    const size_t budget = 64 * 1024;
    for (int T : {256, 128, 64}) {
        size_t fixed = (size_t)(T - 1) * D * sizeof(float2);
        if (fixed + 64 * sizeof(float2) > budget) continue;
        long long room = (long long)((budget - fixed) / sizeof(float2));
        int jc = (int)std::min<long long>(ntp, room / 8 * 8);
        if (jc < 8) continue;
        s.T = T;
        s.JC = jc;
        s.lds = (size_t)((T - 1) * D + jc) * sizeof(float2);
        return true;
    }
    return false;
}


struct BankBuffers {                        // device (or emulated) memory
    const float2 *x = nullptr;
    const float2 *taps_ch = nullptr, *twiddle = nullptr, *krot_ch = nullptr, *rho_ch = nullptr;
    const int *binpos_ch = nullptr, *binnat_ch = nullptr;
    const uint16_t *b2map_fused = nullptr, *b2map_fused_wide = nullptr, *b2map_ch = nullptr, *b2map_noise = nullptr, *b2map_f320 = nullptr;
    float *d = nullptr; float *dcol = nullptr; double *ptile = nullptr, *phead = nullptr;
    double *pfine = nullptr;                // small-M F8 bank: |Y|^2 sums per 25 instants (exact stage's burst scan)
    float2 *Ydebug = nullptr; long long ystride = 0;
    const float2 *taps_n = nullptr, *krot_n = nullptr; const int *binpos_n = nullptr;
    float2 *Z = nullptr; long long zstride = 0;
    unsigned long long *prof = nullptr;
    // small-M banks (pfbm.hip.h)
    const float2 *dftw_ch = nullptr, *dftw_n = nullptr;
    int drow = 80;
};

inline size_t bank_lds_bytes(int span_samples, int nt, int nrows, bool chan, bool own_twiddles = false)
{
    const int span = 2 * ((span_samples + 3) / 2);
    const int ysz = chan ? nt * kPfbYst : 0;
    const int asz = kPfbRegion(span, ysz);
    return (size_t)(asz + nrows * kPfbUst + (own_twiddles ? 100 : 0)) * sizeof(float2) + (chan ? 0 : (80 * 4) * sizeof(float2) + 80 * sizeof(int));
}

// Channel bank (+ fused noise stage 1).  L(kernel, grid, threads, lds_bytes, params) performs the launch.
// Returns the number of channel tiles.
template <class Launcher>
inline int launch_channel_bank(const Design &des, const FastPath &fp, bool fuse_noise, const BankBuffers &b,
                               size_t x_len, long long w0, int S, long long G, int nb, Launcher &&L, int variant = kBankLegacy)
{
    const bool wide = variant == kBankLegacyWide;
    const btgpu_design &d = des.d;
    const PfbBank &bk = fp.channel;
    const int nch = d.high_channel - d.low_channel + 1;
    const int ops = des.outs_per_slot;
    constexpr int NT = kBankNT, TT = NT - 1;
    PfbParams p{};
    p.x = b.x; p.x_len = (long long)x_len; p.x0 = w0 + d.first_channel_sample;
    p.D = bk.D; p.T = G;
    p.taps = b.taps_ch; p.twiddle = b.twiddle;
    p.nsel = nch; p.binpos = b.binpos_ch; p.binnat = b.binnat_ch; p.krot = b.krot_ch;
    p.rot_period = bk.rot_period; p.rho = b.rho_ch; p.rho_real = bk.rho_real ? 1 : 0;
    p.ntiles = (int)((G + TT - 1) / TT);
    p.d = b.d; p.dcol = b.dcol; p.ptile = b.ptile; p.phead = b.phead;
    p.tiles_per_block = ops / TT; p.tail = des.tail; p.nb = nb;
    p.gain = des.demod_gain;
    p.kc = demod_constants(p.gain);
    p.Z = b.Ydebug; p.zstride = b.ystride;
    p.prof = b.prof;
    { static const int dbg = getenv("BTGPU_PFB_DBG") ? atoi(getenv("BTGPU_PFB_DBG")) : 0; p.dbg = dbg; }
    if (fuse_noise) {
        const NoiseStage &ns = fp.noise;
        const long long xn0 = w0 + d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R;
        const long long delta0 = (p.x0 - bk.D) - xn0;                      // tile-0 start minus noise origin
        const long long c0 = (delta0 + ns.R - 1) / ns.R;                   // delta0 >= 0
        p.n_taps = b.taps_n; p.n_binpos = b.binpos_n; p.n_krot = b.krot_n; p.n_period = ns.pfb.rot_period;
        p.n_off = (int)(c0 * ns.R - delta0); p.n_u0 = (int)c0; p.pre_tiles = (int)((c0 + 4) / 5);
        p.n_T = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
        p.n_Z = b.Z; p.n_zstride = b.zstride;
        p.b2map = wide ? b.b2map_fused_wide : b.b2map_fused;
        const size_t lds = bank_lds_bytes((250 - 1) + 250 * 4 + 15 * 100, NT, NT + 5, true);
        const int grid = p.ntiles + p.pre_tiles;
        // the run kernels share the march of the staged input between the two banks: the squelch grid must sit on
        // the channel grid (design_fast.cc aligns it for this geometry)
        // (the run kernels address the squelch plane and the tile sums with 32-bit byte offsets from a uniform base)
        const bool fits32 = (unsigned long long)p.nsel * (unsigned long long)p.n_zstride * 8ull < (1ull << 32) &&
                            (unsigned long long)p.nsel * (unsigned long long)p.ntiles * 8ull < (1ull << 32);
        const bool run_ok = p.n_off == 0 && fits32 && variant >= kBankRun256;
        if (!run_ok && variant >= kBankRun256 && getenv("BTGPU_VERBOSE"))
            fprintf(stderr, "launch_channel_bank: squelch grid off the channel grid (n_off %d): round-2 kernel\n", p.n_off);
        if (run_ok) {
            const size_t lds = bank_lds_bytes((250 - 1) + 250 * 4 + 15 * 100, NT, NT + 5, true, true);
            const int nruns = (grid + kBankKT - 1) / kBankKT;
            if (variant == kBankRun320) {
                p.b2map = b.b2map_f320;
                if (bk.real_taps) L(pfb100f_kernel<kBankThreadsF, true, kBankKT>, nruns, kBankThreadsF, lds, p);
                else L(pfb100f_kernel<kBankThreadsF, false, kBankKT>, nruns, kBankThreadsF, lds, p);
            } else if (variant == kBankRun512 && bk.real_taps && p.rho_real) {
                p.b2map = b.b2map_fused_wide;
                L(pfb100f_kernel<kBankThreadsWide, true, 2 * kBankKT, 239>, (grid + 2 * kBankKT - 1) / (2 * kBankKT), kBankThreadsWide, lds, p);
            } else if (variant == kBankRun512r && bk.real_taps && p.rho_real) {
                p.b2map = b.b2map_fused_wide;
                L(pfb100f_kernel<kBankThreadsWide, true, 2 * kBankKT, 255 + 512>, (grid + 2 * kBankKT - 1) / (2 * kBankKT), kBankThreadsWide, lds, p);
            } else if (variant == kBankRun256d && bk.real_taps) L(pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 7>, (grid + 2 * kBankKT - 1) / (2 * kBankKT), kBankThreads, lds, p);   // the default without the lean epilogue
            else if (variant == kBankRun256e && bk.real_taps && p.rho_real) L(pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 31>, (grid + 2 * kBankKT - 1) / (2 * kBankKT), kBankThreads, lds, p);   // the default without the wave priorities
            else if (variant == kBankRun256a && bk.real_taps && p.rho_real) L(pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 31 + 256>, (grid + 2 * kBankKT - 1) / (2 * kBankKT), kBankThreads, lds, p);   // priorities from BTGPU_PFB_DBG (scripts/r03_n_prio.sh)
            else {
                // default: ten tiles per workgroup, march reads eight steps ahead, packed channel MACs, lean epilogue in lockstep
                // pairs, wave priorities staging 2 / march 0 / DFT passes 3 / epilogue 1 (A/B on the device:
                // profiles/r03_h_bank_times.txt, r03_k_bank_times.txt, r03_n_prio_sweep.txt)
                // (BTGPU_BANK_RUNS, diagnostics: the number of workgroups -- tile k of workgroup b is b + k runs)
                static const int runs_env = [] { const char *e = getenv("BTGPU_BANK_RUNS"); return e ? atoi(e) : 0; }();
                const int nr10 = runs_env > 0 ? (runs_env < grid ? runs_env : grid) : (grid + 2 * kBankKT - 1) / (2 * kBankKT);
                // + the lean epilogue where the per-step rotation of every channel is +-1 (100 Msps: always)
                if (bk.real_taps && p.rho_real) L(pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 255>, nr10, kBankThreads, lds, p);
                else if (bk.real_taps) L(pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 7 + 224>, nr10, kBankThreads, lds, p);
                else L(pfb100f_kernel<kBankThreads, false, 2 * kBankKT, 3 + 224>, nr10, kBankThreads, lds, p);
            }
        } else if (wide) {
            if (bk.real_taps) L(pfb100_kernel<7, 1, NT, true, true, kBankThreadsWide, true>, grid, kBankThreadsWide, lds, p);
            else L(pfb100_kernel<7, 1, NT, false, true, kBankThreadsWide, true>, grid, kBankThreadsWide, lds, p);
        } else {
            if (bk.real_taps) L(pfb100_kernel<7, 1, NT, true, true, kBankThreads, true>, grid, kBankThreads, lds, p);
            else L(pfb100_kernel<7, 1, NT, false, true, kBankThreads, true>, grid, kBankThreads, lds, p);
        }
    } else {
        p.b2map = b.b2map_ch;
        const size_t lds = bank_lds_bytes(bk.D * (NT - 1) + bk.Q * 100, NT, NT, true);
        if (bk.real_taps) L(pfb100_kernel<7, 1, NT, true, true, kBankThreads>, p.ntiles, kBankThreads, lds, p);
        else L(pfb100_kernel<7, 1, NT, false, true, kBankThreads>, p.ntiles, kBankThreads, lds, p);
    }
    return p.ntiles;
}

// Stand-alone noise stage 1 as a polyphase bank (staged squelch without the fused kernel)
template <class Launcher>
inline void launch_noise_bank(const Design &des, const FastPath &fp, const BankBuffers &b, size_t x_len,
                              long long w0, int S, Launcher &&L)
{
    const btgpu_design &d = des.d;
    const NoiseStage &ns = fp.noise;
    const PfbBank &bk = ns.pfb;
    const int nch = d.high_channel - d.low_channel + 1;
    constexpr int NT = kNoiseNT;
    const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
    PfbParams p{};
    p.x = b.x; p.x_len = (long long)x_len;
    p.x0 = w0 + d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R;
    p.D = bk.D; p.T = Tn;
    p.taps = b.taps_n; p.twiddle = b.twiddle;
    p.nsel = nch; p.binpos = b.binpos_n; p.krot = b.krot_n; p.rot_period = bk.rot_period;
    p.ntiles = (int)((Tn + NT - 1) / NT);
    p.Z = b.Z; p.zstride = b.zstride;
    p.b2map = b.b2map_noise;
    const size_t lds = bank_lds_bytes(bk.D * (NT - 1) + bk.Q * 100, NT, NT, false);
    L(pfb100_kernel<15, 5, NT, false, false, kBankThreads>, p.ntiles, kBankThreads, lds, p);
}

// Parameters of window_kernel / finish_kernel for a batch of S slots (nb energy blocks per channel, d rows ystride
// apart in the direct-form buffers).  dbg_stop / fin_prio are left 0 / 3.
inline WindowParams make_window_params(const Design &des, int S, int nb, long long ystride, int max_hits, bool want_syms,
                                       const uint64_t *btbb_pcol)
{
    const btgpu_design &d = des.d;
    WindowParams p{};
    p.nch = d.high_channel - d.low_channel + 1; p.S = S; p.outs_per_slot = des.outs_per_slot;
    p.ddc_out = d.ddc_out; p.noise_out = d.noise_out;
    p.blocks_per_window = des.blocks_per_window; p.tail = des.tail; p.nb = nb; p.ystride = ystride;
    p.target_snr = des.cfg.squelch_db;
    p.gain_mu = des.gain_mu; p.mu0 = des.mu0; p.omega_relative_limit = des.omega_relative_limit;
    p.omega0 = des.omega0; p.gain_omega = des.gain_omega; p.omega_mid = des.omega_mid;
    p.mode = des.cfg.mode; p.max_hits = max_hits;
    p.a0_lo = des.ac.a0_lo; p.a0_hi = des.ac.a0_hi;
    p.le = (des.cfg.flags & BTGPU_FLAG_LE) ? 1 : 0; p.low_channel = d.low_channel;
    p.syms = want_syms ? 1 : 0;
    p.btbb = d.correlator == BTGPU_CORRELATOR_BTBB ? 1 : 0;
    p.btbb_pcol = btbb_pcol;
    p.fin_prio = 3;
    p.want_len = 1;
    p.qn_stride = S; p.rows_per_slot = des.outs_per_slot;
    return p;
}

// ---- exact confirmation (verify.hip.h): geometry and parameters shared by the runtime and the emulator ----
struct VerifyBuffers {                                   // device (or emulated) memory of one in-flight batch
    VerifyTask *tasks = nullptr; unsigned int *vcount = nullptr;
    float *dxt = nullptr;
    int vcap = 0;
    uint32_t *bm1 = nullptr, *bm2 = nullptr; int bm_tiles = 0;   // exact rows' bitmaps [bm_tiles][kExBmWords] (presence / the first run's uncovered hits)
};
constexpr int kVerCountWords = 8;                       // vcount: 0 tasks, 1 pairs marked, 2 turned away, 3 busy windows, 4 / 5 pairs computed by the first / second launch of exact_rows_kernel
inline int verify_capacity(int S, int nch) { return (int)std::min<long long>((long long)S * nch, 32768); }
// the small-M bank in its F8 form (C8) also leaves 25-instant sums: finer than its 250-instant tiles
inline bool verify_has_fine(const Design &des, const FastPath &fp, int drow)
{
    const int nch = des.d.high_channel - des.d.low_channel + 1;
    return fp.channel.natural && fp.channel.M == 8 && nch == 8 && drow == 8 && pfbm_tile(8) + 1 <= kPfbmThreads && pfbm_tile(8) % 25 == 0;
}
inline int verify_rows(const Design &des) { return des.d.ddc_out < kVerRows ? des.d.ddc_out : kVerRows; }
constexpr int kVerGridFill = 1024;
// tile length (outputs) of the |Y|^2 tile sums the polyphase banks leave behind
inline int verify_tile_outs(const FastPath &fp, bool small) { return small ? pfbm_tile(fp.channel.M) : kBankNT - 1; }

// presence_kernel's and the first run's parameters (mode 1: presence + uncovered hits, 2: uncovered hits only)
// (ptile / ntiles / tile_outs: the tile sums presence reads -- the bank's own tiles, or the F8 bank's 25-instant sums)
inline void set_verify_flagging(WindowParams &p, const Design &des, const FastPath &fp, bool small, int mode,
                                const double *ptile, int ntiles, const VerifyBuffers &vb, bool headers, int tile_outs = 0)
{
    p.verify = mode;
    p.ptile = ptile; p.ptile_stride = ntiles; p.tile_outs = tile_outs > 0 ? tile_outs : verify_tile_outs(fp, small);
    p.tiles_per_slot = des.outs_per_slot / p.tile_outs;
    p.vtasks = vb.tasks; p.vcount = vb.vcount; p.vcap = vb.vcap; p.bm1 = vb.bm1; p.bm2 = vb.bm2; p.bm_tiles = vb.bm_tiles;
    const int ntm = (2 * kDetectSyms + 16 + p.tile_outs - 1) / p.tile_outs;
    if (ntm + kBurstFront > 64) p.verify = 2;            // (presence stages <= 64 tiles per channel)
    // The scan's statistic is the energy of W tiles, ~50 us.  Its threshold is 2.0 x the MEAN noise block; the scan knows the
    // span's QUIETEST aligned block, which lies z sigma under the mean: sigma of a 25-output tile sum is 0.28 of its mean (the
    // 2 Msps stream of a ~1 MHz filter holds ~12.6 independent values per 25), z the expected minimum of n normal draws.
    p.burst_w = std::max(1, 100 / p.tile_outs);
    {
        const float nblocks = (float)std::max(2, (ntm + kBurstFront) / p.burst_w);
        const float sigma = 0.28f * std::sqrt(25.0f / (float)(p.tile_outs * p.burst_w));
        const float beta = 1.0f - (0.5f + 0.45f * std::log(nblocks)) * sigma;
        const char *ea = std::getenv("BTGPU_BURST_A");                  // (diagnostics: the scan's threshold in mean noise blocks)
        const float A = ea ? (float)std::atof(ea) : 2.0f;
        p.burst_abs = A / beta;
        // the single-tile stand-in (kernels.hip.h): the minimum of the span's tiles lies z(tiles) sigma_tile under the mean; it counts
        // 1.6 x, so that noise alone never prefers it to the block estimate
        const float sigma1 = 0.28f * std::sqrt(25.0f / (float)p.tile_outs);
        const float beta1 = std::max(0.2f, 1.0f - (0.5f + 0.45f * std::log((float)(ntm + kBurstFront))) * sigma1);
        p.burst_abs1 = A * (float)p.burst_w * 1.6f / beta1;
        // ... and of the channel's quietest tile of the whole batch (n tiles): the lower tail of a sum of ~12.6 values per 25 outputs --
        // 0.48 of the mean for the least of 100, 0.40 of 1e3, 0.33 of 1e4, 0.27 of 1e5 (tabulated, log-linear between)
        {
            const float lg = std::log10(std::max(100.0f, (float)ntiles) / 100.0f);
            const float beta2 = std::max(0.2f, 1.0f - (1.0f - (0.48f - 0.065f * lg)) * std::sqrt(std::min(1.0f, 25.0f / (float)p.tile_outs)));
            p.burst_abs2 = A * (float)p.burst_w * 1.6f / beta2;
        }
        p.chan_floor = nullptr;                          // (set by the caller once channel_floor_kernel has run)
        const char *eh = std::getenv("BTGPU_BURST_A_HOT");
        p.burst_abs_hot = std::max(1.0f, (eh ? (float)std::atof(eh) : 3.0f) / A);      // 3.0 x the mean noise beside a hot neighbour
        p.burst_hot = 100.0f / A;                        // a neighbour 20 dB over the noise
    }
    p.span_extra = headers ? 58 : 0;                     // 54 header symbols + the 4-symbol trailer
}
static_assert(kExactSlotRows == kExSlotRows, "design.h and kernels.hip.h agree on the slot");
// exact_rows_kernel's parameters for one bitmap of the batch (tapsA: exact_pack_taps of the direct-form channel bank)
inline ExactParams make_exact_params(const Design &des, size_t x_len, long long w0, long long G, const float *tapsA, const float2 *rot,
                                     const float *atan_tab, const uint32_t *bitmap, int bm_tiles, float *d, int drow, float *dcol, unsigned int *stat)
{
    const btgpu_design &dd = des.d;
    ExactParams e{};
    e.x_len = (long long)x_len; e.first0 = w0 + dd.first_channel_sample; e.G = G;
    e.tapsA = tapsA; e.rot = rot; e.Qr = des.channel.rot_period; e.atan_tab = atan_tab; e.gain = des.demod_gain;
    e.bitmap = bitmap; e.ntiles = bm_tiles; e.stat = stat; e.d = d; e.drow = drow; e.dcol = dcol;
    e.ydbg = nullptr; e.ystride = 0; e.nch = dd.high_channel - dd.low_channel + 1;
    return e;
}
inline VerifyFillParams make_verify_fill_params(const Design &des, const float *d_stream, const float *dcol, int drow, long long G,
                                                const VerifyBuffers &vb)
{
    VerifyFillParams f{};
    f.tasks = vb.tasks; f.vcount = vb.vcount; f.vcap = vb.vcap;
    f.d = d_stream; f.dcol = dcol; f.drow = drow; f.d_rows = G;
    f.nch = des.d.high_channel - des.d.low_channel + 1; f.outs_per_slot = des.outs_per_slot; f.rows = verify_rows(des);
    f.dxt = vb.dxt;
    return f;
}
// the exact stage's window_kernel<LAY, true> launch: tasks as lanes, nch per pseudo-slot
inline WindowParams make_verify_window_params(const WindowParams &first, const VerifyBuffers &vb)
{
    WindowParams p = first;
    p.verify = 0;
    p.S = (vb.vcap + first.nch - 1) / first.nch;          // pseudo-slots (workgroups beyond the task count leave at once)
    p.rows_per_slot = kVerRows;
    return p;
}

// ---- the rates below 100 Msps: pfbm_kernel (M = fs / 1 MHz bins) ----
// Squelch stage 1 inside the 8-bin channel bank (C8): both banks are 8-bin natural-order banks over the same input, stage 1 at a hop
// that is a multiple of the channel bank's, and the channel bank's tiles cover every stage-1 instant the squelch needs.
// first stage-1 instant of channel tile 0: the two banks start at different samples (stage 1 reads from the window's start, the channel
// bank from its first detection sample), so tile k is paired with the stage-1 instants whose span overlaps its own
inline int pfbm_fuse_first(const Design &des, const FastPath &fp)
{
    const NoiseStage &ns = fp.noise;
    const long long d0 = (long long)des.d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R - des.d.first_channel_sample + fp.channel.D;
    if (d0 >= 0) return 0;
    return (int)((-d0 + ns.R / 2) / ns.R);
}
inline bool pfbm_fuse_noise(const Design &des, const FastPath &fp, int S, long long G, int drow)
{
    const btgpu_design &d = des.d;
    const PfbBank &bk = fp.channel, &nk = fp.noise.pfb;
    const NoiseStage &ns = fp.noise;
    const int nch = d.high_channel - d.low_channel + 1;
    if (!(bk.available && ns.available && nk.available)) return false;
    if (!(bk.natural && bk.M == 8 && nch == 8 && drow == 8 && bk.Q == 7 && !bk.real_taps)) return false;
    if (!(nk.natural && nk.M == 8 && nk.Q == 15 && nk.D == ns.R)) return false;
    const int TT = pfbm_tile(8);
    if (TT + 1 > kPfbmThreads || (TT * bk.D) % nk.D != 0) return false;
    const int npt = TT * bk.D / nk.D;
    if (npt > kPfbmThreads || npt * 9 * 2 > TT * 8) return false;                 // its branch outputs borrow the angle tile
    const long long ntiles = (G + TT - 1) / TT;
    const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
    return ntiles * npt + pfbm_fuse_first(des, fp) >= Tn;
}
template <class Launcher>
inline int launch_channel_bank_m(const Design &des, const FastPath &fp, const BankBuffers &b, size_t x_len,
                                 long long w0, long long G, Launcher &&L, int fuse_S = 0)
{
    const btgpu_design &d = des.d;
    const PfbBank &bk = fp.channel;
    const int nch = d.high_channel - d.low_channel + 1;
    PfbmParams p{};
    p.x = b.x; p.x_len = (long long)x_len; p.x0 = w0 + d.first_channel_sample;
    p.M = bk.M; p.D = bk.D; p.Q = bk.Q; p.T = G; p.TT = pfbm_tile(bk.M);
    p.taps = b.taps_ch; p.dftw = b.dftw_ch; p.nsel = nch;
    p.rho = b.rho_ch; p.rho_real = bk.rho_real ? 1 : 0;
    p.krot = b.krot_ch; p.rot_period = bk.rot_period;
    p.ntiles = (int)((G + p.TT - 1) / p.TT);
    p.d = b.d; p.drow = b.drow; p.ptile = b.ptile; p.phead = b.phead;
    p.pfine = nullptr;
    p.tiles_per_block = des.outs_per_slot / p.TT; p.tail = des.tail;
    p.gain = des.demod_gain;
    p.Z = b.Ydebug; p.zstride = b.ystride;
    size_t lds = pfbm_lds_bytes(bk.M, bk.D, bk.Q, nch, true);
    const bool f8 = bk.natural && bk.M == 8 && nch == 8 && b.drow == 8 && p.TT + 1 <= kPfbmThreads;
    if (f8 && p.TT % 25 == 0) p.pfine = b.pfine;
    if (fuse_S > 0 && pfbm_fuse_noise(des, fp, fuse_S, G, b.drow)) {
        // stage 1 from the same staged span: noise instant n reads x[xn0 + R n + 8 q + p]; a tile's channel span starts at
        // gs = x0 + D (TT tile - 1), its noise instants at xn0 + R npt tile = xn0 + D TT tile
        const NoiseStage &ns = fp.noise;
        const PfbBank &nk = ns.pfb;
        const long long xn0 = w0 + d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R;
        const int npt = p.TT * bk.D / nk.D;
        const int n_first = pfbm_fuse_first(des, fp);
        const int delta = (int)(xn0 - p.x0) + bk.D + n_first * nk.D;               // first noise sample of a tile relative to gs
        const int c_len = bk.D * p.TT + bk.Q * bk.M, n_len = nk.D * (npt - 1) + nk.Q * nk.M;
        const int lo = delta < 0 ? delta : 0;
        const int hi = delta + n_len > c_len ? delta + n_len : c_len;
        p.n_on = 1; p.n_per_tile = npt; p.n_D = nk.D; p.n_rot_period = nk.rot_period; p.n_first = n_first;
        p.stage_lo = lo; p.stage_len = hi - lo; p.n_ofs = delta - lo;
        p.n_T = (long long)ns.outs * (fuse_S - 1) + ns.nw + ns.L3 - 1;
        p.n_taps = b.taps_n; p.n_krot = b.krot_n; p.n_Z = b.Z; p.n_zstride = b.zstride;
        lds = pfbm_lds_bytes(bk.M, bk.D, bk.Q, nch, true, p.stage_len);
        L(pfbm_kernel<false, true, 8, 7, true, true>, p.ntiles, kPfbmThreads, lds, p);
        if (n_first > 0) {
            // the stage-1 instants in front of tile 0's: one or two tiles of the stand-alone form
            PfbmParams q{};
            q.x = b.x; q.x_len = (long long)x_len; q.x0 = xn0;
            q.M = nk.M; q.D = nk.D; q.Q = nk.Q; q.TT = pfbm_tile(nk.M);
            q.T = n_first < p.n_T ? n_first : p.n_T;
            q.taps = b.taps_n; q.dftw = b.dftw_n; q.nsel = nch;
            q.krot = b.krot_n; q.rot_period = nk.rot_period;
            q.ntiles = (int)((q.T + q.TT - 1) / q.TT);
            q.Z = b.Z; q.zstride = b.zstride;
            L(pfbm_kernel<false, false, 8, 15, true>, q.ntiles, kPfbmThreads, pfbm_lds_bytes(nk.M, nk.D, nk.Q, nch, false), q);
        }
        return p.ntiles;
    }
    if (f8 && bk.Q == 7 && !bk.real_taps) L(pfbm_kernel<false, true, 8, 7, true>, p.ntiles, kPfbmThreads, lds, p);          // C8: half-MHz grid
    else if (f8 && bk.Q == 7) L(pfbm_kernel<true, true, 8, 7, true>, p.ntiles, kPfbmThreads, lds, p);
    else if (bk.M == 8 && bk.Q == 7 && !bk.real_taps) L(pfbm_kernel<false, true, 8, 7>, p.ntiles, kPfbmThreads, lds, p);
    else if (bk.M == 20 && bk.Q == 7 && bk.real_taps) L(pfbm_kernel<true, true, 20, 7>, p.ntiles, kPfbmThreads, lds, p);
    else if (bk.real_taps) L(pfbm_kernel<true, true>, p.ntiles, kPfbmThreads, lds, p);
    else L(pfbm_kernel<false, true>, p.ntiles, kPfbmThreads, lds, p);
    return p.ntiles;
}

template <class Launcher>
inline void launch_noise_bank_m(const Design &des, const FastPath &fp, const BankBuffers &b, size_t x_len,
                                long long w0, int S, Launcher &&L)
{
    const btgpu_design &d = des.d;
    const NoiseStage &ns = fp.noise;
    const PfbBank &bk = ns.pfb;
    const int nch = d.high_channel - d.low_channel + 1;
    const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
    PfbmParams p{};
    p.x = b.x; p.x_len = (long long)x_len;
    p.x0 = w0 + d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R;
    p.M = bk.M; p.D = bk.D; p.Q = bk.Q; p.T = Tn; p.TT = pfbm_tile(bk.M);
    p.taps = b.taps_n; p.dftw = b.dftw_n; p.nsel = nch;
    p.krot = b.krot_n; p.rot_period = bk.rot_period;
    p.ntiles = (int)((Tn + p.TT - 1) / p.TT);
    p.Z = b.Z; p.zstride = b.zstride;
    const size_t lds = pfbm_lds_bytes(bk.M, bk.D, bk.Q, nch, false);
    if (bk.natural && bk.M == 8 && nch == 8 && bk.Q == 15) L(pfbm_kernel<false, false, 8, 15, true>, p.ntiles, kPfbmThreads, lds, p);
    else if (bk.M == 8 && bk.Q == 15) L(pfbm_kernel<false, false, 8, 15>, p.ntiles, kPfbmThreads, lds, p);
    else L(pfbm_kernel<false, false>, p.ntiles, kPfbmThreads, lds, p);
}

}  // namespace btgpu
