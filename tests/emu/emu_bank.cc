// Host emulation of the HIP kernels (TEST INFRASTRUCTURE; see hip/hip_runtime.h): the same source hipcc compiles
// for gfx950 -- polyphase banks (pfb100.hip.h, pfbm.hip.h), direct-form banks, energy / demod kernels, noise stage 2,
// window / finish / nsym-patch kernels, the symbol-stream correlator and the header sweep (kernels.hip.h) -- launched
// through the same bank_launch.h the runtime uses, run lane by lane on the CPU (fibers, real barriers, wave shuffles
// and ballots through an exchange buffer), results handed to the Python tests, which compare them with the oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <vector>


#include "bank_launch.h"

using namespace btgpu;

extern "C" {

// iq: interleaved complex64, x_len samples; window 0 of the batch starts at sample w0 (margin in front).
// mode: BTGPU_MODE_*; fuse: 1 = channel bank + fused noise stage 1, 0 = channel bank alone, 2 = also run the
// stand-alone noise bank into Z.
// Outputs (caller-allocated): d [G][80] float, P/Pt [nch][nb] double (block sums as block_sum_kernel forms them),
// Z [nch][zstride] complex64 (noise stage 1), Y [nch][ystride] complex64 (de-rotated channel output, optional).
// sizes[]: G, nb, nch, zstride, ystride, Tn (filled in).  Returns 0 or a negative error.
static std::vector<float> *g_keep_dcol = nullptr;      // emu_front_run: keep the tile-blocked copy the bank kernel wrote
static std::vector<double> *g_keep_ptile = nullptr;    // ... and the |Y|^2 tile sums (exact confirmation: burst energy)
static int g_keep_ntiles = 0;
static int g_fuse_m = 1, g_last_fused_m = 0;           // emu_set_fuse_m: squelch stage 1 inside the 8-bin channel bank (as the runtime decides) or apart
static std::vector<double> *g_keep_pfine = nullptr;    // small-M F8 bank: the 25-instant sums

int emu_bank_run(double fs, double fc, int mode, const float *iq, long long x_len, long long w0, int S, int fuse,
                 float *d_out, double *P_out, double *Pt_out, float *Z_out, float *Y_out, long long *sizes)
{
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = 10.0; cfg.mode = mode;
    static Design des; static FastPath fp;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    rc = make_fast_path(des, fp);
    if (rc) return rc;
    if (!fp.channel.available || !fp.noise.available || !fp.noise.pfb.available) return BTGPU_EUNSUPPORTED;
    const btgpu_design &d = des.d;
    const int nch = d.high_channel - d.low_channel + 1;
    const int ops = des.outs_per_slot;
    const long long G = (long long)ops * (S - 1) + d.ddc_out;
    const int nb = (int)((G + ops - 1) / ops);
    const NoiseStage &ns = fp.noise;
    const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
    const long long zstride = (Tn + 10 + 63) / 64 * 64, ystride = (G + 63) / 64 * 64;
    sizes[0] = G; sizes[1] = nb; sizes[2] = nch; sizes[3] = zstride; sizes[4] = ystride; sizes[5] = Tn;
    if (!d_out) return 0;                                  // size query

    const std::vector<uint16_t> mf = make_dft_pass2_map(kBankNT + 5, kBankThreads, 2);
    const std::vector<uint16_t> mc = make_dft_pass2_map(kBankNT, kBankThreads, 2);
    const std::vector<uint16_t> mn = make_dft_pass2_map(kNoiseNT, kBankThreads, 2);
    const std::vector<uint16_t> mw = make_dft_pass2_map(kBankNT + 5, kBankThreadsWide, 1);
    const std::vector<uint16_t> m5 = make_dft_pass2_map(kBankNT + 5, kBankThreadsF, 1);
    if (mf.empty() || mc.empty() || mn.empty() || mw.empty() || m5.empty()) return BTGPU_EUNSUPPORTED;
    const int ntiles_max = (int)((G + 24) / 25);
    std::vector<double> ptile((size_t)nch * ntiles_max, 0.0), phead((size_t)nch * ntiles_max, 0.0);
    // x must be readable as float4 at even sample offsets: keep a 16-byte aligned copy with slack
    std::vector<float4> xbuf((size_t)x_len / 2 + 16);
    std::memcpy(xbuf.data(), iq, (size_t)x_len * sizeof(float2));
    BankBuffers b;
    b.x = (const float2 *)xbuf.data();
    // the 100-bin kernel reads its taps branch-major in 16-byte pieces: keep the packed tables 16-byte aligned
    const std::vector<float> tch = pack_branch_major(fp.channel), tn = pack_branch_major(ns.pfb);
    std::vector<float4> tch4((tch.size() + 3) / 4 + 1), tn4((tn.size() + 3) / 4 + 1);
    std::memcpy(tch4.data(), tch.data(), tch.size() * sizeof(float));
    std::memcpy(tn4.data(), tn.data(), tn.size() * sizeof(float));
    b.taps_ch = (const float2 *)tch4.data(); b.twiddle = (const float2 *)fp.channel.twiddle.data();
    b.krot_ch = (const float2 *)fp.channel.krot.data(); b.rho_ch = (const float2 *)fp.channel.rho.data();
    b.binpos_ch = fp.channel.binpos.data(); b.binnat_ch = fp.channel.binnat.data();
    b.b2map_fused = mf.data(); b.b2map_fused_wide = mw.data(); b.b2map_ch = mc.data(); b.b2map_noise = mn.data(); b.b2map_f320 = m5.data();
    b.d = d_out; b.ptile = ptile.data(); b.phead = phead.data();
    std::vector<float> dcol((size_t)ntiles_max * 80 * 25, -77.f);
    b.dcol = dcol.data();
    b.Ydebug = (float2 *)Y_out; b.ystride = ystride;
    b.taps_n = (const float2 *)tn4.data(); b.krot_n = (const float2 *)ns.pfb.krot.data();
    b.binpos_n = ns.pfb.binpos.data();
    b.Z = (float2 *)Z_out; b.zstride = zstride;
    auto L = [&](void (*kern)(PfbParams), int grid, int threads, size_t lds, const PfbParams &p) {
        if (lds > sizeof emu::dyn_lds) { std::fprintf(stderr, "emu: LDS %zu\n", lds); std::abort(); }
        std::memset(emu::dyn_lds, 0xff, sizeof emu::dyn_lds);      // NaN pattern: reads of unwritten LDS show up
        emu::launch(dim3((unsigned)grid), dim3((unsigned)threads), [&]() { kern(p); });
    };
    // fuse: 1 = fused, the product's default (pfb100f_kernel, four waves, ten tiles per workgroup); 4 / 6 / 7 / 8 = its other
    // forms (five tiles and fenced epilogue / packed channel MACs / ten tiles / five waves); 5 = the round-2 kernel,
    // four waves per tile; 3 = the round-2 kernel, eight waves
    const int variant = fuse == 3 ? kBankLegacyWide : fuse == 5 ? kBankLegacy : fuse == 4 ? kBankRun256a : fuse == 6 ? kBankRun256d :
                        fuse == 7 ? kBankRun256e : fuse == 8 ? kBankRun320 : fuse == 9 ? kBankRun512 : fuse == 10 ? kBankRun512r : kBankRun256;
    const int ntiles = launch_channel_bank(des, fp, fuse == 1 || fuse >= 3, b, (size_t)x_len, w0, S, G, nb, L, variant);
    if (fuse == 2) launch_noise_bank(des, fp, b, (size_t)x_len, w0, S, L);
    // the tile-blocked copy finish_kernel reads must hold the very same angles: dcol[tile][c][r] == d[25 tile + r][c]
    for (long long g = 0; g < G; g++)
        for (int c = 0; c < nch; c++) {
            const float a = d_out[(size_t)g * 80 + c], bcol = dcol[((size_t)(g / 25) * 80 + c) * 25 + g % 25];
            if (std::memcmp(&a, &bcol, sizeof a) != 0) { std::fprintf(stderr, "emu: dcol != d at g %lld c %d\n", g, c); return -100; }
        }
    if (g_keep_dcol) *g_keep_dcol = dcol;
    if (g_keep_ptile) { *g_keep_ptile = ptile; g_keep_ntiles = ntiles; }
    // block sums exactly as block_sum_kernel orders them (per block: tiles ascending)
    const int tpb = ops / 25, tail_tiles = des.tail / 25;
    for (int c = 0; c < nch; c++)
        for (int bi = 0; bi < nb; bi++) {
            double s = 0.0, h = 0.0;
            for (int k = 0; k < tpb; k++) {
                const int t = bi * tpb + k;
                if (t < ntiles) {
                    const double v = ptile[(size_t)c * ntiles + t];
                    s += v;
                    if (k < tail_tiles) h += v;
                    else if (k == tail_tiles) h += phead[(size_t)c * ntiles + t];
                }
            }
            P_out[(size_t)c * nb + bi] = s;
            Pt_out[(size_t)c * nb + bi] = h;
        }
    return 0;
}

// The small-M banks (pfbm.hip.h): channel bank -> d [G][drow] + block sums, noise bank -> Z.  Same outputs and sizes[]
// as emu_bank_run, plus sizes[6] = drow.
int emu_bank_m_run(double fs, double fc, int mode, const float *iq, long long x_len, long long w0, int S,
                   float *d_out, double *P_out, double *Pt_out, float *Z_out, float *Y_out, long long *sizes)
{
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = 10.0; cfg.mode = mode;
    static Design des; static FastPath fp;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    rc = make_fast_path(des, fp);
    if (rc) return rc;
    if (!fp.channel.available || fp.channel.M >= kPfbM || !fp.noise.available || !fp.noise.pfb.available) return BTGPU_EUNSUPPORTED;
    const btgpu_design &d = des.d;
    const int nch = d.high_channel - d.low_channel + 1;
    const int ops = des.outs_per_slot;
    const long long G = (long long)ops * (S - 1) + d.ddc_out;
    const int nb = (int)((G + ops - 1) / ops);
    const NoiseStage &ns = fp.noise;
    const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
    const long long zstride = (Tn + 10 + 63) / 64 * 64, ystride = (G + 63) / 64 * 64;
    const int drow = win_drow(nch);
    sizes[0] = G; sizes[1] = nb; sizes[2] = nch; sizes[3] = zstride; sizes[4] = ystride; sizes[5] = Tn; sizes[6] = drow;
    if (!d_out) return 0;
    const int TTm = pfbm_tile(fp.channel.M);
    const int ntiles_max = (int)((G + TTm - 1) / TTm);
    std::vector<double> ptile((size_t)nch * ntiles_max, 0.0), phead((size_t)nch * ntiles_max, 0.0);
    std::vector<float4> xbuf((size_t)x_len / 2 + 16);
    std::memcpy(xbuf.data(), iq, (size_t)x_len * sizeof(float2));
    BankBuffers b;
    b.x = (const float2 *)xbuf.data();
    b.taps_ch = (const float2 *)fp.channel.taps.data(); b.dftw_ch = (const float2 *)fp.channel.dftw.data();
    b.krot_ch = (const float2 *)fp.channel.krot.data(); b.rho_ch = (const float2 *)fp.channel.rho.data();
    b.d = d_out; b.drow = drow; b.ptile = ptile.data(); b.phead = phead.data();
    std::vector<double> pfine;
    if (verify_has_fine(des, fp, drow)) { pfine.assign((size_t)nch * ntiles_max * (TTm / 25), -1.0); b.pfine = pfine.data(); }
    b.Ydebug = (float2 *)Y_out; b.ystride = ystride;
    b.taps_n = (const float2 *)ns.pfb.taps.data(); b.dftw_n = (const float2 *)ns.pfb.dftw.data();
    b.krot_n = (const float2 *)ns.pfb.krot.data();
    b.Z = (float2 *)Z_out; b.zstride = zstride;
    auto L = [&](void (*kern)(PfbmParams), int grid, int threads, size_t lds, const PfbmParams &p) {
        if (lds > sizeof emu::dyn_lds) { std::fprintf(stderr, "emu: LDS %zu\n", lds); std::abort(); }
        std::memset(emu::dyn_lds, 0xff, sizeof emu::dyn_lds);
        emu::launch(dim3((unsigned)grid), dim3((unsigned)threads), [&]() { kern(p); });
    };
    // (as the runtime: stage 1 inside the 8-bin channel bank where the geometry allows it; emu_set_fuse_m(0): the two launches)
    const bool fused_m = g_fuse_m && pfbm_fuse_noise(des, fp, S, G, drow);
    const int ntiles = launch_channel_bank_m(des, fp, b, (size_t)x_len, w0, G, L, fused_m ? S : 0);
    if (!fused_m) launch_noise_bank_m(des, fp, b, (size_t)x_len, w0, S, L);
    g_last_fused_m = fused_m ? 1 : 0;
    if (g_keep_ptile) { *g_keep_ptile = ptile; g_keep_ntiles = ntiles; }
    if (g_keep_pfine) *g_keep_pfine = pfine;
    const int tpb = ops / TTm, tail_tiles = des.tail / TTm;
    for (int c = 0; c < nch; c++)
        for (int bi = 0; bi < nb; bi++) {
            double s = 0.0, h = 0.0;
            for (int k = 0; k < tpb; k++) {
                const int t = bi * tpb + k;
                if (t < ntiles) {
                    const double v = ptile[(size_t)c * ntiles + t];
                    s += v;
                    if (k < tail_tiles) h += v;
                    else if (k == tail_tiles) h += phead[(size_t)c * ntiles + t];
                }
            }
            P_out[(size_t)c * nb + bi] = s;
            Pt_out[(size_t)c * nb + bi] = h;
        }
    return 0;
}

// staged-squelch stage 2 constants (host design) for the tests' numpy restatement of noise_stage2_kernel
int emu_stage2_design(double fs, double fc, int mode, float *h3, double *w, int *ints /* outs, nw, L3, R, pad, Jm */)
{
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = 10.0; cfg.mode = mode;
    static Design des; static FastPath fp;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    rc = make_fast_path(des, fp);
    if (rc || !fp.noise.available) return BTGPU_EUNSUPPORTED;
    const NoiseStage &ns = fp.noise;
    ints[0] = ns.outs; ints[1] = ns.nw; ints[2] = ns.L3; ints[3] = ns.R; ints[4] = ns.pad; ints[5] = ns.Jm;
    if (h3) std::memcpy(h3, ns.h3.data(), ns.h3.size() * sizeof(float));
    if (w) std::memcpy(w, ns.weights.data(), ns.weights.size() * sizeof(double));
    return 0;
}

// noise_stage2_kernel alone, on a given stage-1 plane Z [nch][zstride] (the tests compare with a numpy restatement)
int emu_stage2_run(int outs, int nw, int L3, const float *h3, const double *w, const float *Z, long long zstride, int nch, int S,
                   double *Qn)
{
    const size_t lds2 = s2_lds_bytes(outs, nw, L3);
    if (lds2 > sizeof emu::dyn_lds) return BTGPU_EUNSUPPORTED;
    std::memset(emu::dyn_lds, 0xff, sizeof emu::dyn_lds);            // NaNs: a read of anything not staged shows
    emu::launch(dim3((unsigned)((S + kS2Slots - 1) / kS2Slots), (unsigned)nch), dim3(256), [&]() {
        noise_stage2_kernel((const float2 *)Z, zstride, outs, nw, L3, h3, w, Qn, S, BlockSumArgs{});
    });
    return 0;
}

// the three forms of the polynomial arctangent (pfb100.hip.h): demod_poly (canonicalises -0), demod_poly_pz (arguments never
// -0), demod_poly_pz2 (two in lockstep) -- the tests want them bit-identical wherever the arguments carry no -0
int emu_demod_variants(int n, float gain, const float *pr, const float *pi, float *a_ref, float *a_pz, float *a_pz2)
{
    const DemodConst k = demod_constants(gain);
    for (int i = 0; i < n; i++) {
        a_ref[i] = demod_poly(k, pr[i], pi[i]);
        a_pz[i] = demod_poly_pz(k, k.c[5], pr[i], pi[i]);
    }
    for (int i = 0; i + 1 < n; i += 2) demod_poly_pz2(k, k.c[5], pr[i], pi[i], pr[i + 1], pi[i + 1], a_pz2[i], a_pz2[i + 1]);
    if (n & 1) a_pz2[n - 1] = a_pz[n - 1];
    return 0;
}

// the pass-2 lane map, for the bank-conflict check of the tests
int emu_b2map(int rows, int lanes, int sweeps, uint16_t *out)
{
    const std::vector<uint16_t> m = make_dft_pass2_map(rows, lanes, sweeps);
    if (m.empty()) return -1;
    std::memcpy(out, m.data(), m.size() * sizeof(uint16_t));
    return (int)m.size();
}

}  // extern "C"


// window_kernel + finish_kernel + nsym_patch_kernel over the demodulated stream d (time-major, `drow` floats per row;
// dcol = the 100-bin bank's tile-blocked copy or null), the channel block energies P / Pt and the noise energies Qn
struct StudyCapture { std::vector<float> d; std::vector<double> snr; long long G = 0; int drow = 0, nch = 0, S = 0;
                      std::vector<VerifyTask> tasks; };
static uint32_t *g_sym_out = nullptr; static uint8_t *g_hdr_out = nullptr;     // set by emu_front_direct_headers_run
static bool g_rows_only = false;                        // emu_verify_check: the demodulated rows are all it reads -- the exact noise bank is left out (E_off = 0: every window passes)
static StudyCapture *g_study = nullptr;                 // emu_margin_study: keep the demodulated stream and the squelch SNRs

// exact confirmation in the emulated front end: what the runtime's tail stream does (btgpu.hip process_batch)
struct VerifyEmu {
    const float2 *x = nullptr; long long x_len = 0; const FastPath *fp = nullptr; bool small = false;
    const double *ptile = nullptr; int ntiles = 0; int mode = 1; int tile_outs = 0;
    unsigned int counts[4] = {0, 0, 0, 0};              // out: tasks, tiles, turned away
};
static int g_verify_mode = 1;                            // emu_set_verify: 0 off, 1 hits + burst energy, 2 hits only
static unsigned int g_verify_counts[kVerCountWords] = {0};  // of the last emulated front end: tasks, pairs marked, turned away, busy windows, pairs computed (two launches)
extern "C" void emu_set_verify(int mode) { g_verify_mode = mode; }
static int g_exact_payload = 0;                          // emu_set_exact_payload: BTGPU_FLAG_EXACT_PAYLOAD (with symbols)
extern "C" void emu_set_exact_payload(int on) { g_exact_payload = on; }
static unsigned int g_long_counts[4] = {0, 0, 0, 0};
extern "C" void emu_long_counts(unsigned int *out) { std::memcpy(out, g_long_counts, sizeof g_long_counts); }
extern "C" void emu_set_fuse_m(int on) { g_fuse_m = on; }
extern "C" int emu_last_fused_m(void) { return g_last_fused_m; }
extern "C" void emu_verify_counts(unsigned int *out) { std::memcpy(out, g_verify_counts, sizeof g_verify_counts); }
static std::vector<VerifyTask> g_verify_tasks;           // of the last emulated front end: windows with exact rows (n_exact: covered rows from the window's first)
static std::vector<VerifyTask> g_second_run;             // ... and the windows its second run took
static std::vector<uint32_t> g_exact_bitmap; static int g_exact_tiles = 0;   // bm1 | bm2 of the last emulated front end
extern "C" int emu_exact_bitmap(uint32_t *out, int cap_words)
{
    const int n = (int)std::min<size_t>(g_exact_bitmap.size(), (size_t)cap_words);
    if (out) std::memcpy(out, g_exact_bitmap.data(), (size_t)n * 4);
    return g_exact_tiles;
}
extern "C" int emu_verify_tasks(int *w_out, int *rows_out, int cap)
{
    const int n = (int)std::min<size_t>(g_verify_tasks.size(), (size_t)cap);
    for (int i = 0; i < n; i++) { w_out[i] = g_verify_tasks[i].w; rows_out[i] = g_verify_tasks[i].n_exact; }
    return n;
}

static int run_detect(const Design &des, int S, int nb, int nch, int drow, long long G, const float *d, const float *dcol_p,
                      const double *P, const double *Pt, const double *Qn, long long *rec_out, double *snr_out, int cap,
                      uint32_t *sym_out = nullptr, uint8_t *hdr_out = nullptr, VerifyEmu *ve = nullptr)
{
    // sym_out [cap][kSymWords]: packed symbols of each record's window (BTGPU_FLAG_SYMBOLS); hdr_out [cap][132]: the
    // header sweep (64 UAPs, 64 types, fec13_ok as 4 bytes) of header_sweep_kernel (BTGPU_FLAG_HEADERS)
    const bool want_syms = sym_out != nullptr;
    const int max_hits = 1 << 16;
    std::vector<uint64_t> pcol(des.ac.btbb_pcol, des.ac.btbb_pcol + 24);
    WindowParams p = make_window_params(des, S, nb, 0, max_hits, want_syms, pcol.data());
    const size_t W = (size_t)S * nch;
    std::vector<double> e_on(W), e_off(W), snr(W);
    std::vector<int> win_len(W, -1), win_fin(W, -1);
    std::vector<DeviceHit> hits((size_t)max_hits);
    std::vector<FinishRec> fin(W);
    unsigned int counts[2] = {0, 0};
    std::vector<uint32_t> winbits((size_t)((S + 2) / 3 + 1) * kBitWords * kWinThreads, 0u);
    std::vector<uint32_t> symbits(want_syms ? W * kSymWords : 1, 0u);
    // exact rows: bitmaps, the second run's task list and task stream
    VerifyBuffers vb;
    std::vector<VerifyTask> vtasks; std::vector<float4> dxt4; std::vector<uint32_t> bm;
    unsigned int vcount[kVerCountWords] = {0};
    const bool verify = ve && ve->mode > 0 && exact_rows_available(des) && exact_rows_pick(des.d.decimation);
    if (verify) {
        vb.vcap = verify_capacity(S, nch);
        vtasks.resize((size_t)vb.vcap);
        const int nps = (vb.vcap + nch - 1) / nch;
        dxt4.assign(((size_t)nps * kVerRows * drow + 3) / 4 + 16, make_float4(-66.f, -66.f, -66.f, -66.f));
        vb.bm_tiles = exact_ntiles(G);
        bm.assign((size_t)2 * vb.bm_tiles * kExBmWords, 0u);
        vb.bm1 = bm.data(); vb.bm2 = bm.data() + (size_t)vb.bm_tiles * kExBmWords;
        vb.tasks = vtasks.data(); vb.vcount = vcount; vb.dxt = (float *)dxt4.data();
        set_verify_flagging(p, des, *ve->fp, ve->small, ve->mode, ve->ptile, ve->ntiles, vb, want_syms, ve->tile_outs);
    }
    std::vector<float> chan_floor(81, 3.0e38f);
    // the runtime's choreography (btgpu.hip process_batch): channel floor, presence, exact rows in place (bm1), the window kernel, the
    // exact rows under its uncovered hits (bm2), fill, the second run
    std::vector<float> tapsA;
    auto run_exact = [&](const uint32_t *bitmap, unsigned int *stat) {
        const int D = des.d.decimation;
        if (tapsA.empty()) { tapsA.resize(exact_taps_floats(nch, D)); exact_pack_taps(des.channel.taps.data(), nch, des.channel.ntp, D, tapsA.data()); }
        const ExactParams ep = make_exact_params(des, (size_t)ve->x_len, 0, G, tapsA.data(), (const float2 *)des.channel.rot.data(), des.atan_tab,
                                                 bitmap, vb.bm_tiles, const_cast<float *>(d), drow, const_cast<float *>(dcol_p), stat);
        const ExactRowsKernel kern = exact_rows_pick(D);
        if (exact_lds_bytes(D) > sizeof emu::dyn_lds) { std::fprintf(stderr, "emu: LDS %zu\n", exact_lds_bytes(D)); std::abort(); }
        bool any = false;
        for (size_t i = 0; i < (size_t)vb.bm_tiles * kExBmWords; i++) any = any || bitmap[i];
        if (!any) return;
        std::memset(emu::dyn_lds, 0xff, sizeof emu::dyn_lds);
        // (ONE emulated workgroup strides over all the tiles, as the kernel allows: the fibers are set up once)
        emu::launch(dim3(1), dim3(kExThreads), [&]() { kern(ep, ve->x); });
    };
    auto launch_window = [&](auto lay) {
        using LAY = decltype(lay);
        if (verify && getenv("EMU_EXACT_ALL")) {                   // BTGPU_FLAG_EXACT_ALL: no selection, every row exact
            emu::launch(dim3((unsigned)((vb.bm_tiles * kExBmWords + 255) / 256)), dim3(256), [&]() { exact_mark_all_kernel(vb.bm1, vb.bm_tiles, nch); });
            run_exact(vb.bm1, &vcount[4]);
        } else if (verify && p.verify == 1) {
            const int full_tiles = (G % p.tile_outs) ? p.ptile_stride - 1 : p.ptile_stride;
            emu::launch(dim3(2u, (unsigned)nch), dim3(256), [&]() { channel_floor_kernel(p.ptile, p.ptile_stride, std::max(1, full_tiles), chan_floor.data()); });
            p.chan_floor = getenv("EMU_NO_FLOOR") ? nullptr : chan_floor.data();
            if (getenv("EMU_DBG_FLOOR")) for (int c = 0; c < nch; c++) std::fprintf(stderr, "floor ch %d %.4g\n", c, chan_floor[c]);
            emu::launch(dim3((unsigned)((S + LAY::kSlots - 1) / LAY::kSlots)), dim3(kWinThreads), [&]() { presence_kernel<LAY>(p); });
            run_exact(vb.bm1, &vcount[4]);
        }
        emu::launch(dim3((unsigned)((S + LAY::kSlots - 1) / LAY::kSlots)), dim3(kWinThreads), [&]() {
            window_kernel<LAY>(p, d, G, P, Pt, Qn, des.mmse, &des.ac.byte_lo[0][0], &des.ac.byte_hi[0][0],
                               e_on.data(), e_off.data(), snr.data(), win_len.data(), hits.data(), &counts[0], fin.data(),
                               &counts[1], &des.le.hdr[0][0], des.le.whiten16, des.le.index_of_channel, win_fin.data(),
                               want_syms ? symbits.data() : (uint32_t *)nullptr, winbits.data());
        });
        if (!verify) return;
        // ---- the second run (runtime: tail stream) ----
        run_exact(vb.bm2, &vcount[5]);
        if (!vcount[0]) return;
        const VerifyFillParams fpz = make_verify_fill_params(des, d, dcol_p, drow, G, vb);
        emu::launch(dim3(16), dim3(256), [&]() { verify_fill_kernel(fpz); });
        const WindowParams pv = make_verify_window_params(p, vb);
        std::vector<uint32_t> winbits_v((size_t)((pv.S + LAY::kSlots - 1) / LAY::kSlots + 1) * kBitWords * kWinThreads, 0u);
        emu::launch(dim3((unsigned)((pv.S + LAY::kSlots - 1) / LAY::kSlots)), dim3(kWinThreads), [&]() {
            window_kernel<LAY, true>(pv, (const float *)dxt4.data(), (long long)pv.S * kVerRows, P, Pt, Qn, des.mmse, &des.ac.byte_lo[0][0],
                                     &des.ac.byte_hi[0][0], e_on.data(), e_off.data(), snr.data(), win_len.data(), hits.data(), &counts[0],
                                     fin.data(), &counts[1], &des.le.hdr[0][0], des.le.whiten16, des.le.index_of_channel, win_fin.data(),
                                     want_syms ? symbits.data() : (uint32_t *)nullptr, winbits_v.data());
        });
    };
    if (drow == 80) launch_window(WinLayout<3, 96, 20>{});
    else if (drow == 40) launch_window(WinLayout<6, 40, 10>{});
    else if (drow == 20) launch_window(WinLayout<12, 20, 5>{});
    else if (drow == 8) launch_window(WinLayout<32, 8, 2>{});
    else launch_window(WinLayout<64, 4, 1>{});
    std::memcpy(g_verify_counts, vcount, sizeof g_verify_counts);
    if (verify && getenv("EMU_DBG_W")) {                  // tile energies of one window's detection span (diagnostics)
        const int wd = atoi(getenv("EMU_DBG_W")), kd = wd / nch, cd = wd % nch;
        const int t0 = kd * p.tiles_per_slot;
        std::fprintf(stderr, "window slot %d ch-index %d: tiles from %d, stride %d, TT %d W %d abs %.2f:", kd, cd, t0, p.ptile_stride, p.tile_outs, p.burst_w, p.burst_abs);
        for (int j = -kBurstFront; j < 60 && t0 + j < p.ptile_stride; j++) if (t0 + j >= 0) std::fprintf(stderr, " %.3g", p.ptile[(size_t)cd * p.ptile_stride + t0 + j]);
        std::fprintf(stderr, "\n");
    }
    // what the tests ask for as "the exact stage's tasks": every window with exact rows, and how many of them from its first row on
    // (presence's marks and the second run's together)
    g_verify_tasks.clear();
    if (verify) {
        std::vector<uint32_t> un((size_t)vb.bm_tiles * kExBmWords);
        for (size_t i = 0; i < un.size(); i++) un[i] = vb.bm1[i] | vb.bm2[i];
        for (int k = 0; k < S; k++)
            for (int c = 0; c < nch; c++) {
                const int cov = exact_covered_rows(un.data(), vb.bm_tiles, (long long)k * p.outs_per_slot, c);
                if (cov > 0) g_verify_tasks.push_back(VerifyTask{k * nch + c, cov, 0.0, 0, 0});
            }
    }
    g_second_run.assign(vtasks.begin(), vtasks.begin() + std::min<size_t>(vtasks.size(), vcount[0]));
    g_exact_bitmap = bm; g_exact_tiles = vb.bm_tiles;
    if (g_study) g_study->tasks = g_verify_tasks;
    {
        const unsigned nblk = (unsigned)((counts[1] + kFinLanes - 1) / kFinLanes + 1);
        emu::launch(dim3(nblk), dim3(kFinLanes), [&]() {
            if (want_syms) finish_kernel<true>(p, d, drow, G, des.mmse, fin.data(), &counts[1], win_len.data(), symbits.data(), dcol_p);
            else finish_kernel<false>(p, d, drow, G, des.mmse, fin.data(), &counts[1], win_len.data(), (uint32_t *)nullptr, dcol_p);
        });
        emu::launch(dim3(4), dim3(256), [&]() {
            nsym_patch_kernel(hits.data(), &counts[0], max_hits, win_len.data(), nch, want_syms ? win_fin.data() : (const int *)nullptr);
        });
    }
    if (g_study) { g_study->d.assign(d, d + (size_t)G * drow); g_study->snr = snr; g_study->G = G; g_study->drow = drow; g_study->nch = nch; g_study->S = S; }
    int n = (int)std::min<unsigned>(counts[0], (unsigned)std::min(cap, max_hits));
    std::vector<HeaderRec> hdr((size_t)std::max(n, 1));
    if (hdr_out && want_syms && n > 0) {
        emu::launch(dim3((unsigned)std::min(n, 64)), dim3(64), [&]() {
            header_sweep_kernel(hits.data(), &counts[0], n, symbits.data(), des.wh.first18,
                                des.d.correlator == BTGPU_CORRELATOR_BTBB ? 68 : 72, hdr.data());
        });
    }
    for (int i = 0; i < n; i++) {
        const DeviceHit &h = hits[i];
        long long *r = rec_out + (size_t)i * 8;
        r[0] = h.slot; r[1] = des.d.low_channel + h.channel_idx; r[2] = h.kind; r[3] = h.offset; r[4] = h.lap; r[5] = h.ac_errors;
        r[6] = h.nsym; r[7] = 0;
        snr_out[i] = h.snr;
        if (sym_out) std::memcpy(sym_out + (size_t)i * kSymWords, h.sym >= 0 ? symbits.data() + (size_t)h.sym * kSymWords : symbits.data(), kSymWords * 4);
        if (hdr_out) {
            std::memcpy(hdr_out + (size_t)i * 132, hdr[i].uap, 64); std::memcpy(hdr_out + (size_t)i * 132 + 64, hdr[i].type, 64);
            std::memcpy(hdr_out + (size_t)i * 132 + 128, &hdr[i].fec13_ok, 4);
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------
// The whole FAST front end on the CPU: channel bank + noise stage 1 (pfbm_kernel, or the fused 100-bin pfb100_kernel
// with its tile-blocked copy dcol for the finish kernel), noise_stage2_kernel, window_kernel (squelch, M&M, slicer,
// access-code / LE search), finish_kernel, nsym_patch_kernel -- the kernels' own source, lanes as fibers (only the
// block sums over the tiles' partial energies are formed by the host loops of emu_bank_*_run).  Records: [n][8] int64 = slot, channel, kind, offset, lap, ac_errors, nsym, 0; snr_out [n].
// Returns the number of records (<= cap) or a negative error.
extern "C" int emu_front_m_run(double fs, double fc, int mode, int le, double squelch_db, const float *iq, long long x_len, int S,
                               long long *rec_out, double *snr_out, int cap)
{
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = squelch_db; cfg.mode = mode;
    cfg.flags = le ? BTGPU_FLAG_LE : 0;
    static Design des; static FastPath fp;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    rc = make_fast_path(des, fp);
    if (rc) return rc;
    long long sizes[8] = {0};
    const bool big = fp.channel.available && fp.channel.M == kPfbM;        // 100 Msps: the 100-bin bank (fused noise stage 1) + dcol
    rc = big ? emu_bank_run(fs, fc, mode, iq, x_len, 0, S, 1, nullptr, nullptr, nullptr, nullptr, nullptr, sizes)
             : emu_bank_m_run(fs, fc, mode, iq, x_len, 0, S, nullptr, nullptr, nullptr, nullptr, nullptr, sizes);
    if (rc) return rc;
    const long long G = sizes[0], zstride = sizes[3];
    const int nb = (int)sizes[1], nch = (int)sizes[2], drow = big ? 80 : (int)sizes[6];
    std::vector<float4> dbuf(((size_t)(G + 64) * drow + 3) / 4 + 4);          // 16-byte aligned rows
    float *d = (float *)dbuf.data();
    std::vector<double> P((size_t)nch * nb), Pt((size_t)nch * nb);
    std::vector<float> Z((size_t)nch * zstride * 2, 0.f);
    std::vector<float> dcol;
    std::vector<double> ptile_keep, pfine_keep;
    g_keep_ptile = &ptile_keep; g_keep_pfine = &pfine_keep;
    if (big) {
        g_keep_dcol = &dcol;
        rc = emu_bank_run(fs, fc, mode, iq, x_len, 0, S, 1, d, P.data(), Pt.data(), Z.data(), nullptr, sizes);
        g_keep_dcol = nullptr;
    } else rc = emu_bank_m_run(fs, fc, mode, iq, x_len, 0, S, d, P.data(), Pt.data(), Z.data(), nullptr, sizes);
    g_keep_ptile = nullptr; g_keep_pfine = nullptr;
    if (rc) return rc;

    // noise stage 2: the kernel itself (its wave-shuffle reduction runs on the emulator's exchange buffer)
    const NoiseStage &ns = fp.noise;
    std::vector<double> Qn((size_t)nch * S);
    {
        const size_t lds2 = s2_lds_bytes(ns.outs, ns.nw, ns.L3);
        if (lds2 > sizeof emu::dyn_lds) return BTGPU_EUNSUPPORTED;
        std::memset(emu::dyn_lds, 0xff, sizeof emu::dyn_lds);
        emu::launch(dim3((unsigned)((S + kS2Slots - 1) / kS2Slots), (unsigned)nch), dim3(256), [&]() {
            noise_stage2_kernel((const float2 *)Z.data(), zstride, ns.outs, ns.nw, ns.L3, ns.h3.data(), ns.weights.data(), Qn.data(), S, BlockSumArgs{});
        });
    }
    std::vector<float2> xv((size_t)x_len + 8);
    std::memcpy(xv.data(), iq, (size_t)x_len * sizeof(float2));
    VerifyEmu ve;
    ve.x = xv.data(); ve.x_len = x_len; ve.fp = &fp; ve.small = !big; ve.ptile = ptile_keep.data(); ve.ntiles = g_keep_ntiles; ve.mode = g_verify_mode;
    if (!big && !pfine_keep.empty()) { ve.ptile = pfine_keep.data(); ve.ntiles = g_keep_ntiles * (pfbm_tile(fp.channel.M) / 25); ve.tile_outs = 25; }
    return run_detect(des, S, nb, nch, drow, G, d, big ? dcol.data() : nullptr, P.data(), Pt.data(), Qn.data(), rec_out, snr_out, cap, g_sym_out, g_hdr_out, &ve);
}

// ---------------------------------------------------------------------------------------------------
// The DIRECT (bit-exact) front end on the CPU: ddc_direct_kernel<2> for the channel bank and for the exact noise
// filter, energy_kernel (wave shuffles emulated), demod_rows_kernel, then window / finish / nsym patch as above --
// launched with the product's own geometry (pick_shape; shared output grid, or one segment per window at the odd
// rates).  Records as emu_front_m_run.
extern "C" int emu_front_direct_run(double fs, double fc, int mode, int le, double squelch_db, const float *iq, long long x_len, int S,
                                    long long *rec_out, double *snr_out, int cap)
{
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = squelch_db; cfg.mode = mode;
    cfg.flags = le ? BTGPU_FLAG_LE : 0;
    static Design des;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    const btgpu_design &d = des.d;
    const int nch = d.high_channel - d.low_channel + 1, ops = des.outs_per_slot, drow = win_drow(nch);
    // odd samples per symbol: every window on its own decimation grid (one segment of outputs per window)
    const int ops_n = des.segmented ? d.noise_out : ops;
    const int seg_ch = des.segmented ? d.ddc_out : 0, seg_n = des.segmented ? d.noise_out : 0;
    const long long seg_stride = d.samples_per_slot;
    const long long G = (long long)ops * (S - 1) + d.ddc_out, Gn = (long long)ops_n * S;
    const int nb = (int)((G + ops - 1) / ops);
    const long long ystride = (G + 63) / 64 * 64, ystride_n = (Gn + 63) / 64 * 64;
    LaunchShape sc, sn;
    if (!pick_shape(d.decimation, des.channel.ntp, sc) || !pick_shape(d.decimation, des.noise.ntp, sn)) return BTGPU_EUNSUPPORTED;
    std::vector<float2> xbuf((size_t)x_len + 8);
    std::memcpy(xbuf.data(), iq, (size_t)x_len * sizeof(float2));
    std::vector<double> st(nch), sno(nch);
    for (int c = 0; c < nch; c++) {
        st[c] = -des.channel.foff[c] * d.decimation / cfg.sample_rate;
        sno[c] = -des.noise.foff[c] * d.decimation / cfg.sample_rate;
    }
    std::vector<float2> Y((size_t)nch * ystride), Yn((size_t)nch * ystride_n);
    auto ddc = [&](const LaunchShape &sh, const FilterBank &bank, long long first, const std::vector<double> &step, float2 *out,
                   long long Gout, long long ys, int seg) {
        if (sh.lds > sizeof emu::dyn_lds) { std::fprintf(stderr, "emu: LDS %zu\n", sh.lds); std::abort(); }
        const unsigned gx = seg ? (unsigned)(((seg + sh.T - 1) / sh.T) * S) : (unsigned)((Gout + sh.T - 1) / sh.T);
        emu::launch(dim3(gx, (unsigned)((nch + 1) / 2)), dim3((unsigned)sh.T), [&]() {
            ddc_direct_kernel<2>(xbuf.data(), x_len, first, d.decimation, bank.ntp, sh.JC, (const float2 *)bank.taps.data(),
                                 (const float2 *)bank.rot.data(), bank.rot_period, step.data(), out, Gout, ys, nch, seg, seg_stride);
        });
    };
    ddc(sc, des.channel, d.first_channel_sample, st, Y.data(), G, ystride, seg_ch);
    if (!g_rows_only) ddc(sn, des.noise, d.first_noise_sample, sno, Yn.data(), Gn, ystride_n, seg_n);   // (20 001 taps per output: most of this function's time)
    std::vector<double> P((size_t)nch * nb), Pt((size_t)nch * nb), Qn((size_t)nch * S);
    emu::launch(dim3((unsigned)nb, (unsigned)nch), dim3(256), [&]() {
        energy_kernel(Y.data(), G, ystride, ops, des.tail, P.data(), Pt.data(), nb, nch, ops);
    });
    emu::launch(dim3((unsigned)S, (unsigned)nch), dim3(256), [&]() {
        energy_kernel(Yn.data(), Gn, ystride_n, ops_n, 0, Qn.data(), (double *)nullptr, S, nch, d.noise_out);
    });
    std::vector<float4> dbuf(((size_t)(G + 64) * drow + 3) / 4 + 4);
    float *dd = (float *)dbuf.data();
    emu::launch(dim3((unsigned)((G + 63) / 64)), dim3(256), [&]() {
        demod_rows_kernel(Y.data(), G, ystride, nch, des.atan_tab, des.demod_gain, dd, drow);
    });
    return run_detect(des, S, nb, nch, drow, G, dd, nullptr, P.data(), Pt.data(), Qn.data(), rec_out, snr_out, cap, g_sym_out, g_hdr_out);
}

// the same with the packed symbols of every record's window and the GPU header sweep (BTGPU_FLAG_HEADERS)
extern "C" int emu_front_direct_headers_run(double fs, double fc, int mode, int le, double squelch_db, const float *iq, long long x_len,
                                            int S, long long *rec_out, double *snr_out, int cap, uint32_t *sym_out, uint8_t *hdr_out)
{
    g_sym_out = sym_out; g_hdr_out = hdr_out;
    const int n = emu_front_direct_run(fs, fc, mode, le, squelch_db, iq, x_len, S, rec_out, snr_out, cap);
    g_sym_out = nullptr; g_hdr_out = nullptr;
    return n;
}

// scan_symbols_kernel (the window kernel's access-code search, search_classic) over a captured symbol stream, as
// btgpu_debug_scan_symbols launches it; every qualifying offset.  out: [n][3] = absolute offset, LAP, errors, sorted.
extern "C" int emu_front_m_syms_run(double fs, double fc, int mode, int le, double squelch_db, const float *iq, long long x_len,
                                    int S, long long *rec_out, double *snr_out, int cap, uint32_t *sym_out)
{
    g_sym_out = sym_out;
    const int n = emu_front_m_run(fs, fc, mode, le, squelch_db, iq, x_len, S, rec_out, snr_out, cap);
    g_sym_out = nullptr;
    return n;
}

extern "C" long emu_scan_symbols(const uint8_t *symbols, long long n, long long *out, long cap)
{
    btgpu_config cfg{};
    cfg.sample_rate = 8e6; cfg.center_freq = 2476.5e6; cfg.squelch_db = 10.0; cfg.mode = BTGPU_MODE_SNIFFER;
    static Design des;
    des = Design();                                  // a fresh design, like a new handle
    int rc = make_design(cfg, des);
    if (rc) return rc;
    const size_t nwords = (size_t)(n + 31) / 32;
    std::vector<uint32_t> words(nwords + 1, 0u);
    for (long long i = 0; i < n; i++) if (symbols[i] & 1) words[(size_t)i >> 5] |= 1u << (i & 31);
    const unsigned long long chunks = ((unsigned long long)n + 624) / 625;
    const int max_hits = 1 << 20;
    std::vector<DeviceHit> hits((size_t)max_hits);
    unsigned int count = 0;
    emu::launch(dim3((unsigned)((chunks + kWinThreads - 1) / kWinThreads)), dim3(kWinThreads), [&]() {
        scan_symbols_kernel(words.data(), (unsigned long long)n, 1, des.ac.a0_lo, des.ac.a0_hi, &des.ac.byte_lo[0][0], &des.ac.byte_hi[0][0],
                            hits.data(), &count, max_hits);
    });
    if ((int)count > max_hits) return BTGPU_EOVERFLOW;
    std::vector<std::array<long long, 3>> all(count);
    for (unsigned int i = 0; i < count; i++) all[i] = {(long long)hits[i].slot * 625 + hits[i].offset, (long long)hits[i].lap, (long long)hits[i].ac_errors};
    std::sort(all.begin(), all.end());
    long m = 0;
    for (const auto &a : all) if (m < cap) { out[3 * m] = a[0]; out[3 * m + 1] = a[1]; out[3 * m + 2] = a[2]; m++; }
    return (long)count;
}


// ---------------------------------------------------------------------------------------------------
// Margin study (round 4, design of the exact-confirmation stage): the FAST and the DIRECT front end over the same
// capture, then the clock-recovery recursion of every window that passes the squelch on both demodulated streams in
// lockstep.  Per window one row of 12 doubles:
//   0 slot, 1 channel index, 2 first symbol index at which the two trajectories part (sliced symbol, interpolator
//   step or input index differ; nsyms if never), 3 symbols run, 4 max |out_fast - out_exact| before that point,
//   5 max |mu_fast - mu_exact| before it, 6 min |out_fast| before it, 7 min rounding margin of mu_fast * 128 before it
//   (distance of the fraction from the half-way point, in 1/128 steps), 8 |out_fast| at the parting symbol,
//   9 rounding margin at the parting symbol, 10 what parted (1 symbol, 2 step, 4 index; OR-ed), 11 squelch passes (1 fast, 2 exact)
// Returns the number of rows (<= cap).
static int g_trace_k = -1, g_trace_c = -1; static float *g_trace_out = nullptr;   // emu_window_trace: soft symbols (fast, exact) of one window
extern "C" int emu_margin_study(double fs, double fc, int mode, double squelch_db, const float *iq, long long x_len, int S,
                                int nsyms, double *rows, int cap);
extern "C" int emu_window_trace(double fs, double fc, int mode, double squelch_db, const float *iq, long long x_len, int S,
                                int nsyms, int k, int c, float *out /* [nsyms][2] */)
{
    std::vector<double> rows((size_t)12 * S * 80);
    g_trace_k = k; g_trace_c = c; g_trace_out = out;
    const int rc = emu_margin_study(fs, fc, mode, squelch_db, iq, x_len, S, nsyms, rows.data(), S * 80);
    g_trace_k = g_trace_c = -1; g_trace_out = nullptr;
    return rc;
}
extern "C" int emu_margin_study(double fs, double fc, int mode, double squelch_db, const float *iq, long long x_len, int S,
                                int nsyms, double *rows, int cap)
{
    StudyCapture fast, exact;
    std::vector<long long> rec((size_t)8 * 65536); std::vector<double> sn(65536);
    g_study = &fast;
    int rc = emu_front_m_run(fs, fc, mode, 0, squelch_db, iq, x_len, S, rec.data(), sn.data(), 65536);
    g_study = &exact;
    if (rc >= 0) rc = emu_front_direct_run(fs, fc, mode, 0, squelch_db, iq, x_len, S, rec.data(), sn.data(), 65536);
    g_study = nullptr;
    if (rc < 0) return rc;
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = squelch_db; cfg.mode = mode;
    static Design des;
    des = Design();
    rc = make_design(cfg, des);
    if (rc) return rc;
    std::vector<uint64_t> pcol(des.ac.btbb_pcol, des.ac.btbb_pcol + 24);
    const WindowParams p = make_window_params(des, S, 1, 0, 1, false, pcol.data());
    const int nch = fast.nch, ops = des.outs_per_slot, demod_n = p.ddc_out - 1;
    const unsigned ni = (unsigned)(demod_n - 8);
    int n = 0;
    for (int k = 0; k < S; k++)
        for (int c = 0; c < nch; c++) {
            const size_t w = (size_t)k * nch + c;
            const bool pf = fast.snr[w] >= p.target_snr, pe = exact.snr[w] >= p.target_snr;
            if (!pf && !pe) continue;
            if (n >= cap) return n;
            double *r = rows + (size_t)n++ * 12;
            for (int i = 0; i < 12; i++) r[i] = 0;
            r[0] = k; r[1] = c; r[11] = (pf ? 1 : 0) + (pe ? 2 : 0);
            r[2] = nsyms; r[6] = 1e9; r[7] = 1e9;
            if (!(pf && pe)) { r[2] = 0; continue; }
            auto sample = [&](const StudyCapture &s, unsigned i) -> float {
                if (i == 0) return 0.f;                                    // policy Q1
                return s.d[((size_t)k * ops + i) * s.drow + c];
            };
            float muf = p.mu0, omf = p.omega0, laf = 0.f, mue = p.mu0, ome = p.omega0, lae = 0.f;
            unsigned iif = 0, iie = 0;
            int oo = 0;
            for (; oo < nsyms && iif < ni && iie < ni; oo++) {
                auto interp = [&](const StudyCapture &s, unsigned ii, float mu, int &imu) -> float {
                    imu = (int)rintf(mu * 128.0f);
                    const float *T = des.mmse + (size_t)imu * 8;
                    float acc = 0.f;
                    for (int q = 0; q < 8; q++) acc = fmaf(T[7 - q], sample(s, ii + q), acc);
                    return acc;
                };
                int imf, ime;
                const float of = interp(fast, iif, muf, imf), oe = interp(exact, iie, mue, ime);
                const double mo = std::fabs((double)of);
                const double fr = (double)muf * 128.0 - std::floor((double)muf * 128.0);
                const double mm = std::fabs(fr - 0.5);
                // the same instant on both sides?  (ii, imu) may differ by a whole sample at mu = 1.0 / 0.0 -- compare positions
                const long long posf = (long long)iif * 128 + imf, pose = (long long)iie * 128 + ime;
                int parted = 0;
                if (g_trace_out && k == g_trace_k && c == g_trace_c) {       // trace mode: both trajectories run on independently
                    g_trace_out[2 * oo] = of; g_trace_out[2 * oo + 1] = oe;
                    iif += (unsigned)mm_update(of, laf, omf, muf, p);
                    iie += (unsigned)mm_update(oe, lae, ome, mue, p);
                    continue;
                }
                if ((of < 0) != (oe < 0)) parted |= 1;
                if (posf != pose) parted |= (iif != iie && imf == ime) ? 4 : 2;
                if (parted) { r[2] = oo; r[8] = mo; r[9] = mm; r[10] = parted; break; }
                r[4] = std::max(r[4], std::fabs((double)of - (double)oe));
                r[5] = std::max(r[5], std::fabs(((double)iif + muf) - ((double)iie + mue)));
                r[6] = std::min(r[6], mo); r[7] = std::min(r[7], mm);
                iif += (unsigned)mm_update(of, laf, omf, muf, p);
                iie += (unsigned)mm_update(oe, lae, ome, mue, p);
            }
            r[3] = oo;
        }
    return n;
}


// The exact rows against the bit-exact front end's (diagnostics / test): every (channel, tile) pair the polyphase run marked --
// presence and the first run's uncovered hits -- must hold the DIRECT path's demodulated rows bit for bit.  Returns the number
// of differing rows (0 = exact), or a negative error; first_bad[0..3] = channel index, grid row, tile of the first difference,
// rows compared.
extern "C" long emu_verify_check(double fs, double fc, int mode, double squelch_db, const float *iq, long long x_len, int S, long long *first_bad)
{
    StudyCapture fast, exact;
    std::vector<long long> rec((size_t)8 * 65536); std::vector<double> sn(65536);
    g_study = &fast;
    int rc = emu_front_m_run(fs, fc, mode, 0, squelch_db, iq, x_len, S, rec.data(), sn.data(), 65536);
    unsigned int fast_counts[kVerCountWords]; std::memcpy(fast_counts, g_verify_counts, sizeof fast_counts);
    const std::vector<uint32_t> fast_bm = g_exact_bitmap; const int fast_tiles = g_exact_tiles;   // (emu_verify_counts after this call: the polyphase run's)
    g_study = &exact; g_rows_only = true;
    if (rc >= 0) rc = emu_front_direct_run(fs, fc, mode, 0, squelch_db, iq, x_len, S, rec.data(), sn.data(), 65536);
    g_study = nullptr; g_rows_only = false;
    std::memcpy(g_verify_counts, fast_counts, sizeof fast_counts);
    if (rc < 0) return rc;
    btgpu_config cfg{};
    cfg.sample_rate = fs; cfg.center_freq = fc; cfg.squelch_db = squelch_db; cfg.mode = mode;
    static Design des;
    des = Design();
    rc = make_design(cfg, des);
    if (rc) return rc;
    const int nch = fast.nch;
    long bad = 0;
    long long checked = 0;
    const uint32_t *b1 = fast_bm.data(), *b2 = b1 + (size_t)fast_tiles * kExBmWords;
    for (int t = 0; t < fast_tiles; t++)
        for (int c = 0; c < nch; c++) {
            if (!(((b1[(size_t)t * kExBmWords + (c >> 5)] | b2[(size_t)t * kExBmWords + (c >> 5)]) >> (c & 31)) & 1u)) continue;
            for (long long g = std::max<long long>(1, exact_tile_row0(t)); exact_tile_of(g) == t && g < fast.G; g++) {
                const float a = fast.d[(size_t)g * fast.drow + c], b = exact.d[(size_t)g * exact.drow + c];
                checked++;
                if (std::memcmp(&a, &b, 4) != 0) {
                    if (!bad && first_bad) { first_bad[0] = c; first_bad[1] = g; first_bad[2] = t; }
                    if (getenv("EMU_DBG_ROWS") && bad < 60) std::fprintf(stderr, "tile %d ch %d row %lld: %.9g vs %.9g\n", t, c, g, a, b);
                    bad++;
                }
            }
        }
    if (first_bad) first_bad[3] = checked;
    return bad;
}
