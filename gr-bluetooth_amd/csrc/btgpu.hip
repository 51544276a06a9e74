// btgpu.hip -- host runtime and C ABI (include/btgpu.h) over the gfx950 kernels.
// No CPU fallback exists: every entry point that needs the GPU fails with
// BTGPU_ENODEVICE / BTGPU_EDEVICE when the device or the code object is unusable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "btgpu.h"
#include "design.h"
#include "kernels.hip.h"
#include "pfb100.hip.h"
#include "bank_launch.h"
#include "hopseq.hip.h"

using namespace btgpu;

#define HIPCHK(h, expr)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            (h)->set_error(std::string(#expr) + ": " + hipGetErrorString(e__));              \
            return BTGPU_EDEVICE;                                                             \
        }                                                                                     \
    } while (0)

namespace {


struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

constexpr int kCtxMax = 3;
// The eight streams of a destroyed handle are kept and handed to the next handle on the same device.  Streams created
// after others were destroyed measured slower: the SECOND handle of a process ran every kernel 4-19 % longer (C8 14.8 against
// 19.8 Gsamples/s, profiles/r06_zz_second_handle.txt); with the first handle's streams it runs like the first.  BTGPU_STREAM_POOL=0: off.
struct StreamSet { int device; hipStream_t s[8]; };
static std::mutex g_stream_pool_mu;
static std::vector<StreamSet> g_stream_pool;

struct btgpu_handle {
    Design des;
    FastPath fp;
    bool use_pfb = false, use_staged = false, keep_Y = false;
    int margin = 0;                  // samples in front of window 0 the kernels may read
    std::vector<float> pre;          // the `margin` samples preceding the next work() buffer
    int device = 0;
    hipStream_t stream = nullptr;
    int drow = 80;                               // row stride (floats) of the time-major demodulated stream
    // Three streams form the pipeline of a batch: `stream` (or the caller's) carries the FRONT -- the banks, which
    // read the input and write only per-context buffers; `post_stream` the POST stage -- tile sums -> block sums,
    // squelch stage 2, window kernel; `tail_stream` the TAIL -- finish / nsym / header sweep / record copies.  The
    // banks of batch n+1 therefore start the moment the banks of batch n end, with post(n) and tail(n) running beside
    // them on the CUs' spare issue slots (round 2: only the tail overlapped, 0.54 ms of a 2.05 ms step waited in line).
    hipStream_t post_stream = nullptr, tail_stream = nullptr;
    // With the exact stage the tail of a batch (direct-form DDC of the handed-over windows -> their window kernel -> finish -> copies)
    // is a dependent chain longer than a step: consecutive batches' tails run on different streams and overlap each other
    hipStream_t tail_extra[kCtxMax - 1] = {nullptr, nullptr};
    int ntail = 1;
    hipStream_t last_tail = nullptr;
    // Deferred squelch (BTGPU_DEFER=1): squelch stage 2 (+ block sums, + the per-window SNR) runs on `sq_stream` BESIDE the window
    // kernel, not in front of it; the decision snr >= threshold is applied where records leave (window_kernel<.., true>,
    // nsym_patch_kernel).
    hipStream_t sq_stream = nullptr;
    bool deferred = false;
    static constexpr int kCtx = kCtxMax; // in-flight batches (BTGPU_FLAG_ASYNC): front(n+2) | post(n+1) | tail(n)
    struct TailCtx {                 // per in-flight batch: everything the front writes and the post / tail stages read
        DevBuf d_winlen, d_hits, d_hitcount, d_fin, d_winfin, d_symbits, d_hdr;
        DevBuf d_d;                           // demodulated stream of the batch
        DevBuf d_dcol;                        // 100-bin bank: the same stream tile by tile channel-major [tile][80][25], what finish_kernel reads
        DevBuf d_ptile, d_phead;              // polyphase banks: |Y|^2 tile sums (-> block_sum_kernel on the post stream)
        DevBuf d_pfine;                       // small-M F8 bank: |Y|^2 sums per 25 instants (the exact stage's burst scan)
        DevBuf d_Z;                           // staged squelch: stage-1 output (-> noise_stage2_kernel on the post stream)
        DevBuf d_vtasks, d_vcount, d_dxt, d_winbits_v;
        DevBuf d_bm;                                            // exact rows' bitmaps bm1 | bm2, [2][bm_tiles][kExBmWords] (presence / the first run's uncovered hits)
        DevBuf d_chanfloor;                                     // presence's last-resort noise reference: each channel's quietest tile of the batch, [80] the quietest of all
        DevBuf d_eon, d_eoff, d_snr;          // E_on, E_off, SNR per window (window_kernel, or squelch_kernel when the squelch is deferred)   // exact confirmation (verify.hip.h): task list, exact rows, task stream
        HeaderRec *h_hdr = nullptr;           // pinned: sweeps of the first eager_hdr hits (records_out_kernel)
        uint32_t *h_sym = nullptr;            // pinned: packed symbols of the first eager_fin hit windows (records_out_kernel)
        unsigned int *h_count = nullptr;      // pinned: {hits, finish records, -, -, verify tasks, verify tiles, turned away, -}
        DeviceHit *h_hits = nullptr;          // pinned: first eager_hits records (records_out_kernel)
        // timing events (recorded only with BTGPU_FLAG_TIMING): front 0 start, 1 channel bank, 2 demod / energy (direct
        // form), 3 noise stage 1 / direct noise bank, 4 direct noise energy; post 5 start, 6 block sums, 7 squelch
        // stage 2, 8 window; tail 9 start, 10 end
        // ... 11 exact stage done (tail)
        // ... 12 / 13 around presence + the exact rows in line (post)
        // ... 14: squelch stage 2 done (on its side stream, beside presence and the exact rows)
        hipEvent_t ev[15] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        hipEvent_t front_done = nullptr, detect_done = nullptr, tail_done = nullptr, squelch_done = nullptr, exact_done = nullptr, floor_done = nullptr;
        int S = 0;
        uint64_t abs_first_slot = 0;
        bool pending = false;
    } tc[kCtx];
    int cur = 0, nctx = 1;
    int last_ctx = 0;                // context of the most recent batch (debug fetch)
    bool async = false;
    bool timing_on = false;          // BTGPU_FLAG_TIMING / _BANK: bracket the kernels with events (btgpu_last_timing)
    bool timing_full = false;        // every kernel (BTGPU_FLAG_TIMING), not just the channel bank
    bool no_nsym = false;            // BTGPU_FLAG_NO_NSYM: skip the M&M continuation that produces hit.nsym
    int verify = 0;                  // exact rows under the polyphase path's records (exact.hip.h): 0 off, 1 presence + uncovered hits, 2 uncovered hits only
    int vcap = 0, bm_tiles = 0;
    ExactRowsKernel ex_kern = nullptr; size_t ex_lds = 0;
    bool exact_all = false;          // BTGPU_FLAG_EXACT_ALL: every row of every channel is recomputed (no presence)
    std::vector<const void *> lds_opted;   // bank kernels that have been granted > 48 KiB of dynamic LDS on this handle's device
    DevBuf d_tapsA;                  // the direct-form channel bank's taps as exact_rows_kernel's A operand (exact_pack_taps)
    bool pipelined = false;          // front writes per-context buffers only: front(n+1) may overlap post(n)
    hipStream_t copy_stream = nullptr;
    hipStream_t spill_stream = nullptr;          // harvest: records beyond the eager copies (never behind an input copy)
    // ... which land in PAGE-LOCKED buffers kept by the handle (grown on demand).  A device-to-host copy into pageable memory is
    // not an asynchronous copy: with three batches queued the call came back only when the device had drained -- 26 ms per five
    // batches at C8 (11 068 hit windows per batch, 2876 beyond the eager copy): the pipeline emptied every fifth call, 7.2 ms per
    // step for 4.0 ms of kernels (profiles/r06_q_c8_*)
    void *h_spill[3] = {nullptr, nullptr, nullptr}; size_t h_spill_cap[3] = {0, 0, 0};     // hits | symbols | headers
    void *spill_buf(int k, size_t bytes)
    {
        if (bytes <= h_spill_cap[k]) return h_spill[k];
        if (h_spill[k]) { (void)hipHostFree(h_spill[k]); h_spill[k] = nullptr; h_spill_cap[k] = 0; }
        const size_t cap = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
        if (hipHostMalloc(&h_spill[k], cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h_spill[k] = nullptr; return nullptr; }
        h_spill_cap[k] = cap;
        return h_spill[k];
    }
    static constexpr unsigned kEagerHits = 65536;
    static constexpr unsigned kEagerFin = 8192;
    unsigned eager_fin = kEagerFin, eager_hdr = kEagerFin, eager_hits = kEagerHits;   // capacity (records) of the page-locked h_sym / h_hdr / h_hits of a context: records_out_kernel fills them
    bool want_syms = false, want_hdrs = false;
    // The host queue: records of harvested batches in emission order, consumed from the front (qhead).  Flat arenas -- a record
    // costs one memcpy of its symbols, no allocation (a vector per record was 5 ms of host time per 11 000-record batch at C8:
    // twice the batch's kernels, profiles/r06_q_c8_*).
    std::vector<btgpu_header> qhdr;              // header sweep per queued hit (BTGPU_FLAG_HEADERS)
    std::vector<uint32_t> qbits;                 // packed symbols per queued hit [n][kSymWords] (BTGPU_FLAG_SYMBOLS)
    std::vector<uint8_t> qhas;                   // ... and whether the record has any (a window the tail did not finish has none)
    size_t qhead = 0;                            // first record not yet polled
    size_t pending_records() const { return queue.size() - qhead; }
    void pop_records(size_t n)
    {
        qhead += n;
        if (qhead == queue.size()) { queue.clear(); qhdr.clear(); qbits.clear(); qhas.clear(); qhead = 0; }
        else if (qhead >= 8192 && 2 * qhead >= queue.size()) {                     // (amortised: at most one move per record)
            queue.erase(queue.begin(), queue.begin() + qhead);
            if (want_hdrs) qhdr.erase(qhdr.begin(), qhdr.begin() + qhead);
            if (want_syms) { qbits.erase(qbits.begin(), qbits.begin() + qhead * kSymWords); qhas.erase(qhas.begin(), qhas.begin() + qhead); }
            qhead = 0;
        }
    }
    std::string err;
    int sticky = BTGPU_OK;

    // batch geometry
    int max_slots = 0, max_hits = 0;
    long long ystride = 0, ystride_n = 0;
    int nb_max = 0;
    size_t in_cap = 0;           // complex samples the staging buffer holds

    // device memory
    DevBuf d_in, d_in_b, d_taps_ch, d_taps_n, d_rot_ch, d_rot_n, d_rotstep_ch, d_rotstep_n;
    DevBuf d_Y, d_Yn, d_P, d_Pt, d_Q, d_mmse, d_atan, d_aclo, d_achi;
    DevBuf d_le_hdr, d_le_whiten, d_le_index, d_winbits;
    DevBuf d_pfb_taps_ch, d_pfb_tw, d_binpos_ch, d_binnat_ch, d_rho_ch, d_krot_ch, d_b2map_fused, d_b2map_fused_wide, d_b2map_ch, d_b2map_noise, d_b2map_f320, d_dftw_ch, d_dftw_n;
    DevBuf d_pfb_taps_n, d_binpos_n, d_krot_n, d_h3, d_w, d_taps_s1, d_rot_s1, d_rotstep_s1, d_prof, d_pcol, d_wh18;
    LaunchShape shape_s1;
    bool noise_pfb = false;
    bool pfb_small = false, noise_small = false;   // small-M polyphase banks (rates below 100 Msps)
    bool fuse_noise = false;         // noise stage 1 rides on the channel bank's staged input
    bool use_dcol = false;          // 100-bin bank: also writes the tile-blocked channel-major copy the tail reads
    long long zstride = 0;
    int ntiles_max = 0;
    LaunchShape shape_ch, shape_n;

    // last-batch bookkeeping (debug fetch)
    int last_S = 0;
    long long last_G = 0;

    // host side
    std::vector<btgpu_hit> queue;
    std::vector<float> carry;    // btgpu_push history carry (interleaved)
    uint64_t push_slot = 0;
    btgpu_timing timing{};

    void set_error(const std::string &s) { err = s; }

    int alloc(DevBuf &b, size_t bytes)
    {
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(&b.p, bytes);
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
            return BTGPU_ENOMEM;
        }
        b.bytes = bytes;
        return BTGPU_OK;
    }
    int upload(DevBuf &b, const void *src, size_t bytes)
    {
        int rc = alloc(b, bytes);
        if (rc) return rc;
        if (bytes) {
            hipError_t e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { set_error(std::string("hipMemcpy: ") + hipGetErrorString(e)); return BTGPU_EDEVICE; }
        }
        return BTGPU_OK;
    }
    void release()
    {
        DevBuf *all[] = {&d_in, &d_in_b, &d_taps_ch, &d_taps_n, &d_rot_ch, &d_rot_n, &d_rotstep_ch, &d_rotstep_n,
                         &d_Y, &d_Yn, &d_P, &d_Pt, &d_Q, &d_mmse, &d_atan, &d_aclo, &d_achi,
                         &d_le_hdr, &d_le_whiten, &d_le_index, &d_winbits,
                         &d_pfb_taps_ch, &d_pfb_tw, &d_binpos_ch, &d_binnat_ch, &d_rho_ch, &d_krot_ch, &d_b2map_fused, &d_b2map_fused_wide, &d_b2map_ch, &d_b2map_noise, &d_b2map_f320, &d_dftw_ch, &d_dftw_n,
                         &d_pfb_taps_n, &d_binpos_n, &d_krot_n, &d_h3, &d_w, &d_taps_s1, &d_rot_s1, &d_rotstep_s1, &d_prof, &d_pcol, &d_wh18, &d_tapsA};
        for (DevBuf *b : all) if (b->p) { (void)hipFree(b->p); b->p = nullptr; }
        for (TailCtx &t : tc) {
            DevBuf *tb[] = {&t.d_winlen, &t.d_hits, &t.d_hitcount, &t.d_fin, &t.d_winfin, &t.d_symbits, &t.d_hdr, &t.d_d, &t.d_dcol,
                            &t.d_ptile, &t.d_phead, &t.d_pfine, &t.d_Z, &t.d_vtasks, &t.d_vcount, &t.d_dxt, &t.d_winbits_v, &t.d_bm, &t.d_chanfloor, &t.d_eon, &t.d_eoff, &t.d_snr};
            if (t.h_hdr) { (void)hipHostFree(t.h_hdr); t.h_hdr = nullptr; }
            if (t.h_sym) { (void)hipHostFree(t.h_sym); t.h_sym = nullptr; }
            for (DevBuf *b : tb) if (b->p) { (void)hipFree(b->p); b->p = nullptr; }
            for (auto &e : t.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            for (hipEvent_t *e : {&t.front_done, &t.detect_done, &t.tail_done, &t.squelch_done, &t.exact_done, &t.floor_done}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
            if (t.h_count) { (void)hipHostFree(t.h_count); t.h_count = nullptr; }
            if (t.h_hits) { (void)hipHostFree(t.h_hits); t.h_hits = nullptr; }
        }
        for (int k = 0; k < 3; k++) if (h_spill[k]) { (void)hipHostFree(h_spill[k]); h_spill[k] = nullptr; h_spill_cap[k] = 0; }
        for (int k = 0; k < 2; k++) {
            if (h_stage[k]) { (void)hipHostFree(h_stage[k]); h_stage[k] = nullptr; }
            if (ev_copied[k]) { (void)hipEventDestroy(ev_copied[k]); ev_copied[k] = nullptr; }
            if (ev_consumed[k]) { (void)hipEventDestroy(ev_consumed[k]); ev_consumed[k] = nullptr; }
            if (ev_vdone[k]) { (void)hipEventDestroy(ev_vdone[k]); ev_vdone[k] = nullptr; }
        }
        if (streams_poolable && stream && post_stream && tail_stream && copy_stream && spill_stream && tail_extra[0] && tail_extra[1] && sq_stream) {
            StreamSet ss; ss.device = device;
            hipStream_t *all[8] = {&stream, &post_stream, &tail_stream, &copy_stream, &spill_stream, &tail_extra[0], &tail_extra[1], &sq_stream};
            for (int k = 0; k < 8; k++) { (void)hipStreamSynchronize(*all[k]); ss.s[k] = *all[k]; *all[k] = nullptr; }
            std::lock_guard<std::mutex> lk(g_stream_pool_mu);
            g_stream_pool.push_back(ss);
        }
        for (hipStream_t *st : {&stream, &post_stream, &tail_stream, &copy_stream, &spill_stream, &tail_extra[0], &tail_extra[1], &sq_stream})
            if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
    }
    bool streams_poolable = false;

    BankBuffers bank_buffers(const float2 *d_x, const TailCtx &t, bool with_dcol = false) const
    {
        BankBuffers b;
        const DevBuf &d_d = t.d_d, &d_ptile = t.d_ptile, &d_phead = t.d_phead, &d_Z = t.d_Z;
        b.dcol = (use_dcol && with_dcol) ? (float *)t.d_dcol.p : nullptr;
        b.x = d_x;
        b.taps_ch = (const float2 *)d_pfb_taps_ch.p; b.twiddle = (const float2 *)d_pfb_tw.p;
        b.krot_ch = (const float2 *)d_krot_ch.p; b.rho_ch = (const float2 *)d_rho_ch.p;
        b.binpos_ch = (const int *)d_binpos_ch.p; b.binnat_ch = (const int *)d_binnat_ch.p;
        b.b2map_fused = (const uint16_t *)d_b2map_fused.p; b.b2map_fused_wide = (const uint16_t *)d_b2map_fused_wide.p;
        b.b2map_ch = (const uint16_t *)d_b2map_ch.p;
        b.b2map_noise = (const uint16_t *)d_b2map_noise.p; b.b2map_f320 = (const uint16_t *)d_b2map_f320.p;
        b.d = (float *)d_d.p; b.ptile = (double *)d_ptile.p; b.phead = (double *)d_phead.p; b.pfine = (double *)t.d_pfine.p;
        b.Ydebug = (keep_Y && use_pfb) ? (float2 *)d_Y.p : nullptr; b.ystride = ystride;
        b.taps_n = (const float2 *)d_pfb_taps_n.p; b.krot_n = (const float2 *)d_krot_n.p;
        b.binpos_n = (const int *)d_binpos_n.p;
        b.Z = (float2 *)d_Z.p; b.zstride = zstride;
        b.prof = (unsigned long long *)d_prof.p;
        b.dftw_ch = (const float2 *)d_dftw_ch.p; b.dftw_n = (const float2 *)d_dftw_n.p; b.drow = drow;
        return b;
    }
    // host -> device staging of btgpu_work / btgpu_process_host: two pinned host buffers and two device input
    // buffers, so that the copy of batch n+1 (host memcpy + DMA on the copy stream) overlaps the kernels of batch n
    float2 *h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    hipEvent_t ev_vdone[2] = {nullptr, nullptr};      // the batch's tail (whose exact stage re-reads the input) has left the tail stream
    bool stage_used[2] = {false, false};
    unsigned stage_turn = 0;
    int stage_batch(const float *head, size_t n_head_zero, const float *body, size_t n_body, long long w0,
                    uint64_t abs_first_slot, int S);
    int process_batch(const float2 *d_x, size_t x_len, long long w0, uint64_t abs_first_slot, int S, hipStream_t st);
    int harvest(TailCtx &t);
    int harvest_all(bool block);
};

// ---------------------------------------------------------------------------------------
// one batch of S slots; d_x[w0] = first sample of window 0 of the batch, d_x[0..w0) = margin
// ---------------------------------------------------------------------------------------
int btgpu_handle::process_batch(const float2 *d_x, size_t x_len, long long w0, uint64_t abs_first_slot,
                                int S, hipStream_t st)
{
    const btgpu_design &d = des.d;
    const int nch = d.high_channel - d.low_channel + 1;
    const int ops = des.outs_per_slot;
    const long long G = (long long)ops * (S - 1) + d.ddc_out;
    const int ops_n = des.segmented ? d.noise_out : ops;       // noise outputs per slot
    const long long Gn = (long long)ops_n * S;
    const int seg_ch = des.segmented ? d.ddc_out : 0, seg_n = des.segmented ? d.noise_out : 0;
    const long long seg_stride = d.samples_per_slot;
    const int nb = (int)((G + ops - 1) / ops);
    last_S = S;
    last_G = G;
    TailCtx &t = tc[cur];
    hipStream_t tail_stream = ntail > 1 && (cur % ntail) > 0 ? tail_extra[(cur % ntail) - 1] : this->tail_stream;   // this batch's tail
    last_tail = tail_stream;
    int carried = BTGPU_OK;                                      // overflow of the batch harvested here
    if (t.pending) { int hrc = harvest(t); if (hrc == BTGPU_EOVERFLOW) carried = hrc; else if (hrc != BTGPU_OK) return hrc; }
    hipEvent_t *ev = t.ev;
    DevBuf &d_winlen = t.d_winlen, &d_hits = t.d_hits, &d_hitcount = t.d_hitcount, &d_fin = t.d_fin, &d_d = t.d_d;
    DevBuf &d_winfin = t.d_winfin, &d_symbits = t.d_symbits;
    t.S = S; t.abs_first_slot = abs_first_slot;
    // BTGPU_FLAG_TIMING: every mark; BTGPU_FLAG_TIMING_BANK alone: only the two around the channel bank
    auto mark = [&](int k, hipStream_t s_) -> hipError_t { return (timing_on && (timing_full || k <= 1 || k == 12 || k == 13)) ? hipEventRecord(ev[k], s_) : hipSuccess; };   // (light form: the bank and the exact rows)

    // =========================== FRONT (stream `st`): the banks ===========================
    // (The post stage runs behind it on the same stream unless BTGPU_PIPE=1, see btgpu_create.)
    HIPCHK(this, hipMemsetAsync(d_hitcount.p, 0, 2 * sizeof(unsigned int), st));
    if (verify) HIPCHK(this, hipMemsetAsync(t.d_vcount.p, 0, kVerCountWords * sizeof(unsigned int), st));
    HIPCHK(this, mark(0, st));
    int ntiles = 0, tiles_per_block = 1, tail_tiles = 0;
    bool fused_m = false;                                       // small-M banks: stage 1 ran inside the channel bank's kernel

    // ---- channel bank -> demodulated stream d[g][nch] + |Y|^2 tile sums (polyphase) or block sums P, Pt (direct) ----
    if (use_pfb && pfb_small) {
        BankBuffers bb = bank_buffers(d_x, t);
        auto L = [&](void (*kern)(PfbmParams), int grid, int threads, size_t lds, const PfbmParams &p) {
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
        };
        // squelch stage 1 from the channel bank's staged input where both are 8-bin banks (C8).  Opt-in (BTGPU_C8_FUSE=1) until the
        // whole GPU suite has run with it: bit-identical outputs under the emulator and in the 8 Msps GPU tests, profiles/r04_v_*
        static const bool fuse_m_on = getenv("BTGPU_C8_FUSE") && atoi(getenv("BTGPU_C8_FUSE")) == 1;
        fused_m = fuse_m_on && use_staged && noise_small && pfbm_fuse_noise(des, fp, S, G, drow);
        ntiles = launch_channel_bank_m(des, fp, bb, x_len, w0, G, L, fused_m ? S : 0);
        tiles_per_block = ops / pfbm_tile(fp.channel.M); tail_tiles = des.tail / pfbm_tile(fp.channel.M);
        HIPCHK(this, mark(1, st));
    } else if (use_pfb) {
        constexpr int TT = kBankNT - 1;
        BankBuffers bb = bank_buffers(d_x, t, true);
        // BTGPU_BANK_LDS_PAD (diagnostics): extra dynamic LDS per workgroup, i.e. fewer resident tiles per CU (occupancy sweeps)
        static const size_t lds_pad = [] { const char *e = getenv("BTGPU_BANK_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
        auto L = [&](void (*kern)(PfbParams), int grid, int threads, size_t lds, const PfbParams &p) {
            // more than 48 KiB of dynamic LDS is an opt-in per kernel: asked for here, next to the launch, once per
            // instantiation actually launched (a list kept elsewhere drifts from what launch_channel_bank picks)
            // (per handle: btrx_amd --gpus N runs one handle per device, each on its own host thread -- ADVICE r4)
            std::vector<const void *> &opted = lds_opted;
            if (lds + lds_pad > 48 * 1024 && std::find(opted.begin(), opted.end(), (const void *)kern) == opted.end()) {
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds + lds_pad, 64 * 1024));
                opted.push_back((const void *)kern);
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)threads), lds + lds_pad, st, p);
        };
        // BTGPU_BANK: run256 (default) | run320 -- pfb100f_kernel, runs of tiles per workgroup, four / five waves;
        // legacy | wide -- the round-2 kernel with four / eight waves per tile (A/B timing)
        static const int variant = [] {
            const char *e = getenv("BTGPU_BANK");
            if (!e) return (int)kBankRun256;
            const std::string v(e);
            return v == "legacy" ? (int)kBankLegacy : v == "wide" ? (int)kBankLegacyWide : v == "run320" ? (int)kBankRun320 :
                   v == "run512" ? (int)kBankRun512 : v == "run512r" ? (int)kBankRun512r : v == "run256d" ? (int)kBankRun256d :
                   v == "run256e" ? (int)kBankRun256e : v == "run256a" ? (int)kBankRun256a : (int)kBankRun256;
        }();
        ntiles = launch_channel_bank(des, fp, fuse_noise, bb, x_len, w0, S, G, nb, L, variant);
        tiles_per_block = ops / TT; tail_tiles = des.tail / TT;
        HIPCHK(this, mark(1, st));
    } else {
        const LaunchShape &s = shape_ch;
        const unsigned gx = seg_ch ? (unsigned)(((seg_ch + s.T - 1) / s.T) * S) : (unsigned)((G + s.T - 1) / s.T);
        dim3 grid(gx, (unsigned)((nch + 1) / 2));
        hipLaunchKernelGGL(ddc_direct_kernel<2>, grid, dim3(s.T), s.lds, st, d_x, (long long)x_len,
                           w0 + (long long)d.first_channel_sample, d.decimation, des.channel.ntp, s.JC,
                           (const float2 *)d_taps_ch.p, (const float2 *)d_rot_ch.p, des.channel.rot_period,
                           (const double *)d_rotstep_ch.p, (float2 *)d_Y.p, G, ystride, nch, seg_ch, seg_stride);
        HIPCHK(this, mark(1, st));
        dim3 g2((unsigned)nb, (unsigned)nch);
        hipLaunchKernelGGL(energy_kernel, g2, dim3(256), 0, st, (const float2 *)d_Y.p, G,
                           ystride, ops, des.tail, (double *)d_P.p, (double *)d_Pt.p, nb, nch, ops);
        hipLaunchKernelGGL(demod_rows_kernel, dim3((unsigned)((G + 63) / 64)), dim3(256), 0, st, (const float2 *)d_Y.p, G,
                           ystride, nch, (const float *)d_atan.p, des.demod_gain, (float *)d_d.p, drow);
    }
    HIPCHK(this, mark(2, st));

    // ---- noise bank: stage 1 of the staged squelch (unless fused into the channel bank) or the direct form ----
    const NoiseStage &ns = fp.noise;
    if (use_staged) {
        const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
        const long long xs0 = w0 + d.first_noise_sample - ns.pad - (long long)ns.Jm * ns.R;
        if (fuse_noise || fused_m) {
            // stage 1 already ran inside the channel-bank kernel
        } else if (noise_pfb) {
            BankBuffers bb = bank_buffers(d_x, t);
            auto L = [&](void (*kern)(PfbParams), int grid, int threads, size_t lds, const PfbParams &p) {
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
            };
            launch_noise_bank(des, fp, bb, x_len, w0, S, L);
        } else if (noise_small) {
            BankBuffers bb = bank_buffers(d_x, t);
            auto L = [&](void (*kern)(PfbmParams), int grid, int threads, size_t lds, const PfbmParams &p) {
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)threads), lds, st, p);
            };
            launch_noise_bank_m(des, fp, bb, x_len, w0, S, L);
        } else {
            // stage 1 as a direct-form bank: B-spline prototype (a few hundred taps at most), hop R
            const LaunchShape &s = shape_s1;
            dim3 grid((unsigned)((Tn + s.T - 1) / s.T), (unsigned)((nch + 1) / 2));
            hipLaunchKernelGGL(ddc_direct_kernel<2>, grid, dim3(s.T), s.lds, st, d_x, (long long)x_len, xs0, ns.R,
                               ns.direct.ntp, s.JC, (const float2 *)d_taps_s1.p, (const float2 *)d_rot_s1.p,
                               ns.direct.rot_period, (const double *)d_rotstep_s1.p, (float2 *)t.d_Z.p, Tn, zstride, nch, 0, 0LL);
        }
        HIPCHK(this, mark(3, st));
        HIPCHK(this, mark(4, st));
    } else {
        const LaunchShape &s = shape_n;
        const unsigned gx = seg_n ? (unsigned)(((seg_n + s.T - 1) / s.T) * S) : (unsigned)((Gn + s.T - 1) / s.T);
        dim3 grid(gx, (unsigned)((nch + 1) / 2));
        hipLaunchKernelGGL(ddc_direct_kernel<2>, grid, dim3(s.T), s.lds, st, d_x, (long long)x_len,
                           w0 + (long long)d.first_noise_sample, d.decimation, des.noise.ntp, s.JC,
                           (const float2 *)d_taps_n.p, (const float2 *)d_rot_n.p, des.noise.rot_period,
                           (const double *)d_rotstep_n.p, (float2 *)d_Yn.p, Gn, ystride_n, nch, seg_n, seg_stride);
        HIPCHK(this, mark(3, st));
        dim3 g2((unsigned)S, (unsigned)nch);
        hipLaunchKernelGGL(energy_kernel, g2, dim3(256), 0, st, (const float2 *)d_Yn.p, Gn,
                           ystride_n, ops_n, 0, (double *)d_Q.p, (double *)nullptr, S, nch, d.noise_out);
        HIPCHK(this, mark(4, st));
    }
    HIPCHK(this, hipEventRecord(t.front_done, st));

    // =========================== POST (post_stream): sums, squelch stage 2, window kernel ===========================
    hipStream_t ps = pipelined ? post_stream : st;
    if (pipelined) HIPCHK(this, hipStreamWaitEvent(ps, t.front_done, 0));
    // deferred squelch: the sums, stage 2 and the per-window SNR go to the side stream (behind this batch's banks), the window
    // kernel follows the banks directly on `ps`
    // (round 6) squelch stage 2 + block sums on the side stream as well where presence and the exact rows stand between the banks and
    // the window kernel: they do not depend on each other, and the exact rows' kernel leaves a fifth of the issue cycles idle
    static const bool side_sq_off = getenv("BTGPU_SQ_INLINE") != nullptr;        // (A/B)
    const bool side_sq = !deferred && !pipelined && verify == 1 && use_pfb && use_staged && !side_sq_off;
    hipStream_t qs = (deferred || side_sq) ? sq_stream : ps;
    if (deferred || side_sq) HIPCHK(this, hipStreamWaitEvent(qs, t.front_done, 0));
    // ---- window parameters, and the exact stage's burst scan right behind the banks (round 5) ----
    WindowParams p = make_window_params(des, S, nb, ystride, max_hits, want_syms, (const uint64_t *)d_pcol.p);
    { static const int ws = getenv("BTGPU_WIN_STOP") ? atoi(getenv("BTGPU_WIN_STOP")) : 0; p.dbg_stop = ws; }
    { static const int fp_ = getenv("BTGPU_FIN_PRIO") ? atoi(getenv("BTGPU_FIN_PRIO")) : 3; p.fin_prio = fp_; }
    p.want_len = no_nsym ? 0 : 1;
    p.deferred = deferred ? 1 : 0;
    p.snr_arr = (const double *)t.d_snr.p;
    // exact rows (exact.hip.h): presence marks the busy windows' rows (bm1), exact_rows_kernel recomputes them IN PLACE in the
    // demodulated stream, in line, before the window kernel reads it
    VerifyBuffers vb;
    if (verify) {
        vb.tasks = (VerifyTask *)t.d_vtasks.p; vb.vcount = (unsigned int *)t.d_vcount.p;
        vb.dxt = (float *)t.d_dxt.p; vb.vcap = vcap;
        vb.bm_tiles = exact_ntiles(G); vb.bm1 = (uint32_t *)t.d_bm.p; vb.bm2 = vb.bm1 + (size_t)bm_tiles * kExBmWords;
        if (pfb_small && t.d_pfine.p && verify_has_fine(des, fp, drow))
            set_verify_flagging(p, des, fp, pfb_small, verify, (const double *)t.d_pfine.p, ntiles * (pfbm_tile(fp.channel.M) / 25), vb, want_syms, 25);
        else
            set_verify_flagging(p, des, fp, pfb_small, verify, (const double *)t.d_ptile.p, ntiles, vb, want_syms);
        HIPCHK(this, hipMemsetAsync(t.d_bm.p, 0, (size_t)2 * bm_tiles * kExBmWords * sizeof(uint32_t), ps));
    }
    auto launch_exact_rows = [&](const uint32_t *bitmap, unsigned int *stat, hipStream_t s_) {
        const ExactParams ep = make_exact_params(des, x_len, w0, G, (const float *)d_tapsA.p, (const float2 *)d_rot_ch.p, (const float *)d_atan.p,
                                                 bitmap, vb.bm_tiles, (float *)d_d.p, drow, (float *)(use_dcol ? t.d_dcol.p : nullptr), stat);
        hipLaunchKernelGGL(ex_kern, dim3((unsigned)vb.bm_tiles), dim3(kExThreads), ex_lds, s_, ep, d_x);
    };
    if (verify && exact_all) {
        hipLaunchKernelGGL(exact_mark_all_kernel, dim3((unsigned)((vb.bm_tiles * kExBmWords + 255) / 256)), dim3(256), 0, ps, vb.bm1, vb.bm_tiles, nch);
        HIPCHK(this, mark(12, ps));
        launch_exact_rows(vb.bm1, vb.vcount + 4, ps);
        HIPCHK(this, mark(13, ps));
    } else if (verify && p.verify == 1) {
        // presence's last-resort noise reference: each channel's quietest full tile of the batch
        HIPCHK(this, hipMemsetAsync(t.d_chanfloor.p, 0x7f, 81 * sizeof(float), ps));       // (0x7f7f7f7f = 3.4e38: no tile yet)
        const int full_tiles = (G % p.tile_outs) ? p.ptile_stride - 1 : p.ptile_stride;    // (the batch's last tile may be a partial one: ADVICE r5)
        hipLaunchKernelGGL(channel_floor_kernel, dim3((unsigned)std::max(1, std::min(256, p.ptile_stride / 2048)), (unsigned)nch), dim3(256), 0, ps, p.ptile, p.ptile_stride,
                           std::max(1, full_tiles), (float *)t.d_chanfloor.p);
        p.chan_floor = (const float *)t.d_chanfloor.p;
        auto launch_presence = [&](auto lay) {
            using LAY = decltype(lay);
            hipLaunchKernelGGL(presence_kernel<LAY>, dim3((S + LAY::kSlots - 1) / LAY::kSlots), dim3(kWinThreads), 0, ps, p);
        };
        if (drow == 80) launch_presence(WinLayout<3, 96, 20>{});
        else if (drow == 40) launch_presence(WinLayout<6, 40, 10>{});
        else if (drow == 20) launch_presence(WinLayout<12, 20, 5>{});
        else if (drow == 8) launch_presence(WinLayout<32, 8, 2>{});
        else launch_presence(WinLayout<64, 4, 1>{});
        HIPCHK(this, mark(12, ps));
        launch_exact_rows(vb.bm1, vb.vcount + 4, ps);
        HIPCHK(this, mark(13, ps));
    } else { HIPCHK(this, mark(12, ps)); HIPCHK(this, mark(13, ps)); }
    HIPCHK(this, mark(5, qs));
    // tile sums -> block sums: as extra rows of the squelch stage-2 launch where both exist (a kernel of its own costs 0.05 ms
    // of launch ramp and tail for microseconds of work; on a side stream it saved those and cost 0.4 ms per step in
    // cross-stream dependencies -- 71.6 -> 59.5 Gsamples/s, profiles/r03_h_*)
    const bool fused_sums = use_pfb && use_staged;
    if (use_pfb && !fused_sums)
        hipLaunchKernelGGL(block_sum_kernel, dim3((nb * nch + 3) / 4), dim3(256), 0, qs,
                           (const double *)t.d_ptile.p, (const double *)t.d_phead.p, ntiles, tiles_per_block,
                           tail_tiles, (double *)d_P.p, (double *)d_Pt.p, nb, nch);
    HIPCHK(this, mark(6, qs));
    if (use_staged) {
        const size_t lds2 = s2_lds_bytes(ns.outs, ns.nw, ns.L3);
        BlockSumArgs bsa;
        if (fused_sums) {
            bsa.ptile = (const double *)t.d_ptile.p; bsa.phead = (const double *)t.d_phead.p; bsa.ntiles = ntiles;
            bsa.tiles_per_block = tiles_per_block; bsa.tail_tiles = tail_tiles; bsa.P = (double *)d_P.p; bsa.Pt = (double *)d_Pt.p;
            bsa.nb = nb; bsa.nch = nch; bsa.rows = kS2SumRows;
        }
        hipLaunchKernelGGL(noise_stage2_kernel, dim3((S + kS2Slots - 1) / kS2Slots, nch + bsa.rows), dim3(256), lds2, qs,
                           (const float2 *)t.d_Z.p, zstride, ns.outs, ns.nw, ns.L3, (const float *)d_h3.p,
                           (const double *)d_w.p, (double *)d_Q.p, S, bsa);
    }
    HIPCHK(this, mark(14, qs));
    if (side_sq) { HIPCHK(this, hipEventRecord(t.squelch_done, qs)); HIPCHK(this, hipStreamWaitEvent(ps, t.squelch_done, 0)); }
    HIPCHK(this, mark(7, deferred ? qs : ps));

    // ---- K3: squelch + M&M + slicer + access-code search ----
    {
        if (deferred) {
            hipLaunchKernelGGL(squelch_kernel, dim3((unsigned)(((long long)S * nch + 255) / 256)), dim3(256), 0, qs, p, (const double *)d_P.p,
                               (const double *)d_Pt.p, (const double *)d_Q.p, (double *)t.d_eon.p, (double *)t.d_eoff.p, (double *)t.d_snr.p);
            HIPCHK(this, hipEventRecord(t.squelch_done, qs));
        }
        auto launch_window = [&](auto lay) {
            using LAY = decltype(lay);
            hipLaunchKernelGGL(window_kernel<LAY>, dim3((S + LAY::kSlots - 1) / LAY::kSlots), dim3(kWinThreads), 0, ps, p,
                               (const float *)d_d.p, G,
                               (const double *)d_P.p, (const double *)d_Pt.p, (const double *)d_Q.p,
                               (const float *)d_mmse.p, (const uint64_t *)d_aclo.p, (const uint32_t *)d_achi.p,
                               (double *)t.d_eon.p, (double *)t.d_eoff.p, (double *)t.d_snr.p, (int *)d_winlen.p,
                               (DeviceHit *)d_hits.p, (unsigned int *)d_hitcount.p, (FinishRec *)d_fin.p,
                               (unsigned int *)d_hitcount.p + 1, (const uint8_t *)d_le_hdr.p,
                               (const uint16_t *)d_le_whiten.p, (const int8_t *)d_le_index.p, (int *)d_winfin.p,
                               (uint32_t *)d_symbits.p, (uint32_t *)d_winbits.p);
        };
        // layout by channel count (kernels.hip.h): as many slots per workgroup as fill its 256 lanes
        // (BTGPU_WIN_ROWS=29: 29 rows per chunk = 39.8 KB of LDS, four workgroups per CU instead of three, for A/B -- no faster)
        static const bool win29 = [] { const char *e = getenv("BTGPU_WIN_ROWS"); return e && atoi(e) == 29; }();
        // (deferred squelch: the 39.8 KB layout -- three window workgroups leave a CU 40 KB for a stage-2 workgroup beside them)
        if (drow == 80 && (win29 || deferred)) launch_window(WinLayout<3, 96, 20, kWinRowsSmall>{});
        else if (drow == 80) launch_window(WinLayout<3, 96, 20>{});
        else if (drow == 40) launch_window(WinLayout<6, 40, 10>{});
        else if (drow == 20) launch_window(WinLayout<12, 20, 5>{});
        else if (drow == 8) launch_window(WinLayout<32, 8, 2>{});
        else launch_window(WinLayout<64, 4, 1>{});
        HIPCHK(this, mark(8, ps));
        // =========================== TAIL (tail_stream): finish + nsym + record copies ===========================
        HIPCHK(this, hipEventRecord(t.detect_done, ps));
        HIPCHK(this, hipStreamWaitEvent(tail_stream, t.detect_done, 0));
        if (deferred) HIPCHK(this, hipStreamWaitEvent(tail_stream, t.squelch_done, 0));
        HIPCHK(this, mark(9, tail_stream));
        if (verify) {
            // the second run: the rows under the first run's uncovered hits (bm2), then those windows again
            launch_exact_rows(vb.bm2, vb.vcount + 5, tail_stream);
            const VerifyFillParams fz = make_verify_fill_params(des, (const float *)d_d.p, (const float *)(use_dcol ? t.d_dcol.p : nullptr),
                                                                drow, G, vb);
            hipLaunchKernelGGL(verify_fill_kernel, dim3(kVerGridFill), dim3(256), 0, tail_stream, fz);
            const WindowParams pv = make_verify_window_params(p, vb);
            auto launch_exact = [&](auto lay) {
                using LAY = decltype(lay);
                hipLaunchKernelGGL((window_kernel<LAY, true>), dim3((pv.S + LAY::kSlots - 1) / LAY::kSlots), dim3(kWinThreads), 0, tail_stream, pv,
                                   (const float *)t.d_dxt.p, (long long)pv.S * kVerRows,
                                   (const double *)d_P.p, (const double *)d_Pt.p, (const double *)d_Q.p,
                                   (const float *)d_mmse.p, (const uint64_t *)d_aclo.p, (const uint32_t *)d_achi.p,
                                   (double *)t.d_eon.p, (double *)t.d_eoff.p, (double *)t.d_snr.p, (int *)d_winlen.p,
                                   (DeviceHit *)d_hits.p, (unsigned int *)d_hitcount.p, (FinishRec *)d_fin.p,
                                   (unsigned int *)d_hitcount.p + 1, (const uint8_t *)d_le_hdr.p,
                                   (const uint16_t *)d_le_whiten.p, (const int8_t *)d_le_index.p, (int *)d_winfin.p,
                                   (uint32_t *)d_symbits.p, (uint32_t *)t.d_winbits_v.p);
            };
            if (drow == 80) launch_exact(WinLayout<3, 96, 20>{});
            else if (drow == 40) launch_exact(WinLayout<6, 40, 10>{});
            else if (drow == 20) launch_exact(WinLayout<12, 20, 5>{});
            else if (drow == 8) launch_exact(WinLayout<32, 8, 2>{});
            else launch_exact(WinLayout<64, 4, 1>{});
            HIPCHK(this, hipMemcpyAsync(t.h_count + 4, t.d_vcount.p, kVerCountWords * sizeof(unsigned int), hipMemcpyDeviceToHost, tail_stream));
        }
        if (timing_on && timing_full) HIPCHK(this, hipEventRecord(ev[11], tail_stream));
        {
            // windows with hits: at most one FinishRec per window; lanes beyond fin_count exit
            // (the kernel strides over the records: the grid only bounds the waves in flight)
            const long long cap = (long long)S * nch;
            const unsigned nblk = (unsigned)std::min<long long>((cap + kFinLanes - 1) / kFinLanes, 4096);
            static const bool tail_off = getenv("BTGPU_TAIL_OFF") != nullptr;   // timing experiments only: records lose nsym
            if (!tail_off && (!no_nsym || want_syms)) {
            if (want_syms)
                hipLaunchKernelGGL(finish_kernel<true>, dim3(nblk), dim3(kFinLanes), 0, tail_stream, p, (const float *)d_d.p,
                                   drow, G, (const float *)d_mmse.p, (const FinishRec *)d_fin.p,
                                   (const unsigned int *)d_hitcount.p + 1, (int *)d_winlen.p, (uint32_t *)d_symbits.p,
                                   (const float *)(use_dcol ? t.d_dcol.p : nullptr));
            else
                hipLaunchKernelGGL(finish_kernel<false>, dim3(nblk), dim3(kFinLanes), 0, tail_stream, p, (const float *)d_d.p,
                                   drow, G, (const float *)d_mmse.p, (const FinishRec *)d_fin.p,
                                   (const unsigned int *)d_hitcount.p + 1, (int *)d_winlen.p, (uint32_t *)nullptr,
                                   (const float *)(use_dcol ? t.d_dcol.p : nullptr));
            }
            hipLaunchKernelGGL(nsym_patch_kernel, dim3(32), dim3(256), 0, tail_stream, (DeviceHit *)d_hits.p,
                               (const unsigned int *)d_hitcount.p, max_hits, (const int *)d_winlen.p, nch,
                               want_syms ? (const int *)d_winfin.p : (const int *)nullptr,
                               deferred ? (const double *)t.d_snr.p : (const double *)nullptr, des.cfg.squelch_db);
            if (want_hdrs) {
                const btgpu_design &dd = des.d;
                hipLaunchKernelGGL(header_sweep_kernel, dim3(1024), dim3(64), 0, tail_stream, (const DeviceHit *)d_hits.p,
                                   (const unsigned int *)d_hitcount.p, max_hits, (const uint32_t *)d_symbits.p,
                                   (const uint32_t *)d_wh18.p, dd.correlator == BTGPU_CORRELATOR_BTBB ? 68 : 72,
                                   (HeaderRec *)t.d_hdr.p);
            }
            // records travel to page-locked host memory on the tail stream too -- through a kernel that knows the counts (the
            // first eager_fin hit windows' symbols, eager_hdr sweeps, eager_hits records; beyond: harvest's spill copies) --
            // harvesting a batch is then pure host work and never waits on the other streams
            HIPCHK(this, hipMemcpyAsync(t.h_count, d_hitcount.p, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, tail_stream));
            hipLaunchKernelGGL(records_out_kernel, dim3(128), dim3(256), 0, tail_stream, (const unsigned int *)d_hitcount.p, max_hits,
                               (const uint4 *)(want_syms ? d_symbits.p : nullptr), (uint4 *)(want_syms ? t.h_sym : nullptr), eager_fin,
                               (const uint2 *)(want_hdrs ? t.d_hdr.p : nullptr), (uint2 *)(want_hdrs ? t.h_hdr : nullptr), eager_hdr,
                               (const uint4 *)d_hits.p, (uint4 *)t.h_hits, std::min<unsigned>((unsigned)max_hits, eager_hits));
        }
    }
    HIPCHK(this, mark(10, tail_stream));
    HIPCHK(this, hipEventRecord(t.tail_done, tail_stream));
    HIPCHK(this, hipGetLastError());
    t.pending = true;
    last_ctx = cur;
    cur = (cur + 1) % nctx;
    if (!async) { const int hrc = harvest(t); return hrc != BTGPU_OK ? hrc : carried; }
    return carried;
}

// One batch from host memory: [n_head_zero zeros | body] (complex samples) -> pinned buffer -> device buffer ->
// kernels.  `head` (may be null) supplies real samples for the first n_head samples instead of zeros when the
// caller has them (the margin carried between btgpu_work calls).  Buffers alternate; a buffer is reused only
// after the batch that read it has left the main stream.
int btgpu_handle::stage_batch(const float *head, size_t n_head, const float *body, size_t n_body, long long w0,
                              uint64_t abs_first_slot, int S)
{
    const unsigned k = stage_turn++ & 1u;
    const size_t cap = in_cap + (size_t)margin + 64;
    if (n_head + n_body > cap) { set_error("staging overflow"); return BTGPU_EINVAL; }
    if (!h_stage[k]) {
        if (hipHostMalloc((void **)&h_stage[k], cap * sizeof(float2), hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc (pinned staging buffer)"); return BTGPU_ENOMEM;
        }
        HIPCHK(this, hipEventCreateWithFlags(&ev_copied[k], hipEventDisableTiming));
        HIPCHK(this, hipEventCreateWithFlags(&ev_consumed[k], hipEventDisableTiming));
        HIPCHK(this, hipEventCreateWithFlags(&ev_vdone[k], hipEventDisableTiming));
        if (k == 1 && !d_in_b.p) { int rc = alloc(d_in_b, cap * sizeof(float2)); if (rc) return rc; }
    }
    if (stage_used[k]) {                                                         // the batch that used this pair is done with it
        HIPCHK(this, hipEventSynchronize(ev_consumed[k]));
        if (verify) HIPCHK(this, hipEventSynchronize(ev_vdone[k]));
    }
    float *dst = (float *)h_stage[k];
    if (n_head) {
        if (head) std::memcpy(dst, head, n_head * sizeof(float2));
        else std::memset(dst, 0, n_head * sizeof(float2));
    }
    float2 *d_buf = (float2 *)(k == 0 ? d_in.p : d_in_b.p);
    // A caller's buffer that is page-locked already (hipHostMalloc / hipHostRegister: what a block can do once with the
    // scheduler's buffer) is copied to the device as it lies -- no pass through the staging buffer; the call
    // then returns when that copy is done (the buffer is the caller's again), the kernels of this and the previous batches
    // run on regardless.  Pageable memory goes through the pinned staging buffer, the host copy split over a few threads (one
    // thread moves ~10 GB/s, a PCIe Gen5 link takes 50).
    bool direct = false;
    {
        hipPointerAttribute_t attr;
        // page-locked from its first byte to its last (a buffer registered only in part would be copied as if all of it were: ADVICE r4)
        if (n_body > 0 && hipPointerGetAttributes(&attr, body) == hipSuccess && attr.type == hipMemoryTypeHost) {
            hipPointerAttribute_t last;
            if (hipPointerGetAttributes(&last, (const char *)body + n_body * sizeof(float2) - 1) == hipSuccess && last.type == hipMemoryTypeHost) direct = true;
            else (void)hipGetLastError();
        } else (void)hipGetLastError();                                      // (an unregistered pointer is reported as an error: cleared)
        static const bool no_direct = getenv("BTGPU_NO_DIRECT_H2D") != nullptr;
        if (no_direct) direct = false;
    }
    if (direct) {
        if (n_head) HIPCHK(this, hipMemcpyAsync(d_buf, h_stage[k], n_head * sizeof(float2), hipMemcpyHostToDevice, copy_stream));
        HIPCHK(this, hipMemcpyAsync(d_buf + n_head, body, n_body * sizeof(float2), hipMemcpyHostToDevice, copy_stream));
    } else {
        const size_t bytes = n_body * sizeof(float2);
        const unsigned nthr = bytes >= ((size_t)64 << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (nthr <= 1) std::memcpy(dst + 2 * n_head, body, bytes);
        else {
            std::vector<std::thread> pool;
            const size_t chunk = (bytes / nthr + 4095) & ~(size_t)4095;
            for (unsigned i = 0; i < nthr; i++) {
                const size_t o = (size_t)i * chunk;
                if (o >= bytes) break;
                const size_t len = std::min(chunk, bytes - o);
                pool.emplace_back([=] { std::memcpy((char *)(dst + 2 * n_head) + o, (const char *)body + o, len); });
            }
            for (auto &th : pool) th.join();
        }
        HIPCHK(this, hipMemcpyAsync(d_buf, h_stage[k], (n_head + n_body) * sizeof(float2), hipMemcpyHostToDevice, copy_stream));
    }
    HIPCHK(this, hipEventRecord(ev_copied[k], copy_stream));
    HIPCHK(this, hipStreamWaitEvent(stream, ev_copied[k], 0));
    const int rc = process_batch(d_buf, n_head + n_body, w0, abs_first_slot, S, stream);
    // (behind the enqueue: the kernels queue up under the copy; also where process_batch failed -- the buffer is the caller's again
    // only once the copy has left it)
    if (direct) { const hipError_t se = hipEventSynchronize(ev_copied[k]); if (rc != BTGPU_OK) return rc; HIPCHK(this, se); }
    HIPCHK(this, hipEventRecord(ev_consumed[k], stream));
    if (verify) HIPCHK(this, hipEventRecord(ev_vdone[k], last_tail));
    stage_used[k] = true;
    return rc;
}

// wait for a batch's tail, move its hit records to the host queue, account its kernel times
int btgpu_handle::harvest(TailCtx &t)
{
    if (!t.pending) return BTGPU_OK;
    const btgpu_design &d = des.d;
    HIPCHK(this, hipEventSynchronize(t.tail_done));
    t.pending = false;
    if (timing_on) {
        // BTGPU_K_*: channel bank | demod + energy (direct) or tile sums -> block sums (polyphase) | noise stage 1 or
        // direct noise bank | squelch stage 2 or direct noise energy | window | tail
        float ms = 0;
        const bool blk = use_pfb;
        // (deferred squelch: the window kernel starts behind the banks' last mark, not behind stage 2, which runs beside it)
        hipEvent_t a[7] = {t.ev[0], blk ? t.ev[5] : t.ev[1], t.ev[2], use_staged ? t.ev[6] : t.ev[3], deferred ? t.ev[4] : t.ev[7], t.ev[11], verify ? t.ev[8] : t.ev[9]};   // (the exact stage from the end of the window kernel: its DDC runs in line, before the tail's first mark -- ADVICE r4)
        hipEvent_t e[7] = {t.ev[1], blk ? t.ev[6] : t.ev[2], t.ev[3], use_staged ? t.ev[14] : t.ev[4], t.ev[8], t.ev[10], t.ev[11]};
        for (int i = 0; i < (timing_full ? 7 : 1); i++) {
            HIPCHK(this, hipEventElapsedTime(&ms, a[i], e[i]));
            timing.kernel_ms[i] += ms;
            timing.kernel_launches[i] += 1;
        }
        { HIPCHK(this, hipEventElapsedTime(&ms, t.ev[12], t.ev[13])); timing.kernel_ms[BTGPU_K_EXACT] += ms; timing.kernel_launches[BTGPU_K_EXACT] += 1; }   // exact_rows_kernel over presence's marks, in line
        if (timing_full) { HIPCHK(this, hipEventElapsedTime(&ms, t.ev[0], t.ev[10])); timing.total_ms += ms; }
    }
    timing.batches += 1;
    if (verify) {
        // h_count[4..]: vcount -- tasks of the second run, pairs marked, turned away, busy windows, pairs computed by the two launches
        timing.verify_windows += t.h_count[7] + std::min<unsigned>(t.h_count[4], (unsigned)vcap);
        timing.verify_rows += (uint64_t)(t.h_count[8] + t.h_count[9]) * kExSlotRows / kExSlotTiles;   // (eleven tiles per 1250-row slot)
        timing.verify_turned_away += t.h_count[6];
        timing.long_tasks += std::min<unsigned>(t.h_count[4], (unsigned)vcap);      // (reused: the second run's windows and rows)
        timing.long_rows += (uint64_t)t.h_count[9] * kExSlotRows / kExSlotTiles;
    }
    timing.slots += (uint64_t)t.S;
    timing.samples += (uint64_t)t.S * (uint64_t)d.samples_per_slot;

    unsigned int count = t.h_count[0];
    int rc = BTGPU_OK;
    if (count > (unsigned)max_hits) { count = (unsigned)max_hits; rc = BTGPU_EOVERFLOW; sticky = rc; set_error("hit buffer overflow"); }
    if (count) {
        // what the eager copies do not carry (more records, sweeps or hit windows in one batch than the page-locked buffers hold: dense
        // captures at the small rates) comes over in ONE bulk copy each, into page-locked memory, and one wait for the three
        const unsigned nfin = t.h_count[1];
        const size_t n_hit_spill = count > eager_hits ? count - eager_hits : 0;
        const size_t n_sym_spill = (want_syms && nfin > eager_fin) ? (size_t)(nfin - eager_fin) : 0;
        const size_t n_hdr_spill = (want_hdrs && count > eager_hdr) ? (size_t)(count - eager_hdr) : 0;
        const DeviceHit *hit_spill = nullptr; const uint32_t *sym_spill = nullptr; const HeaderRec *hdr_spill = nullptr;
        if (n_hit_spill) {
            void *b = spill_buf(0, n_hit_spill * sizeof(DeviceHit));
            if (!b) { set_error("hipHostMalloc (record spill)"); return BTGPU_ENOMEM; }
            HIPCHK(this, hipMemcpyAsync(b, (const DeviceHit *)t.d_hits.p + eager_hits, n_hit_spill * sizeof(DeviceHit), hipMemcpyDeviceToHost, spill_stream));
            hit_spill = (const DeviceHit *)b;
        }
        if (n_sym_spill) {
            void *b = spill_buf(1, n_sym_spill * kSymWords * sizeof(uint32_t));
            if (!b) { set_error("hipHostMalloc (symbol spill)"); return BTGPU_ENOMEM; }
            HIPCHK(this, hipMemcpyAsync(b, (const uint32_t *)t.d_symbits.p + (size_t)eager_fin * kSymWords, n_sym_spill * kSymWords * sizeof(uint32_t), hipMemcpyDeviceToHost, spill_stream));
            sym_spill = (const uint32_t *)b;
        }
        if (n_hdr_spill) {
            void *b = spill_buf(2, n_hdr_spill * sizeof(HeaderRec));
            if (!b) { set_error("hipHostMalloc (header spill)"); return BTGPU_ENOMEM; }
            HIPCHK(this, hipMemcpyAsync(b, (const HeaderRec *)t.d_hdr.p + eager_hdr, n_hdr_spill * sizeof(HeaderRec), hipMemcpyDeviceToHost, spill_stream));
            hdr_spill = (const HeaderRec *)b;
        }
        if (n_hit_spill || n_sym_spill || n_hdr_spill) HIPCHK(this, hipStreamSynchronize(spill_stream));
        auto hit_at = [&](size_t i) -> const DeviceHit & { return i < eager_hits ? t.h_hits[i] : hit_spill[i - eager_hits]; };
        // the order the reference's loops print in -- slot, channel, kind, offset -- formed FIRST, on packed keys; the records then go
        // to the queue's arenas in that order, once
        std::vector<std::pair<uint64_t, uint32_t>> order;
        order.reserve(count);
        for (size_t i = 0; i < count; i++) {
            const DeviceHit &x = hit_at(i);
            if (x.kind < 0) continue;                              // deferred squelch: the window failed it, the reference never looked
            const uint64_t key = ((uint64_t)(uint32_t)x.slot << 40) | ((uint64_t)((uint32_t)x.channel_idx & 0xffu) << 32) |
                                 ((uint64_t)((uint32_t)x.kind & 0xffu) << 24) | (uint64_t)((uint32_t)x.offset & 0xffffffu);
            order.emplace_back(key, (uint32_t)i);
        }
        std::sort(order.begin(), order.end());
        const size_t q0 = queue.size(), n = order.size();
        queue.resize(q0 + n);
        if (want_syms) { qbits.resize((q0 + n) * kSymWords); qhas.resize(q0 + n); }
        if (want_hdrs) qhdr.resize(q0 + n);
        for (size_t k = 0; k < n; k++) {
            const size_t hit_index = order[k].second;
            const DeviceHit &x = hit_at(hit_index);
            btgpu_hit o{};
            o.slot = t.abs_first_slot + x.slot;
            o.channel = d.low_channel + x.channel_idx;
            o.offset = x.offset;
            o.lap = x.lap;
            o.ac_errors = x.ac_errors;
            o.kind = x.kind;
            o.nsym = x.nsym;
            o.snr_db = x.snr;
            queue[q0 + k] = o;
            if (want_syms) {
                const uint32_t *src = nullptr;
                if (x.sym >= 0) {
                    if ((unsigned)x.sym < eager_fin) src = t.h_sym + (size_t)x.sym * kSymWords;
                    else if ((size_t)((unsigned)x.sym - eager_fin) < n_sym_spill) src = sym_spill + (size_t)((unsigned)x.sym - eager_fin) * kSymWords;
                }
                qhas[q0 + k] = src != nullptr;
                if (src) std::memcpy(&qbits[(q0 + k) * kSymWords], src, kSymWords * sizeof(uint32_t));
            }
            if (want_hdrs) {
                btgpu_header &hd = qhdr[q0 + k];
                std::memset(&hd, 0, sizeof hd);
                const HeaderRec &r = hit_index < eager_hdr ? t.h_hdr[hit_index] : hdr_spill[hit_index - eager_hdr];
                std::memcpy(hd.uap, r.uap, 64); std::memcpy(hd.type, r.type, 64); hd.fec13_ok = r.fec13_ok;
            }
        }
    }
    return rc;
}

// harvest pending batches in submission order; block = false takes only the finished ones
int btgpu_handle::harvest_all(bool block)
{
    int rc_all = BTGPU_OK;
    for (int i = 0; i < nctx; i++) {
        TailCtx &t = tc[(cur + i) % nctx];          // tc[cur] is the oldest one
        if (!t.pending) continue;
        if (!block && hipEventQuery(t.tail_done) != hipSuccess) break;
        int rc = harvest(t);
        if (rc == BTGPU_EOVERFLOW) rc_all = rc;
        else if (rc != BTGPU_OK) return rc;
    }
    return rc_all;
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

const char *btgpu_version(void) { return "btgpu 0.1 (gfx950)"; }

const char *btgpu_strerror(int code)
{
    switch (code) {
        case BTGPU_OK: return "ok";
        case BTGPU_EINVAL: return "invalid argument or unsupported configuration";
        case BTGPU_ENOMEM: return "out of memory";
        case BTGPU_EDEVICE: return "HIP runtime error";
        case BTGPU_ENODEVICE: return "no gfx950 device available";
        case BTGPU_EOVERFLOW: return "hit buffer overflow";
        case BTGPU_EUNSUPPORTED: return "configuration not supported by the GPU path";
        default: return "unknown error";
    }
}

int btgpu_design_query(const btgpu_config *cfg, btgpu_design *out)
{
    if (!cfg || !out) return BTGPU_EINVAL;
    Design *d = new (std::nothrow) Design();
    if (!d) return BTGPU_ENOMEM;
    int rc = make_design(*cfg, *d);
    if (rc == BTGPU_OK) *out = d->d;
    delete d;
    return rc;
}

int btgpu_acgen(uint32_t lap, uint8_t ac[9])
{
    if (!ac) return BTGPU_EINVAL;
    access_code_bytes(lap, ac);
    return BTGPU_OK;
}

int btgpu_filter_taps(const btgpu_config *cfg, int which, float *taps, int cap)
{
    if (!cfg) return BTGPU_EINVAL;
    std::vector<float> h = which == 0 ? firdes_low_pass_hann(1.0, cfg->sample_rate, 500000.0, 300000.0)
                                      : firdes_low_pass_hann(1.0, cfg->sample_rate, 22500.0, 10000.0);
    if (!taps) return (int)h.size();
    if (cap < (int)h.size()) return BTGPU_EINVAL;
    std::memcpy(taps, h.data(), h.size() * sizeof(float));
    return (int)h.size();
}

/* host-side view of the constant tables the kernels use (CPU tests) */
int btgpu_debug_tables(const btgpu_config *cfg, float *mmse /*129*8*/, float *atan_tab /*257*/,
                       uint32_t lap, uint64_t *ac_lo, uint32_t *ac_hi)
{
    if (!cfg) return BTGPU_EINVAL;
    Design *d = new (std::nothrow) Design();
    if (!d) return BTGPU_ENOMEM;
    int rc = make_design(*cfg, *d);
    if (rc == BTGPU_OK) {
        if (mmse) std::memcpy(mmse, d->mmse, sizeof d->mmse);
        if (atan_tab) std::memcpy(atan_tab, d->atan_tab, sizeof d->atan_tab);
        lap &= 0xffffff;
        if (ac_lo) *ac_lo = d->ac.a0_lo ^ d->ac.byte_lo[0][lap & 0xff] ^ d->ac.byte_lo[1][(lap >> 8) & 0xff] ^
                            d->ac.byte_lo[2][lap >> 16];
        if (ac_hi) *ac_hi = d->ac.a0_hi ^ d->ac.byte_hi[0][lap & 0xff] ^ d->ac.byte_hi[1][(lap >> 8) & 0xff] ^
                            d->ac.byte_hi[2][lap >> 16];
    }
    delete d;
    return rc;
}

/* the regenerated integer tables the kernels use, by the reference's names / by the derived forms of
 * tests/golden/make_lut_digests.py (CPU tests: digests against the reference's literals) */
int btgpu_debug_lut(const char *name, void *out, int cap_bytes)
{
    if (!name || !out) return BTGPU_EINVAL;
    Design *d = new (std::nothrow) Design();
    if (!d) return BTGPU_ENOMEM;
    btgpu_config cfg{};
    cfg.sample_rate = 8e6; cfg.center_freq = 2476.5e6; cfg.squelch_db = 10.0; cfg.mode = BTGPU_MODE_SNIFFER;
    int rc = make_design(cfg, *d);
    if (rc == BTGPU_OK) {
        const void *src = nullptr; int n = 0;
        const std::string k(name);
        if (k == "le_packet::ACCESS_HEADER_DISTANCE_LSB") { src = d->le.hdr[0]; n = 256; }
        else if (k == "le_packet::ACCESS_HEADER_DISTANCE_MSB") { src = d->le.hdr[1]; n = 256; }
        else if (k == "le_packet::DATA_HEADER_DISTANCE_LSB") { src = d->le.hdr[2]; n = 256; }
        else if (k == "le_packet::DATA_HEADER_DISTANCE_MSB") { src = d->le.hdr[3]; n = 256; }
        else if (k == "derived/classic_first18") { src = d->wh.first18; n = (int)sizeof d->wh.first18; }
        else if (k == "derived/le_whiten16") { src = d->le.whiten16; n = (int)sizeof d->le.whiten16; }
        if (!src || cap_bytes < n) rc = BTGPU_EINVAL;
        else { std::memcpy(out, src, (size_t)n); rc = n; }
    }
    delete d;
    return rc;
}

/* host-side view of the staged squelch design (CPU tests): composite-filter fit error etc. */
int btgpu_debug_staged_design(const btgpu_config *cfg, double *fit_l1_error, int *R, int *L1, int *L3, int *nw,
                              double *weight_sum)
{
    if (!cfg) return BTGPU_EINVAL;
    Design *d = new (std::nothrow) Design();
    FastPath *fp = new (std::nothrow) FastPath();
    if (!d || !fp) { delete d; delete fp; return BTGPU_ENOMEM; }
    int rc = make_design(*cfg, *d);
    if (rc == BTGPU_OK) {
        (void)make_fast_path(*d, *fp);
        if (!fp->noise.available) rc = BTGPU_EUNSUPPORTED;
        else {
            if (fit_l1_error) *fit_l1_error = fp->noise.fit_l1_error;
            if (R) *R = fp->noise.R;
            if (L1) *L1 = fp->noise.L1;
            if (L3) *L3 = fp->noise.L3;
            if (nw) *nw = fp->noise.nw;
            if (weight_sum) { double s = 0; for (double w : fp->noise.weights) s += w; *weight_sum = s; }
        }
    }
    delete d; delete fp;
    return rc;
}

int btgpu_create(const btgpu_config *cfg, btgpu_handle **out)
{
    if (!cfg || !out) return BTGPU_EINVAL;
    *out = nullptr;
    btgpu_handle *h = new (std::nothrow) btgpu_handle();
    if (!h) return BTGPU_ENOMEM;
    int rc = make_design(*cfg, h->des);
    if (rc != BTGPU_OK) { delete h; return rc; }
    {
        FastPath *fp = &h->fp;
        int frc = make_fast_path(h->des, *fp);
        const int nch0 = h->des.d.high_channel - h->des.d.low_channel + 1;
        // 100 Msps: the 10 x 10 FFT bank (pfb100.hip.h); other even integer rates: the small-M bank (pfbm.hip.h)
        const bool pfb100_ok = fp->channel.available && fp->channel.M == kPfbM && fp->channel.Q == 7 && fp->channel.S == 1 &&
                               h->des.outs_per_slot % 25 == 0;
        // dynamic LDS a workgroup may ask for on the device the handle will live on (gfx950: 160 KB; the banks need up to 96)
        size_t lds_cap = 64 * 1024;
        {
            int dq = cfg->device, mx = 0;
            if (dq < 0 && hipGetDevice(&dq) != hipSuccess) dq = 0;
            if (hipDeviceGetAttribute(&mx, hipDeviceAttributeMaxSharedMemoryPerBlock, dq) == hipSuccess && mx > 0) lds_cap = (size_t)mx;
            if (lds_cap > 96 * 1024) lds_cap = 96 * 1024;          // what hipFuncSetAttribute is asked for below
        }
        const bool pfbm_ok = fp->channel.available && fp->channel.M >= 4 && fp->channel.M < kPfbM &&
                             h->des.outs_per_slot % pfbm_tile(fp->channel.M) == 0 &&
                             pfbm_lds_bytes(fp->channel.M, fp->channel.D, fp->channel.Q, nch0, true) <= lds_cap;
        const bool pfb_ok = !h->des.segmented && (frc == BTGPU_OK || frc == BTGPU_EUNSUPPORTED) && (pfb100_ok || pfbm_ok);
        const bool noise_pfb_ok = fp->noise.available && fp->noise.pfb.available && fp->noise.pfb.M == kPfbM &&
                                  fp->noise.pfb.Q == 15 && fp->noise.pfb.S == 5;
        const bool noise_small_ok = fp->noise.available && fp->noise.pfb.available && fp->noise.pfb.M >= 4 && fp->noise.pfb.M < kPfbM &&
                                    pfbm_lds_bytes(fp->noise.pfb.M, fp->noise.pfb.D, fp->noise.pfb.Q, nch0, false) <= lds_cap;
        const bool staged_ok = !h->des.segmented && fp->noise.available &&
                               (noise_pfb_ok || noise_small_ok || pick_shape(fp->noise.R, fp->noise.direct.ntp, h->shape_s1));
        h->noise_pfb = noise_pfb_ok;
        h->noise_small = noise_small_ok && !getenv("BTGPU_NO_PFBM_NOISE");
        h->fuse_noise = false;
        int ch = cfg->channelizer, sq = cfg->squelch;
        // BTGPU_AUTO=direct: AUTO resolves to the bit-exact direct forms (tests of the host protocol layer that
        // compare printed text with the oracle's, character for character)
        if (getenv("BTGPU_AUTO") && std::strcmp(getenv("BTGPU_AUTO"), "direct") == 0) {
            if (ch == BTGPU_CHANNELIZER_AUTO) ch = BTGPU_CHANNELIZER_DIRECT;
            if (sq == BTGPU_SQUELCH_AUTO) sq = BTGPU_SQUELCH_DIRECT;
        }
        if (ch == BTGPU_CHANNELIZER_AUTO) ch = pfb_ok ? BTGPU_CHANNELIZER_POLYPHASE : BTGPU_CHANNELIZER_DIRECT;
        if (sq == BTGPU_SQUELCH_AUTO) sq = staged_ok ? BTGPU_SQUELCH_STAGED : BTGPU_SQUELCH_DIRECT;
        if ((ch == BTGPU_CHANNELIZER_POLYPHASE && !pfb_ok) || (sq == BTGPU_SQUELCH_STAGED && !staged_ok) ||
            (ch != BTGPU_CHANNELIZER_POLYPHASE && ch != BTGPU_CHANNELIZER_DIRECT) ||
            (sq != BTGPU_SQUELCH_STAGED && sq != BTGPU_SQUELCH_DIRECT)) { delete h; return BTGPU_EUNSUPPORTED; }
        h->use_pfb = ch == BTGPU_CHANNELIZER_POLYPHASE;
        h->pfb_small = h->use_pfb && !pfb100_ok;
        h->use_dcol = h->use_pfb && !h->pfb_small && getenv("BTGPU_NO_DCOL") == nullptr;   // (the knob: A/B timing only)
        h->use_staged = sq == BTGPU_SQUELCH_STAGED;
        h->keep_Y = !h->use_pfb || (cfg->flags & BTGPU_FLAG_DEBUG_Y);
        h->margin = h->use_staged ? kNoiseMargin : 0;
        h->fuse_noise = h->use_pfb && h->use_staged && noise_pfb_ok && fp->channel.D == 50 && fp->noise.R == 250 &&
                        !getenv("BTGPU_NO_FUSE");
        h->des.d.channelizer = ch;
        h->des.d.squelch = sq;
        h->des.d.left_margin = h->margin;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete h; return BTGPU_ENODEVICE; }
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete h; return BTGPU_EINVAL; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return BTGPU_ENODEVICE; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete h; return BTGPU_ENODEVICE; }
    if (hipSetDevice(dev) != hipSuccess) { delete h; return BTGPU_ENODEVICE; }
    h->device = dev;

    const Design &des = h->des;
    const btgpu_design &d = des.d;
    const int nch = d.high_channel - d.low_channel + 1;
    const int ops = des.outs_per_slot;

    auto fail = [&](int code) { h->release(); delete h; return code; };

    if (!pick_shape(d.decimation, des.channel.ntp, h->shape_ch) ||
        !pick_shape(d.decimation, des.noise.ntp, h->shape_n))
        return fail(BTGPU_EUNSUPPORTED);

    // batch size: bounded by ~24 GiB of intermediates (288 GB HBM; larger batches amortise the latency-bound window kernel)
    size_t per_slot = (size_t)nch * ops * ((h->keep_Y ? sizeof(float2) : 0) + (h->use_staged ? 2 : sizeof(float2)) + sizeof(float)) + (size_t)d.samples_per_slot * 8;
    int S = cfg->max_batch_slots > 0 ? cfg->max_batch_slots : 512;
    size_t cap = (size_t)24 << 30;
    if ((size_t)S * per_slot > cap) S = (int)std::max<size_t>(8, cap / per_slot);
    h->max_slots = S;
    h->max_hits = cfg->max_hits > 0 ? cfg->max_hits : std::max(4096, S * nch * 2);

    const long long G = (long long)ops * (S - 1) + d.ddc_out;
    h->ystride = (G + 63) / 64 * 64;
    h->ystride_n = ((long long)(des.segmented ? d.noise_out : ops) * S + 63) / 64 * 64;
    h->nb_max = (int)((G + ops - 1) / ops);
    h->in_cap = (size_t)d.history + (size_t)(S - 1) * d.samples_per_slot;

    bool from_pool = false;
    if (!(getenv("BTGPU_STREAM_POOL") && atoi(getenv("BTGPU_STREAM_POOL")) == 0) && !getenv("BTGPU_TAIL_CUS") && !getenv("BTGPU_TAIL_PRIO") && !getenv("BTGPU_POST_PRIO")) {
        h->streams_poolable = true;
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        for (size_t i = 0; i < g_stream_pool.size(); i++) if (g_stream_pool[i].device == h->device) {
            const StreamSet ss = g_stream_pool[i];
            g_stream_pool.erase(g_stream_pool.begin() + i);
            hipStream_t *all[8] = {&h->stream, &h->post_stream, &h->tail_stream, &h->copy_stream, &h->spill_stream, &h->tail_extra[0], &h->tail_extra[1], &h->sq_stream};
            for (int k = 0; k < 8; k++) *all[k] = ss.s[k];
            from_pool = true;
            break;
        }
    }
    if (!from_pool) {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(BTGPU_EDEVICE);
    {
        // the tail (a few dozen latency-bound waves) gets the highest stream priority so that it is
        // not starved of issue slots by the throughput kernels of the next batch
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // BTGPU_TAIL_CUS=n (experiment): confine the tail to n compute units (every (256/n)-th bit of the CU mask)
        const int tail_cus = getenv("BTGPU_TAIL_CUS") ? atoi(getenv("BTGPU_TAIL_CUS")) : 0;
        if (tail_cus > 0 && tail_cus <= 256) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int step = 256 / tail_cus;
            const int off = getenv("BTGPU_TAIL_CU_OFF") ? atoi(getenv("BTGPU_TAIL_CU_OFF")) : 0;
            for (int i = 0; i < tail_cus; i++) { const int b = (i * step + off) & 255; mask[b >> 5] |= 1u << (b & 31); }
            if (hipExtStreamCreateWithCUMask(&h->tail_stream, 8, mask) != hipSuccess) return fail(BTGPU_EDEVICE);
            // (the tails of consecutive batches alternate between these streams: the same mask on all of them -- ADVICE r4)
            for (auto &te : h->tail_extra) if (hipExtStreamCreateWithCUMask(&te, 8, mask) != hipSuccess) return fail(BTGPU_EDEVICE);
        } else {
            // BTGPU_TAIL_PRIO=lo|mid (A/B): the tail at the lowest / the default stream priority -- measured (round 6, profiles/r06_y_tail_prio_ab.txt):
            // hi 32.9 / 32.7, mid 32.1 / 32.6, lo 32.0 / 32.2 Gsamples/s: the highest stays
            const char *tp = getenv("BTGPU_TAIL_PRIO");
            const int tprio = tp && !strcmp(tp, "lo") ? lo : tp && !strcmp(tp, "mid") ? 0 : hi;
            if (hipStreamCreateWithPriority(&h->tail_stream, hipStreamNonBlocking, tprio) != hipSuccess) return fail(BTGPU_EDEVICE);
            for (auto &te : h->tail_extra) if (hipStreamCreateWithPriority(&te, hipStreamNonBlocking, tprio) != hipSuccess) return fail(BTGPU_EDEVICE);
        }
    }
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(BTGPU_EDEVICE);
    if (hipStreamCreateWithFlags(&h->sq_stream, hipStreamNonBlocking) != hipSuccess) return fail(BTGPU_EDEVICE);
    if (hipStreamCreateWithFlags(&h->spill_stream, hipStreamNonBlocking) != hipSuccess) return fail(BTGPU_EDEVICE);
    {
        // stream of the post stage in the BTGPU_PIPE=1 experiment (BTGPU_POST_PRIO: its priority)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int pp = getenv("BTGPU_POST_PRIO") ? atoi(getenv("BTGPU_POST_PRIO")) : 0;       // 1 high, 0 normal, -1 low
        if (hipStreamCreateWithPriority(&h->post_stream, hipStreamNonBlocking, pp > 0 ? hi : pp < 0 ? lo : 0) != hipSuccess) return fail(BTGPU_EDEVICE);
    }
    }   // (!from_pool)
    for (auto &t : h->tc) {
        for (auto &e : t.ev) if (hipEventCreate(&e) != hipSuccess) return fail(BTGPU_EDEVICE);
        for (hipEvent_t *e : {&t.front_done, &t.detect_done, &t.tail_done, &t.squelch_done, &t.exact_done, &t.floor_done})
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(BTGPU_EDEVICE);
    }
    h->async = (cfg->flags & BTGPU_FLAG_ASYNC) != 0;
    h->nctx = h->async ? btgpu_handle::kCtx : 1;
    if (getenv("BTGPU_CTX")) h->nctx = std::max(1, std::min((int)btgpu_handle::kCtx, atoi(getenv("BTGPU_CTX"))));   // A/B timing only
    h->timing_full = (cfg->flags & BTGPU_FLAG_TIMING) != 0 || getenv("BTGPU_TIMING") != nullptr;
    h->timing_on = h->timing_full || (cfg->flags & BTGPU_FLAG_TIMING_BANK) != 0;
    h->no_nsym = (cfg->flags & BTGPU_FLAG_NO_NSYM) != 0;
    // exact confirmation: on wherever the channelizer is the polyphase one (BTGPU_VERIFY=0 | 1 | 2: A/B timing and tests)
    h->verify = (h->use_pfb && !(cfg->flags & BTGPU_FLAG_NO_VERIFY)) ? 1 : 0;
    if (h->use_pfb && getenv("BTGPU_VERIFY")) h->verify = std::max(0, std::min(2, atoi(getenv("BTGPU_VERIFY"))));
    // (exact rows on the shared grid need every window's rotator to start at exactly +-1 there -- true wherever the polyphase banks
    // exist: their channels sit on bins of fs / M, multiples of 0.5 MHz, i.e. of the 800 Hz a slot's rotation is periodic in)
    if (h->verify && !(exact_rows_available(h->des) && exact_rows_pick(h->des.d.decimation))) {
        if (cfg->channelizer == BTGPU_CHANNELIZER_POLYPHASE && !(cfg->flags & BTGPU_FLAG_NO_VERIFY)) return fail(BTGPU_EUNSUPPORTED);
        h->verify = 0;
    }
    h->ntail = h->verify ? h->nctx : 1;
    if (getenv("BTGPU_TAILS")) h->ntail = std::max(1, std::min(h->nctx, atoi(getenv("BTGPU_TAILS"))));   // A/B timing
    // front(n+1) beside post(n) (BTGPU_PIPE=1; possible only where the front writes nothing but per-context buffers).
    // OFF by default -- measured (profiles/r03_a_*): with today's kernels the overlap LOSES.  Every one of them is
    // occupancy-bound by LDS (bank tile 49.8 KB, window workgroup 51 KB, squelch stage 2 24 KB per workgroup of four
    // waves): a resident window workgroup displaces a bank tile for its whole 0.3-0.6 ms life, so the bank kernel
    // went 1.5 -> 2.7 ms while the post stage went 0.54 -> 1.0 ms: 3.0 ms per step against 2.1 ms in line.  Only
    // the tail (finish_kernel: 12 KB, a few dozen waves) fits beside three bank tiles and keeps overlapping.
    h->pipelined = h->use_pfb && h->use_staged && getenv("BTGPU_PIPE") && atoi(getenv("BTGPU_PIPE")) == 1;
    // Deferred squelch (BTGPU_DEFER=1; OFF by default): squelch stage 2 on a side stream beside the window kernel, the SNR decision
    // applied where records leave.  Built and measured in round 4 (profiles/r04_l_deferred_squelch_ab.txt): identical records,
    // and no gain -- 58.7 against 58.9 Gsamples/s.  Stage 2 is a throughput kernel (packed multiply-adds, 0.21 ms with the
    // device to itself): beside the window kernel and the exact stage's DDC it took 0.93 ms and stretched them by what it saved
    // (window 0.39 -> 0.46, DDC 0.42 -> 0.59 ms).  Possible where the threshold lets (nearly) every window through anyway
    // (white noise alone sits at ~13.5 dB); above ~12 dB skipping the squelched windows' clock recovery pays.
    h->deferred = h->use_pfb && !h->pfb_small && h->use_staged && !h->pipelined && cfg->squelch_db < 12.0 &&
                  getenv("BTGPU_DEFER") && atoi(getenv("BTGPU_DEFER")) == 1;
    h->want_hdrs = (cfg->flags & BTGPU_FLAG_HEADERS) != 0;
    h->want_syms = (cfg->flags & BTGPU_FLAG_SYMBOLS) != 0 || h->want_hdrs;
    h->exact_all = h->verify && (cfg->flags & BTGPU_FLAG_EXACT_ALL) != 0;
    // (BTGPU_FLAG_EXACT_PAYLOAD: accepted, and always on since round 6 -- a packet's air time is busy, so its rows are exact to its end)

#define TRY(x) do { int rc__ = (x); if (rc__ != BTGPU_OK) { int c__ = rc__; std::string m__ = h->err; \
        if (getenv("BTGPU_VERBOSE")) fprintf(stderr, "btgpu_create: %s\n", m__.c_str()); return fail(c__); } } while (0)
    TRY(h->alloc(h->d_in, (h->in_cap + h->margin + 64) * sizeof(float2)));
    TRY(h->upload(h->d_taps_ch, des.channel.taps.data(), des.channel.taps.size() * sizeof(float)));
    TRY(h->upload(h->d_taps_n, des.noise.taps.data(), des.noise.taps.size() * sizeof(float)));
    TRY(h->upload(h->d_rot_ch, des.channel.rot.data(), des.channel.rot.size() * sizeof(float)));
    TRY(h->upload(h->d_rot_n, des.noise.rot.data(), des.noise.rot.size() * sizeof(float)));
    {
        std::vector<double> st(nch), sn(nch);
        for (int c = 0; c < nch; c++) {
            st[c] = -des.channel.foff[c] * d.decimation / cfg->sample_rate;
            sn[c] = -des.noise.foff[c] * d.decimation / cfg->sample_rate;
        }
        TRY(h->upload(h->d_rotstep_ch, st.data(), st.size() * sizeof(double)));
        TRY(h->upload(h->d_rotstep_n, sn.data(), sn.size() * sizeof(double)));
    }
    if (h->keep_Y) TRY(h->alloc(h->d_Y, (size_t)nch * h->ystride * sizeof(float2)));
    if (!h->use_staged) TRY(h->alloc(h->d_Yn, (size_t)nch * h->ystride_n * sizeof(float2)));
    if (h->use_pfb && h->pfb_small) {
        const PfbBank &b = h->fp.channel;
        h->ntiles_max = (int)((G + pfbm_tile(b.M) - 1) / pfbm_tile(b.M));
        TRY(h->upload(h->d_pfb_taps_ch, b.taps.data(), b.taps.size() * sizeof(float)));
        TRY(h->upload(h->d_dftw_ch, b.dftw.data(), b.dftw.size() * sizeof(float)));
        TRY(h->upload(h->d_rho_ch, b.rho.data(), b.rho.size() * sizeof(float)));
        TRY(h->upload(h->d_krot_ch, b.krot.data(), b.krot.size() * sizeof(float)));
    } else if (h->use_pfb) {
        const PfbBank &b = h->fp.channel;
        h->ntiles_max = (int)((G + 24) / 25);
        { const std::vector<float> tp = pack_branch_major(b); TRY(h->upload(h->d_pfb_taps_ch, tp.data(), tp.size() * sizeof(float))); }
        TRY(h->upload(h->d_pfb_tw, b.twiddle.data(), b.twiddle.size() * sizeof(float)));
        TRY(h->upload(h->d_binpos_ch, b.binpos.data(), b.binpos.size() * sizeof(int)));
        TRY(h->upload(h->d_binnat_ch, b.binnat.data(), b.binnat.size() * sizeof(int)));
        TRY(h->upload(h->d_rho_ch, b.rho.data(), b.rho.size() * sizeof(float)));
        {
            const std::vector<uint16_t> mf = make_dft_pass2_map(kBankNT + 5, kBankThreads, 2);
            const std::vector<uint16_t> mc = make_dft_pass2_map(kBankNT, kBankThreads, 2);
            const std::vector<uint16_t> mw = make_dft_pass2_map(kBankNT + 5, kBankThreadsWide, 1);
            const std::vector<uint16_t> m5 = make_dft_pass2_map(kBankNT + 5, kBankThreadsF, 1);
            if (mf.empty() || mc.empty() || mw.empty() || m5.empty()) return fail(BTGPU_EUNSUPPORTED);
            TRY(h->upload(h->d_b2map_f320, m5.data(), m5.size() * sizeof(uint16_t)));
            TRY(h->upload(h->d_b2map_fused_wide, mw.data(), mw.size() * sizeof(uint16_t)));
            TRY(h->upload(h->d_b2map_fused, mf.data(), mf.size() * sizeof(uint16_t)));
            TRY(h->upload(h->d_b2map_ch, mc.data(), mc.size() * sizeof(uint16_t)));
        }
        TRY(h->upload(h->d_krot_ch, b.krot.data(), b.krot.size() * sizeof(float)));
    }
    if (h->use_staged) {
        const NoiseStage &ns = h->fp.noise;
        const long long Tn = (long long)ns.outs * (S - 1) + ns.nw + ns.L3 - 1;
        h->zstride = (Tn + 10 + 63) / 64 * 64;
        if (h->noise_pfb) {
            { const std::vector<float> tp = pack_branch_major(ns.pfb); TRY(h->upload(h->d_pfb_taps_n, tp.data(), tp.size() * sizeof(float))); }
            if (!h->d_pfb_tw.p) TRY(h->upload(h->d_pfb_tw, ns.pfb.twiddle.data(), ns.pfb.twiddle.size() * sizeof(float)));
            TRY(h->upload(h->d_binpos_n, ns.pfb.binpos.data(), ns.pfb.binpos.size() * sizeof(int)));
            {
                const std::vector<uint16_t> mn = make_dft_pass2_map(kNoiseNT, kBankThreads, 2);
                if (mn.empty()) return fail(BTGPU_EUNSUPPORTED);
                TRY(h->upload(h->d_b2map_noise, mn.data(), mn.size() * sizeof(uint16_t)));
            }
            TRY(h->upload(h->d_krot_n, ns.pfb.krot.data(), ns.pfb.krot.size() * sizeof(float)));
        } else if (h->noise_small) {
            TRY(h->upload(h->d_pfb_taps_n, ns.pfb.taps.data(), ns.pfb.taps.size() * sizeof(float)));
            TRY(h->upload(h->d_dftw_n, ns.pfb.dftw.data(), ns.pfb.dftw.size() * sizeof(float)));
            TRY(h->upload(h->d_krot_n, ns.pfb.krot.data(), ns.pfb.krot.size() * sizeof(float)));
        } else {
            TRY(h->upload(h->d_taps_s1, ns.direct.taps.data(), ns.direct.taps.size() * sizeof(float)));
            TRY(h->upload(h->d_rot_s1, ns.direct.rot.data(), ns.direct.rot.size() * sizeof(float)));
            std::vector<double> st(nch);
            for (int c = 0; c < nch; c++) st[c] = -ns.direct.foff[c] * ns.R / cfg->sample_rate;
            TRY(h->upload(h->d_rotstep_s1, st.data(), st.size() * sizeof(double)));
        }
        TRY(h->upload(h->d_h3, ns.h3.data(), ns.h3.size() * sizeof(float)));
        TRY(h->upload(h->d_w, ns.weights.data(), ns.weights.size() * sizeof(double)));
    }
    h->drow = (h->use_pfb && !h->pfb_small) ? 80 : win_drow(nch);   // the 100-bin bank's epilogue writes 80-float rows
    if (nch > 80) return fail(BTGPU_EUNSUPPORTED);
    TRY(h->alloc(h->d_P, (size_t)nch * h->nb_max * sizeof(double)));
    TRY(h->alloc(h->d_Pt, (size_t)nch * h->nb_max * sizeof(double)));
    TRY(h->alloc(h->d_Q, (size_t)nch * S * sizeof(double)));
    TRY(h->upload(h->d_mmse, des.mmse, sizeof des.mmse));
    TRY(h->upload(h->d_atan, des.atan_tab, sizeof des.atan_tab));
    if (getenv("BTGPU_PFB_PROF")) {                       // per-phase cycle sums of the bank kernel (diagnostics)
        TRY(h->alloc(h->d_prof, (size_t)(h->ntiles_max + 64) * 8 * sizeof(unsigned long long)));
        if (hipMemset(h->d_prof.p, 0, (size_t)(h->ntiles_max + 64) * 8 * sizeof(unsigned long long)) != hipSuccess) return fail(BTGPU_EDEVICE);
    }
    TRY(h->upload(h->d_aclo, des.ac.byte_lo, sizeof des.ac.byte_lo));
    TRY(h->upload(h->d_achi, des.ac.byte_hi, sizeof des.ac.byte_hi));
    TRY(h->upload(h->d_pcol, des.ac.btbb_pcol, sizeof des.ac.btbb_pcol));
    TRY(h->upload(h->d_wh18, des.wh.first18, sizeof des.wh.first18));
    TRY(h->upload(h->d_le_hdr, des.le.hdr, sizeof des.le.hdr));
    TRY(h->upload(h->d_le_whiten, des.le.whiten16, sizeof des.le.whiten16));
    TRY(h->upload(h->d_le_index, des.le.index_of_channel, sizeof des.le.index_of_channel));
    TRY(h->alloc(h->d_winbits, (size_t)((S + 2) / 3) * kBitWords * kWinThreads * sizeof(uint32_t)));   // most workgroups: 3 slots each
    if (h->verify) {
        h->vcap = verify_capacity(S, nch);
        std::vector<float> ta(exact_taps_floats(nch, d.decimation));
        exact_pack_taps(des.channel.taps.data(), nch, des.channel.ntp, d.decimation, ta.data());
        TRY(h->upload(h->d_tapsA, ta.data(), ta.size() * sizeof(float)));
        h->ex_kern = exact_rows_pick(d.decimation); h->ex_lds = exact_lds_bytes(d.decimation);
        h->bm_tiles = exact_ntiles(h->ystride + 64);
        (void)hipFuncSetAttribute((const void *)h->ex_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(h->ex_lds, 64 * 1024));
    }
    for (int i = 0; i < h->nctx; i++) {
        auto &t = h->tc[i];
        if (h->use_pfb) {
            TRY(h->alloc(t.d_ptile, (size_t)nch * h->ntiles_max * sizeof(double)));
            TRY(h->alloc(t.d_phead, (size_t)nch * h->ntiles_max * sizeof(double)));
            if (h->verify && h->pfb_small && verify_has_fine(des, h->fp, win_drow(nch)))
                TRY(h->alloc(t.d_pfine, (size_t)nch * h->ntiles_max * (pfbm_tile(h->fp.channel.M) / 25) * sizeof(double)));
        }
        if (h->use_staged) TRY(h->alloc(t.d_Z, (size_t)nch * h->zstride * sizeof(float2)));
        TRY(h->alloc(t.d_winlen, (size_t)S * nch * sizeof(int)));
        TRY(h->alloc(t.d_hits, (size_t)h->max_hits * sizeof(DeviceHit)));
        TRY(h->alloc(t.d_hitcount, 2 * sizeof(unsigned int)));
        TRY(h->alloc(t.d_fin, (size_t)S * nch * sizeof(FinishRec)));
        TRY(h->alloc(t.d_eon, (size_t)S * nch * sizeof(double)));
        TRY(h->alloc(t.d_eoff, (size_t)S * nch * sizeof(double)));
        TRY(h->alloc(t.d_snr, (size_t)S * nch * sizeof(double)));
        if (h->verify) {
            const int vcap = h->vcap, nps = (vcap + nch - 1) / nch;
            TRY(h->alloc(t.d_vtasks, (size_t)vcap * sizeof(VerifyTask)));
            TRY(h->alloc(t.d_vcount, kVerCountWords * sizeof(unsigned int)));
            TRY(h->alloc(t.d_bm, (size_t)2 * h->bm_tiles * kExBmWords * sizeof(uint32_t)));
            TRY(h->alloc(t.d_chanfloor, 81 * sizeof(float)));
            TRY(h->alloc(t.d_dxt, ((size_t)nps * kVerRows + 64) * h->drow * sizeof(float)));
            TRY(h->alloc(t.d_winbits_v, (size_t)(nps + 1) * kBitWords * kWinThreads * sizeof(uint32_t)));
        }
        TRY(h->alloc(t.d_d, (size_t)h->drow * (h->ystride + 64) * sizeof(float)));   // [G][drow], time-major
        if (h->use_dcol) TRY(h->alloc(t.d_dcol, (size_t)((h->ystride + 64) / (kBankNT - 1) + 2) * 80 * (kBankNT - 1) * sizeof(float)));
        if (h->want_hdrs) TRY(h->alloc(t.d_hdr, (size_t)h->max_hits * sizeof(HeaderRec)));
        if (h->want_syms) {
            const size_t maxfin = (size_t)S * nch;            // one FinishRec per hit window, whatever max_hits is
            TRY(h->alloc(t.d_winfin, (size_t)S * nch * sizeof(int)));
            TRY(h->alloc(t.d_symbits, std::max<size_t>(maxfin, btgpu_handle::kEagerFin) * kSymWords * sizeof(uint32_t)));
        }
    }
    for (int i = 0; i < h->nctx; i++) {
        auto &t = h->tc[i];
        // (one record per hit window / per hit: as many as a batch can hold, within 64 Ki -- 31 MB of symbols and 9 MB of sweeps per context)
        h->eager_fin = (unsigned)std::min<size_t>(std::max<size_t>((size_t)S * nch, 1024), 65536);
        h->eager_hdr = (unsigned)std::min<size_t>(std::max<size_t>((size_t)h->max_hits, 1024), 65536);
        h->eager_hits = (unsigned)std::min<size_t>(std::max<size_t>((size_t)h->max_hits, 1024), btgpu_handle::kEagerHits);
        if (const char *e = getenv("BTGPU_EAGER_CAP")) {      // tests: a small capacity, so that a small capture takes the spill path
            const unsigned v = (unsigned)std::max(1, atoi(e));
            h->eager_fin = std::min(h->eager_fin, v); h->eager_hdr = std::min(h->eager_hdr, v); h->eager_hits = std::min(h->eager_hits, v);
        }
        if (h->want_hdrs && hipHostMalloc((void **)&t.h_hdr, (size_t)h->eager_hdr * sizeof(HeaderRec), hipHostMallocDefault) != hipSuccess) return fail(BTGPU_ENOMEM);
        if (h->want_syms && hipHostMalloc((void **)&t.h_sym, (size_t)h->eager_fin * kSymWords * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return fail(BTGPU_ENOMEM);
        if (hipHostMalloc((void **)&t.h_count, 12 * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) return fail(BTGPU_ENOMEM);
        std::memset(t.h_count, 0, 12 * sizeof(unsigned int));
        if (hipHostMalloc((void **)&t.h_hits, (size_t)h->eager_hits * sizeof(DeviceHit), hipHostMallocDefault) != hipSuccess) return fail(BTGPU_ENOMEM);
    }
#undef TRY
    // allow > 48 KiB of dynamic LDS for the FIR tiles
    (void)hipFuncSetAttribute((const void *)ddc_direct_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, true, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, false, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<15, 5, 10, false, false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, true, true, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, true, 8, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<true, true, 20, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, false, 8, 15>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, true, 8, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, true, 8, 7, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<true, true, 8, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfbm_kernel<false, false, 8, 15, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, true, true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, false, true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100_kernel<7, 1, 26, false, true, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreadsF, true, kBankKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreadsF, false, kBankKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, kBankKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, false, kBankKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, kBankKT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, kBankKT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, kBankKT, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, true, 2 * kBankKT, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void *)pfb100f_kernel<kBankThreads, false, 2 * kBankKT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    h->pre.assign((size_t)h->margin * 2, 0.f);
    if (getenv("BTGPU_VERBOSE"))
        fprintf(stderr, "btgpu_create: %d contexts, d=%p Z=%p ptile=%p\n", h->nctx, h->tc[0].d_d.p, h->tc[0].d_Z.p, h->tc[0].d_ptile.p);

    h->carry.assign((size_t)(d.history - 1) * 2, 0.f);   // GNU Radio pre-fills history()-1 zeros [EXT]
    *out = h;
    return BTGPU_OK;
}

void btgpu_destroy(btgpu_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->release();
    delete h;
}

int btgpu_get_design(const btgpu_handle *h, btgpu_design *out)
{
    if (!h || !out) return BTGPU_EINVAL;
    *out = h->des.d;
    return BTGPU_OK;
}

int btgpu_history(const btgpu_handle *h) { return h ? h->des.d.history : BTGPU_EINVAL; }
int btgpu_device(const btgpu_handle *h) { return h ? h->device : BTGPU_EINVAL; }
int btgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BTGPU_ENODEVICE;
    return n;
}
const char *btgpu_last_error(const btgpu_handle *h) { return h ? h->err.c_str() : "null handle"; }

int btgpu_process_device(btgpu_handle *h, const void *d_iq, size_t n_complex, size_t left_margin,
                         uint64_t first_slot, uint64_t n_slots, void *hip_stream)
{
    if (!h || !d_iq) return BTGPU_EINVAL;
    const btgpu_design &d = h->des.d;
    if (n_slots == 0) return BTGPU_OK;
    const size_t need = left_margin + (size_t)d.history + (size_t)(n_slots - 1) * d.samples_per_slot;
    if (n_complex < need) return BTGPU_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    int rc_all = BTGPU_OK;
    for (uint64_t s0 = 0; s0 < n_slots; s0 += (uint64_t)h->max_slots) {
        const int S = (int)std::min<uint64_t>((uint64_t)h->max_slots, n_slots - s0);
        const long long w0 = (long long)left_margin + (long long)s0 * d.samples_per_slot;
        int rc = h->process_batch((const float2 *)d_iq, n_complex, w0, first_slot + s0, S, st);
        if (rc == BTGPU_EOVERFLOW) rc_all = rc;
        else if (rc != BTGPU_OK) return rc;
    }
    return rc_all;
}

int btgpu_process_host(btgpu_handle *h, const float *iq, size_t n_complex, size_t left_margin, uint64_t first_slot,
                       uint64_t n_slots)
{
    if (!h || !iq) return BTGPU_EINVAL;
    const btgpu_design &d = h->des.d;
    if (n_slots == 0) return BTGPU_OK;
    const size_t H = (size_t)d.history, slot = (size_t)d.samples_per_slot, mg = (size_t)h->margin;
    if (n_complex < left_margin + H + (size_t)(n_slots - 1) * slot) return BTGPU_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    int rc_all = BTGPU_OK;
    for (uint64_t s0 = 0; s0 < n_slots; s0 += (uint64_t)h->max_slots) {
        const int S = (int)std::min<uint64_t>((uint64_t)h->max_slots, n_slots - s0);
        const size_t seg = H + (size_t)(S - 1) * slot;
        const size_t i0 = left_margin + (size_t)s0 * slot;           // sample index of window 0 of this batch
        const size_t have = std::min(mg, i0), zeros = mg - have;     // real samples in front, else implied zeros
        int rc = h->stage_batch(nullptr, zeros, iq + 2 * (i0 - have), have + seg, (long long)mg, first_slot + s0, S);
        if (rc == BTGPU_EOVERFLOW) rc_all = rc;
        else if (rc != BTGPU_OK) return rc;
    }
    return rc_all;
}

int btgpu_work(btgpu_handle *h, const float *items, size_t n_items, size_t *consumed)
{
    if (!h || !items) return BTGPU_EINVAL;
    const btgpu_design &d = h->des.d;
    if (consumed) *consumed = 0;
    const size_t H = (size_t)d.history, slot = (size_t)d.samples_per_slot, mg = (size_t)h->margin;
    if (n_items < H - 1 + slot) return BTGPU_OK;                 // not a whole slot of new items yet
    const uint64_t n_slots = (n_items - (H - 1)) / slot;
    HIPCHK(h, hipSetDevice(h->device));
    int rc_all = BTGPU_OK;
    for (uint64_t s0 = 0; s0 < n_slots; s0 += (uint64_t)h->max_slots) {
        const int S = (int)std::min<uint64_t>((uint64_t)h->max_slots, n_slots - s0);
        const size_t seg = H + (size_t)(S - 1) * slot;
        const size_t i0 = (size_t)s0 * slot;                     // item index of window 0 of this batch
        // staging buffer = [margin samples preceding items[i0] | items[i0 .. i0+seg)]
        size_t from_items = std::min(mg, i0), from_pre = mg - from_items;
        // pre holds the mg samples before items[0]; its last from_pre samples come first
        int rc = h->stage_batch(from_pre ? h->pre.data() + 2 * (mg - from_pre) : nullptr, from_pre,
                                items + 2 * (i0 - from_items), from_items + seg, (long long)mg, h->push_slot + s0, S);
        if (rc == BTGPU_EOVERFLOW) rc_all = rc;
        else if (rc != BTGPU_OK) return rc;
    }
    h->push_slot += n_slots;
    const size_t used = (size_t)n_slots * slot;
    if (mg) {                                                    // margin for the next call: samples before items[used]
        std::vector<float> np(2 * mg);
        for (size_t i = 0; i < mg; i++) {
            long long src = (long long)used - (long long)mg + (long long)i;      // item index, may be < 0 -> old pre
            const float *pp = src >= 0 ? items + 2 * src : h->pre.data() + 2 * (mg + src);
            np[2 * i] = pp[0]; np[2 * i + 1] = pp[1];
        }
        h->pre.swap(np);
    }
    if (consumed) *consumed = used;
    return rc_all;
}

int btgpu_push(btgpu_handle *h, const float *iq, size_t n_complex)
{
    if (!h || (!iq && n_complex)) return BTGPU_EINVAL;
    h->carry.insert(h->carry.end(), iq, iq + 2 * n_complex);
    size_t consumed = 0;
    int rc = btgpu_work(h, h->carry.data(), h->carry.size() / 2, &consumed);
    if (rc != BTGPU_OK && rc != BTGPU_EOVERFLOW) return rc;
    if (consumed) h->carry.erase(h->carry.begin(), h->carry.begin() + 2 * consumed);
    return rc;
}

int btgpu_pending(const btgpu_handle *h)
{
    if (!h) return BTGPU_EINVAL;
    (void)const_cast<btgpu_handle *>(h)->harvest_all(false);      // take batches whose tail has finished
    return (int)h->pending_records();
}

int btgpu_flush(btgpu_handle *h)
{
    if (!h) return BTGPU_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    const int rc = h->harvest_all(true);
    // Every stream of the handle is drained one by one here.  Functionally the harvest above already implies
    // it (the tail depends on the front); measured reason: with the streams left "complete but never waited
    // on", the caller's next device-wide synchronise (torch.cuda.synchronize / hipDeviceSynchronize) took
    // 23-32 ms in one run out of four on an idle device -- after these per-stream waits it takes 20 us
    // (0 long ones in 24 runs, one of 6 ms in the 12 before).
    for (hipStream_t st : {h->stream, h->post_stream, h->tail_stream, h->copy_stream, h->spill_stream, h->tail_extra[0], h->tail_extra[1], h->sq_stream}) if (st) (void)hipStreamSynchronize(st);
    return rc;
}

int btgpu_poll(btgpu_handle *h, btgpu_hit *out, int max_hits)
{
    if (!h || (!out && max_hits > 0) || max_hits < 0) return BTGPU_EINVAL;
    (void)h->harvest_all(false);
    int n = (int)std::min<size_t>((size_t)max_hits, h->pending_records());
    if (n > 0) {
        std::memcpy(out, h->queue.data() + h->qhead, sizeof(btgpu_hit) * n);
        h->pop_records((size_t)n);
    }
    return n;
}

static int poll_symbols_impl(btgpu_handle *h, btgpu_hit *out, btgpu_header *hdr, uint8_t *symbols, int sym_cap, int *sym_len,
                             int max_hits)
{
    if (!h || (!out && max_hits > 0) || max_hits < 0 || sym_cap < 0 || (!symbols && sym_cap > 0)) return BTGPU_EINVAL;
    if (!h->want_syms || (hdr && !h->want_hdrs)) return BTGPU_EUNSUPPORTED;
    (void)h->harvest_all(false);
    int n = (int)std::min<size_t>((size_t)max_hits, h->pending_records());
    for (int i = 0; i < n; i++) {
        const size_t qi = h->qhead + (size_t)i;
        out[i] = h->queue[qi];
        if (hdr) hdr[i] = h->qhdr[qi];
        const uint32_t *bits = &h->qbits[qi * kSymWords];
        // what the reference hands to ac()/aa(): &symp[i], len - i  (one symbol per byte, air order)
        int avail = h->qhas[qi] ? out[i].nsym : 0;
        if (avail > sym_cap) avail = sym_cap;
        const int first = out[i].offset;
        const int limit = kSymWords * 32 - first;
        if (avail > limit) avail = limit < 0 ? 0 : limit;
        uint8_t *dst = symbols + (size_t)i * sym_cap;
        for (int s = 0; s < avail; s++) { const int b = first + s; dst[s] = (uint8_t)((bits[b >> 5] >> (b & 31)) & 1u); }
        if (sym_len) sym_len[i] = avail;
    }
    if (n > 0) h->pop_records((size_t)n);
    return n;
}

int btgpu_poll_symbols(btgpu_handle *h, btgpu_hit *out, uint8_t *symbols, int sym_cap, int *sym_len, int max_hits)
{
    return poll_symbols_impl(h, out, nullptr, symbols, sym_cap, sym_len, max_hits);
}

int btgpu_poll_headers(btgpu_handle *h, btgpu_hit *out, btgpu_header *hdr, uint8_t *symbols, int sym_cap, int *sym_len,
                       int max_hits)
{
    if (!hdr && max_hits > 0) return BTGPU_EINVAL;
    return poll_symbols_impl(h, out, hdr, symbols, sym_cap, sym_len, max_hits);
}

int btgpu_last_timing(const btgpu_handle *h, btgpu_timing *out)
{
    if (!h || !out) return BTGPU_EINVAL;
    *out = h->timing;
    return BTGPU_OK;
}

long btgpu_debug_fetch(btgpu_handle *h, int what, int channel, size_t first, size_t count, void *out)
{
    if (!h || !out) return BTGPU_EINVAL;
    (void)h->harvest_all(true);
    const btgpu_design &d = h->des.d;
    const int nch = d.high_channel - d.low_channel + 1;
    const int c = channel - d.low_channel;
    const void *src = nullptr;
    size_t elem = 0, avail = 0;
    switch (what) {
        case 0:
            if (c < 0 || c >= nch || !h->keep_Y) return BTGPU_EINVAL;
            src = (const float2 *)h->d_Y.p + (size_t)c * h->ystride; elem = sizeof(float2); avail = (size_t)h->last_G; break;
        case 1: {
            if (c < 0 || c >= nch) return BTGPU_EINVAL;
            avail = (size_t)h->last_G;
            if (first >= avail) return 0;
            count = std::min(count, avail - first);
            if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
            if (hipMemcpy2D(out, sizeof(float), (const float *)h->tc[h->last_ctx].d_d.p + first * h->drow + c, (size_t)h->drow * sizeof(float),
                            sizeof(float), count, hipMemcpyDeviceToHost) != hipSuccess) return BTGPU_EDEVICE;
            return (long)count;
        }
        case 7: src = h->tc[h->last_ctx].d_winlen.p; elem = sizeof(int); avail = (size_t)h->last_S * nch; break;
        case 8: src = h->tc[h->last_ctx].d_fin.p; elem = sizeof(FinishRec); avail = (size_t)h->last_S * nch; break;
        case 9: if (!h->d_prof.p) return BTGPU_EINVAL; src = h->d_prof.p; elem = sizeof(unsigned long long); avail = (size_t)(h->ntiles_max + 64) * 8; break;
        case 10:                                             // exact stage: the tasks of the last batch (VerifyTask: w, n_exact, snr)
            if (!h->verify) return BTGPU_EINVAL;
            src = h->tc[h->last_ctx].d_vtasks.p; elem = sizeof(VerifyTask); avail = std::min<size_t>(h->tc[h->last_ctx].h_count[4], (size_t)h->vcap); break;
        case 11:                                             // exact rows: the two bitmaps of the last batch, [2][bm_tiles][kExBmWords] uint32
            if (!h->verify) return BTGPU_EINVAL;
            src = h->tc[h->last_ctx].d_bm.p; elem = sizeof(uint32_t); avail = (size_t)2 * h->bm_tiles * kExBmWords; break;
        case 2: src = h->tc[h->last_ctx].d_eon.p; elem = sizeof(double); avail = (size_t)h->last_S * nch; break;
        case 3: src = h->tc[h->last_ctx].d_eoff.p; elem = sizeof(double); avail = (size_t)h->last_S * nch; break;
        case 4: src = h->tc[h->last_ctx].d_snr.p; elem = sizeof(double); avail = (size_t)h->last_S * nch; break;
        case 5:
            if (c < 0 || c >= nch || h->use_staged) return BTGPU_EINVAL;
            src = (const float2 *)h->d_Yn.p + (size_t)c * h->ystride_n; elem = sizeof(float2);
            avail = (size_t)h->last_S * h->des.outs_per_slot; break;
        default: return BTGPU_EINVAL;
    }
    if (first >= avail) return 0;
    count = std::min(count, avail - first);
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    if (hipMemcpy(out, (const char *)src + first * elem, count * elem, hipMemcpyDeviceToHost) != hipSuccess)
        return BTGPU_EDEVICE;
    return (long)count;
}

// ---------------------------------------------------------------------------------------
// The correlator on a captured symbol stream (parity entry: samples/channel37.dem of the reference)
// ---------------------------------------------------------------------------------------
long btgpu_debug_scan_symbols(const uint8_t *symbols, size_t n, int device, int policy, btgpu_hit *out, long cap)
{
    if (!symbols || cap < 0 || (!out && cap > 0) || (policy != 0 && policy != 1) || n > 0x7fffffffull) return BTGPU_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BTGPU_ENODEVICE;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev || hipSetDevice(device) != hipSuccess) return BTGPU_EINVAL;
    // access-code tables: the same host design the blocks use (they do not depend on the rate)
    Design *des = new (std::nothrow) Design();
    if (!des) return BTGPU_ENOMEM;
    btgpu_config cfg{};
    cfg.sample_rate = 8e6; cfg.center_freq = 2476.5e6; cfg.squelch_db = 10.0; cfg.mode = BTGPU_MODE_SNIFFER;
    int rc = make_design(cfg, *des);
    if (rc != BTGPU_OK) { delete des; return rc; }
    const size_t nwords = (n + 31) / 32;
    std::vector<uint32_t> words(nwords + 1, 0u);
    for (size_t i = 0; i < n; i++) if (symbols[i] & 1) words[i >> 5] |= 1u << (i & 31);
    const unsigned long long chunks = (n + 624) / 625;
    const long max_hits = (long)std::min<unsigned long long>(chunks * 64 + 1024, 1ull << 24);
    uint32_t *d_words = nullptr; uint64_t *d_lo = nullptr; uint32_t *d_hi = nullptr; DeviceHit *d_hits = nullptr; unsigned int *d_count = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_words); (void)hipFree(d_lo); (void)hipFree(d_hi); (void)hipFree(d_hits); (void)hipFree(d_count); delete des; };
    if (hipMalloc((void **)&d_words, words.size() * 4) != hipSuccess || hipMalloc((void **)&d_lo, sizeof des->ac.byte_lo) != hipSuccess ||
        hipMalloc((void **)&d_hi, sizeof des->ac.byte_hi) != hipSuccess || hipMalloc((void **)&d_hits, (size_t)max_hits * sizeof(DeviceHit)) != hipSuccess ||
        hipMalloc((void **)&d_count, sizeof(unsigned int)) != hipSuccess) { cleanup(); return BTGPU_ENOMEM; }
    long result = BTGPU_EDEVICE;
    do {
        if (hipMemcpy(d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_lo, des->ac.byte_lo, sizeof des->ac.byte_lo, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_hi, des->ac.byte_hi, sizeof des->ac.byte_hi, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemset(d_count, 0, sizeof(unsigned int)) != hipSuccess) break;
        const unsigned nblk = (unsigned)((chunks + kWinThreads - 1) / kWinThreads);
        if (nblk)
            hipLaunchKernelGGL(scan_symbols_kernel, dim3(nblk), dim3(kWinThreads), 0, 0, (const uint32_t *)d_words,
                               (unsigned long long)n, 1, des->ac.a0_lo, des->ac.a0_hi, (const uint64_t *)d_lo,
                               (const uint32_t *)d_hi, d_hits, d_count, (int)max_hits);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        unsigned int count = 0;
        if (hipMemcpy(&count, d_count, sizeof count, hipMemcpyDeviceToHost) != hipSuccess) break;
        if ((long)count > max_hits) { result = BTGPU_EOVERFLOW; break; }
        std::vector<DeviceHit> hh(count);
        if (count && hipMemcpy(hh.data(), d_hits, sizeof(DeviceHit) * count, hipMemcpyDeviceToHost) != hipSuccess) break;
        std::vector<btgpu_hit> all(count);
        for (unsigned int i = 0; i < count; i++) {
            btgpu_hit o{};
            o.slot = hh[i].slot;                                     // 625-offset chunk of the stream
            o.offset = (int32_t)((long long)hh[i].slot * 625 + hh[i].offset);   // absolute symbol offset
            o.lap = hh[i].lap; o.ac_errors = hh[i].ac_errors; o.kind = BTGPU_KIND_AC;
            o.nsym = (int32_t)((long long)n - o.offset);
            all[i] = o;
        }
        std::sort(all.begin(), all.end(), [](const btgpu_hit &a, const btgpu_hit &b) { return a.offset < b.offset; });
        long m = 0, total = 0;
        long long next = 0;                                          // policy 1: a hit moves the scan on by 68 symbols
        for (const btgpu_hit &o : all) {
            if (policy == 1) { if (o.offset < next) continue; next = (long long)o.offset + kSymbolsShortAC; }
            if (m < cap) out[m++] = o;
            total++;
        }
        result = total;
    } while (0);
    cleanup();
    return result;
}

// ---------------------------------------------------------------------------------------
// hop reversal (basic_rate_piconet::init_hop_reversal / winnow, lib/piconet_impl.cc:96-338)
// ---------------------------------------------------------------------------------------
struct btgpu_hopseq {
    int device = 0;
    uint8_t *d_sequence = nullptr;
    uint32_t *d_cand[2] = {nullptr, nullptr};
    unsigned int *d_count = nullptr;
    int cur = 0;
    unsigned int ncand = 0;
    size_t cand_cap = 0;
};

int btgpu_hopseq_create(uint32_t address, int afh, int device, btgpu_hopseq **out)
{
    if (!out) return BTGPU_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BTGPU_ENODEVICE;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev || hipSetDevice(device) != hipSuccess) return BTGPU_EINVAL;
    btgpu_hopseq *h = new (std::nothrow) btgpu_hopseq();
    if (!h) return BTGPU_ENOMEM;
    h->device = device;
    h->cand_cap = (size_t)kSequenceLength / 64 + 64;           // every probe could match
    if (hipMalloc((void **)&h->d_sequence, kSequenceLength) != hipSuccess ||
        hipMalloc((void **)&h->d_cand[0], h->cand_cap * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&h->d_cand[1], h->cand_cap * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&h->d_count, sizeof(unsigned int)) != hipSuccess) { btgpu_hopseq_destroy(h); return BTGPU_ENOMEM; }
    address &= 0xfffffff;
    HopAddress ad;                                             // address_precalc (lib/piconet_impl.cc:149-167)
    ad.a1 = (address >> 23) & 0x1f;
    ad.b = (address >> 19) & 0x0f;
    ad.c1 = ((address >> 4) & 0x10) + ((address >> 3) & 0x08) + ((address >> 2) & 0x04) + ((address >> 1) & 0x02) + (address & 0x01);
    ad.d1 = (address >> 10) & 0x1ff;
    ad.e = ((address >> 7) & 0x40) + ((address >> 6) & 0x20) + ((address >> 5) & 0x10) + ((address >> 4) & 0x08) +
           ((address >> 3) & 0x04) + ((address >> 2) & 0x02) + ((address >> 1) & 0x01);
    ad.afh = afh ? 1 : 0;
    hipLaunchKernelGGL(gen_hops_kernel, dim3(kSequenceLength / 2 / 256), dim3(256), 0, 0, ad, h->d_sequence);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { btgpu_hopseq_destroy(h); return BTGPU_EDEVICE; }
    *out = h;
    return BTGPU_OK;
}

void btgpu_hopseq_destroy(btgpu_hopseq *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_sequence) (void)hipFree(h->d_sequence);
    if (h->d_cand[0]) (void)hipFree(h->d_cand[0]);
    if (h->d_cand[1]) (void)hipFree(h->d_cand[1]);
    if (h->d_count) (void)hipFree(h->d_count);
    delete h;
}

int btgpu_hopseq_init_candidates(btgpu_hopseq *h, int channel, int known_clock_bits, int aliased)
{
    if (!h || known_clock_bits < 0 || known_clock_bits > 63) return BTGPU_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    if (hipMemset(h->d_count, 0, sizeof(unsigned int)) != hipSuccess) return BTGPU_EDEVICE;
    h->cur = 0;
    hipLaunchKernelGGL(init_candidates_kernel, dim3(kSequenceLength / 64 / 256), dim3(256), 0, 0, (const uint8_t *)h->d_sequence,
                       channel, known_clock_bits, aliased ? 1 : 0, h->d_cand[0], h->d_count);
    if (hipMemcpy(&h->ncand, h->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) return BTGPU_EDEVICE;
    return (int)h->ncand;
}

int btgpu_hopseq_winnow(btgpu_hopseq *h, int offset, int channel, int aliased)
{
    if (!h) return BTGPU_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    if (h->ncand == 0) return 0;
    if (hipMemset(h->d_count, 0, sizeof(unsigned int)) != hipSuccess) return BTGPU_EDEVICE;
    hipLaunchKernelGGL(winnow_kernel, dim3((h->ncand + 255) / 256), dim3(256), 0, 0, (const uint8_t *)h->d_sequence,
                       (const uint32_t *)h->d_cand[h->cur], h->ncand, (uint32_t)offset, channel, aliased ? 1 : 0,
                       h->d_cand[h->cur ^ 1], h->d_count);
    if (hipMemcpy(&h->ncand, h->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) return BTGPU_EDEVICE;
    h->cur ^= 1;
    return (int)h->ncand;
}

int btgpu_hopseq_candidates(btgpu_hopseq *h, uint32_t *out, int cap)
{
    if (!h || cap < 0 || (!out && cap > 0)) return BTGPU_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    std::vector<uint32_t> all(h->ncand);
    if (h->ncand && hipMemcpy(all.data(), h->d_cand[h->cur], sizeof(uint32_t) * h->ncand, hipMemcpyDeviceToHost) != hipSuccess)
        return BTGPU_EDEVICE;
    std::sort(all.begin(), all.end());                         // the reference's list is ascending
    const int n = std::min<int>(cap, (int)all.size());
    if (n > 0) std::memcpy(out, all.data(), sizeof(uint32_t) * (size_t)n);
    return (int)h->ncand;
}

int btgpu_hopseq_lookup(btgpu_hopseq *h, const uint32_t *index, int n, uint8_t *channel)
{
    if (!h || n < 0 || (n > 0 && (!index || !channel))) return BTGPU_EINVAL;
    if (n == 0) return BTGPU_OK;
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    uint32_t *d_i = nullptr; uint8_t *d_o = nullptr;
    if (hipMalloc((void **)&d_i, sizeof(uint32_t) * (size_t)n) != hipSuccess) return BTGPU_ENOMEM;
    if (hipMalloc((void **)&d_o, (size_t)n) != hipSuccess) { (void)hipFree(d_i); return BTGPU_ENOMEM; }
    int rc = BTGPU_OK;
    if (hipMemcpy(d_i, index, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = BTGPU_EDEVICE;
    if (rc == BTGPU_OK) {
        hipLaunchKernelGGL(hop_lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const uint8_t *)h->d_sequence,
                           (const uint32_t *)d_i, n, d_o);
        if (hipMemcpy(channel, d_o, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = BTGPU_EDEVICE;
    }
    (void)hipFree(d_i); (void)hipFree(d_o);
    return rc;
}

long btgpu_hopseq_fetch(btgpu_hopseq *h, size_t first, size_t count, uint8_t *out)
{
    if (!h || !out) return BTGPU_EINVAL;
    if (first >= kSequenceLength) return 0;
    count = std::min<size_t>(count, (size_t)kSequenceLength - first);
    if (hipSetDevice(h->device) != hipSuccess) return BTGPU_EDEVICE;
    if (hipMemcpy(out, h->d_sequence + first, count, hipMemcpyDeviceToHost) != hipSuccess) return BTGPU_EDEVICE;
    return (long)count;
}

}  // extern "C"
