/*
 * btgpu.h -- C ABI of the MI355X-native gr-bluetooth multi-channel sniffer hot path.
 *
 * This is the drop-in boundary: a GNU Radio block (gr::bluetooth::multi_LAP /
 * gr::bluetooth::multi_sniffer, see gr-bluetooth_amd/host/ and INTEGRATION.md) keeps
 * the reference's make()/work() surface and forwards to these entry points.  Plain
 * pointers and sizes only; no C++ or torch types; no exceptions cross this boundary
 * (every function returns BTGPU_OK or a negative BTGPU_E* code).
 *
 * Reference interfaces replaced (paths relative to the gr-bluetooth checkout):
 *   btgpu_design_query   multi_block::multi_block + set_symbol_history + set_channels
 *                        (lib/multi_block.cc:40-120, :299-342): pure host arithmetic
 *   btgpu_create         multi_LAP::make / multi_sniffer::make
 *                        (lib/multi_LAP_impl.cc:39-56, lib/multi_sniffer_impl.cc:42-72)
 *   btgpu_work           multi_LAP_impl::work / multi_sniffer_impl::work
 *                        (lib/multi_LAP_impl.cc:65-114, lib/multi_sniffer_impl.cc:82-166):
 *                        same input contract (history()-1 old items followed by the new
 *                        ones), but consumes every whole slot in the buffer at once
 *   btgpu_process_device same work on a device-resident stream segment (time-partitioned
 *                        multi-GPU and bench entry; no reference counterpart)
 *   btgpu_process_host   ... on a host-resident segment (what btrx_amd --gpus N gives each device)
 *   btgpu_poll           replaces the printf side effect of work(): hit records in the
 *                        order the reference's loops print them (slot, channel, offset)
 *   btgpu_acgen          classic_packet::acgen (lib/packet_impl.cc:309-364)
 */
#ifndef BTGPU_H
#define BTGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTGPU_OK            0
#define BTGPU_EINVAL       -1   /* bad argument / unsupported configuration */
#define BTGPU_ENOMEM       -2
#define BTGPU_EDEVICE      -3   /* HIP runtime error (see btgpu_last_error)  */
#define BTGPU_ENODEVICE    -4   /* no gfx950 device visible                  */
#define BTGPU_EOVERFLOW    -5   /* hit buffer overflow (hits were dropped)   */
#define BTGPU_EUNSUPPORTED -6

#define BTGPU_MODE_LAP      0   /* gr::bluetooth::multi_LAP     (+68 symbols of history)   */
#define BTGPU_MODE_SNIFFER  1   /* gr::bluetooth::multi_sniffer (+3125 symbols of history) */

#define BTGPU_CHANNELIZER_AUTO      0
#define BTGPU_CHANNELIZER_DIRECT    1   /* per-channel direct-form DDC; bit-exact vs the oracle */
#define BTGPU_CHANNELIZER_POLYPHASE 2   /* polyphase filter bank; equal within float tolerance   */

#define BTGPU_SQUELCH_AUTO      0
#define BTGPU_SQUELCH_DIRECT    1       /* exact direct-form 20001-tap-class noise DDC           */
#define BTGPU_SQUELCH_STAGED    2       /* two-stage equivalent filter + quadrature (tolerance)  */

#define BTGPU_CORRELATOR_AUTO   0       /* multi_LAP: BTBB (what the reference links), multi_sniffer: INTREE */
#define BTGPU_CORRELATOR_INTREE 1       /* classic_packet::sniff_ac / check_ac (lib/packet_impl.cc:247-268,471-510) */
#define BTGPU_CORRELATOR_BTBB   2       /* libbtbb btbb_find_ac with btbb_init(1), max_ac_errors 1, LAP_ANY
                                           (lib/multi_LAP_impl.cc:55,93) [EXT libbtbb, unpinned]; multi_LAP
                                           only; hit.offset is then the sync-word start and hit.ac_errors the
                                           number of corrected bits (0 or 1)                            */

#define BTGPU_FLAG_LE        0x1        /* also run the le_packet::sniff_aa pass (sniffer mode)  */
#define BTGPU_FLAG_DEBUG_Y   0x2        /* keep the channel-bank output Y for btgpu_debug_fetch  */
#define BTGPU_FLAG_SYMBOLS   0x8        /* keep the sliced symbols of every window that reported a
                                           hit; fetch them with btgpu_poll_symbols                */
#define BTGPU_FLAG_HEADERS   0x10       /* (implies SYMBOLS) sweep the packet header of every classic hit
                                           over the 64 CLK1-6 candidates on the GPU; btgpu_poll_headers   */
#define BTGPU_FLAG_ASYNC     0x4        /* work/process_device return once a batch is enqueued;
                                           its records appear in a later btgpu_poll (always in
                                           stream order) or after btgpu_flush.  The tail of batch
                                           n (finish_kernel, record copy) overlaps batch n+1.    */

#define BTGPU_FLAG_TIMING    0x20       /* bracket the kernels of every batch with HIP events for btgpu_last_timing
                                           (off by default: eleven event records per batch are not free)     */
#define BTGPU_FLAG_TIMING_BANK 0x80      /* the light form: events only around the channel-bank kernel (BTGPU_K_DDC_CHANNEL) and the exact
                                           rows' kernel (BTGPU_K_EXACT); what bench.py's timed region uses for its roofline figures */
#define BTGPU_FLAG_NO_NSYM   0x40       /* LAP-list consumers (multi_LAP prints the LAP only, lib/multi_LAP_impl.cc:93-110):
                                           skip the clock-recovery continuation over the rest of a hit window that
                                           only produces hit.nsym; nsym is then -1 unless the window ended inside
                                           the 693-symbol detection span.  Ignored with BTGPU_FLAG_SYMBOLS / HEADERS */

#define BTGPU_FLAG_NO_VERIFY 0x100      /* polyphase channelizer only: do NOT recompute the demodulated rows of the busy windows (energy
                                           anywhere in the detection span: "presence") through the bit-exact direct-form arithmetic.  By
                                           default they are (exact rows, DESIGN.md sections 4.4 and 5): the records then equal the CPU
                                           reference's in slot, channel, kind, offset, LAP and ac_errors, and the symbols handed to the
                                           host layer are the reference's to the end of each packet; with this flag ~1e-4 .. 2e-2 of the
                                           records (by traffic) come out a symbol apart or on one side only (the round-3 behaviour) */

#define BTGPU_FLAG_EXACT_PAYLOAD 0x200   /* accepted; ALWAYS ON since round 6.  The symbols a record hands to the host layer are the reference's
                                           own arithmetic to the END OF THE PACKET: a packet's air time is busy, so its rows are exact rows
                                           anyway (no separate pass).  What every payload decode and CRC of the host layer sees
                                           (lib/packet_impl.cc:1066-1160) equals the CPU reference's */

#define BTGPU_FLAG_EXACT_ALL 0x400       /* polyphase channelizer: NO SELECTION AT ALL -- every row of every channel is recomputed by the reference's
                                           direct-form arithmetic on the matrix pipe (presence marks everything): every field of every record,
                                           nsym and the records born from noise included, equals the CPU reference's with the bit-exact
                                           channelizer's (the squelch figure stays the staged filter's: SNR within 3.4e-6 dB).  ~8 Gsamples/s
                                           at 79 channels / 100 Msps instead of ~30 */

#define BTGPU_KIND_AC 0
#define BTGPU_KIND_AA 1

typedef struct btgpu_handle btgpu_handle;

typedef struct btgpu_config {
    double  sample_rate;      /* multi_*::make(sample_rate, ...)            */
    double  center_freq;      /* multi_*::make(..., center_freq, ...)       */
    double  squelch_db;       /* multi_*::make(..., squelch_threshold)      */
    int32_t mode;             /* BTGPU_MODE_*                               */
    int32_t device;           /* HIP device ordinal, -1 = current device    */
    int32_t channelizer;      /* BTGPU_CHANNELIZER_*                        */
    int32_t squelch;          /* BTGPU_SQUELCH_*                            */
    int32_t flags;            /* BTGPU_FLAG_*                               */
    int32_t max_batch_slots;  /* slots per internal batch, 0 = default      */
    int32_t max_hits;         /* hit-buffer capacity per batch, 0 = default */
    int32_t correlator;       /* BTGPU_CORRELATOR_*                          */
} btgpu_config;

/* Everything the multi_block constructor derives (lib/multi_block.cc:56-119). */
typedef struct btgpu_design {
    double  samples_per_symbol;
    int32_t samples_per_slot;
    int32_t decimation;
    int32_t ntaps_channel, ntaps_noise;
    int32_t low_channel, high_channel;       /* classic channel numbers, inclusive */
    int32_t first_channel_sample, first_noise_sample;
    int32_t history;                          /* history() incl. symbol history     */
    int32_t ddc_out;                          /* channel DDC outputs per window     */
    int32_t noise_out;                        /* noise DDC outputs per window       */
    int32_t channelizer;                      /* resolved BTGPU_CHANNELIZER_*       */
    int32_t squelch;                          /* resolved BTGPU_SQUELCH_*           */
    int32_t left_margin;                      /* samples before a device segment the staged
                                                 squelch may read (0 for DIRECT)     */
    int32_t correlator;                       /* resolved BTGPU_CORRELATOR_*        */
} btgpu_design;

typedef struct btgpu_hit {
    uint64_t slot;        /* work() call index == "time slot" the reference prints          */
    int32_t  channel;     /* classic channel 0..78                                          */
    int32_t  offset;      /* symbol offset in the window's symbol array (sniff_ac result)    */
    uint32_t lap;         /* LAP (kind AC) / access address (kind AA)                       */
    int32_t  ac_errors;   /* mismatches over the 68 checked access-code bits               */
    int32_t  kind;        /* BTGPU_KIND_*                                                   */
    int32_t  nsym;        /* symbols available to the packet handler from `offset` on       */
    double   snr_db;      /* 10 log10(E_on / E_off) of the (slot, channel) window           */
} btgpu_hit;

/* Per-kernel GPU time (needs BTGPU_FLAG_TIMING; zeros otherwise), cumulative since btgpu_create (callers take
 * differences), measured with HIP events recorded on the streams the kernels are launched on; a batch is
 * accounted when its records are harvested.  With BTGPU_FLAG_ASYNC up to three batches are in flight (banks of
 * batch n+2, post stage of n+1, tail of n on three streams): a kernel's time then includes what it loses to
 * the kernels running beside it. */
#define BTGPU_K_DDC_CHANNEL   0   /* channel bank (direct DDC or polyphase channelizer) */
#define BTGPU_K_DEMOD_ENERGY  1   /* direct form: quadrature demod + |Y|^2 block sums; polyphase: tile sums -> block sums */
#define BTGPU_K_DDC_NOISE     2   /* noise bank                                          */
#define BTGPU_K_NOISE_ENERGY  3   /* noise |Y|^2 per-slot sums                           */
#define BTGPU_K_WINDOW        4   /* squelch + M&M + slicer + access-code search         */
#define BTGPU_K_FINISH        5   /* M&M continuation of the windows that reported hits  */
#define BTGPU_K_VERIFY        6   /* the second run (tail stream): exact rows under the first run's uncovered hits, fill, window kernel over those windows */
#define BTGPU_K_EXACT         7   /* exact_rows_kernel over presence's marks, in line before the window kernel (bracketed with BTGPU_FLAG_TIMING_BANK too) */
#define BTGPU_K_COUNT         8
typedef struct btgpu_timing {
    float    kernel_ms[BTGPU_K_COUNT];        /* summed over launches                  */
    uint32_t kernel_launches[BTGPU_K_COUNT];
    float    total_ms;                        /* first kernel start to last kernel end */
    uint32_t batches;
    uint64_t samples;                         /* new complex samples consumed          */
    uint64_t slots;
    uint64_t verify_windows;                  /* busy windows (presence) + windows of the second run (always counted)    */
    uint64_t verify_rows;                     /* demodulated rows recomputed by exact_rows_kernel (both launches), in tiles of 1250 / 11 rows */
    uint64_t verify_turned_away;              /* windows that should have gone to the second run but found its list full */
    uint64_t long_tasks;                      /* windows of the second run (a classic hit on rows presence had not covered) */
    uint64_t long_rows;                       /* ... the rows recomputed for them                                         */
    uint64_t long_turned_away;                /* (unused since round 6)                                                   */
} btgpu_timing;

/* ---- host-only helpers (no GPU needed) ---- */
int  btgpu_design_query(const btgpu_config *cfg, btgpu_design *out);
int  btgpu_acgen(uint32_t lap, uint8_t ac[9]);
int  btgpu_filter_taps(const btgpu_config *cfg, int which /*0 channel, 1 noise*/, float *taps, int cap);
const char *btgpu_strerror(int code);
const char *btgpu_version(void);

/* ---- block lifetime ---- */
int  btgpu_create(const btgpu_config *cfg, btgpu_handle **out);
void btgpu_destroy(btgpu_handle *h);          /* frees the handle's memory; its eight HIP streams are kept by the library and handed
                                                 to the next btgpu_create on the same device (streams created after others were
                                                 destroyed measured 25 % slower steps; BTGPU_STREAM_POOL=0 in the environment: off) */
int  btgpu_get_design(const btgpu_handle *h, btgpu_design *out);
int  btgpu_history(const btgpu_handle *h);
int  btgpu_device(const btgpu_handle *h);     /* HIP ordinal the handle lives on (what device = -1 resolved to) */
int  btgpu_device_count(void);                /* visible HIP devices (whatever their architecture: btgpu_create checks
                                                 for gfx950), or BTGPU_ENODEVICE                                  */
const char *btgpu_last_error(const btgpu_handle *h);

/* ---- work() ----
 * `items` is what GNU Radio hands to work(): interleaved float32 I/Q, history()-1 old
 * samples followed by the new ones, `n_items` complex samples in total (so GNU Radio's
 * noutput_items = n_items - (history()-1)).  Processes S = noutput_items / samples_per_slot
 * whole slots -- the reference processes exactly one per call and returns samples_per_slot
 * (lib/multi_sniffer_impl.cc:165); S consecutive reference calls give the same records --
 * queues their hits, stores S*samples_per_slot in *consumed.  Returns BTGPU_OK or <0. */
int btgpu_work(btgpu_handle *h, const float *items, size_t n_items, size_t *consumed);

/* Streaming convenience: keeps history internally (zero pre-filled like the GNU Radio
 * scheduler) so arbitrary chunks can be pushed. */
int btgpu_push(btgpu_handle *h, const float *iq, size_t n_complex);

/* Device-resident segment: d_iq[left_margin] is absolute sample
 * first_slot*samples_per_slot-(history()-1); from there the segment must hold
 * history() + (n_slots-1)*samples_per_slot complex samples.  The `left_margin` samples in
 * front (real stream data when the segment is cut out of a longer stream; 0 at the stream
 * start, where GNU Radio's pre-filled zeros are implied) let the staged squelch filter see
 * the same samples the reference's noise filter would; btgpu_design.left_margin is enough.
 * `hip_stream` is a hipStream_t (NULL = the handle's own stream).  The segment must stay valid until the
 * batch's records have been handed out (btgpu_flush, or a btgpu_poll that returns them): the exact
 * confirmation of the polyphase path re-reads the samples of the windows it takes on the handle's tail stream. */
int btgpu_process_device(btgpu_handle *h, const void *d_iq, size_t n_complex, size_t left_margin,
                         uint64_t first_slot, uint64_t n_slots, void *hip_stream);

/* The same for a segment in HOST memory (one rank of a time-partitioned run driven from C/C++): the library
 * stages it batch by batch through two pinned buffers, the copy of one batch overlapping the kernels of the
 * previous one (with BTGPU_FLAG_ASYNC).  Same layout and meaning of the arguments as btgpu_process_device. */
int btgpu_process_host(btgpu_handle *h, const float *iq, size_t n_complex, size_t left_margin,
                       uint64_t first_slot, uint64_t n_slots);

/* Drain queued hits, ordered by (slot, channel, kind, offset). Returns count (>=0) or <0. */
int btgpu_poll(btgpu_handle *h, btgpu_hit *out, int max_hits);
/* Like btgpu_poll, plus what the reference hands to its packet handlers with every hit
 * (lib/multi_sniffer_impl.cc:116: ac(&symp[i], len - i, ...)): the window's sliced symbols from the
 * hit offset on, one symbol per byte (air order), up to sym_cap per hit, written to
 * symbols[i*sym_cap ...]; sym_len[i] = number written (min(nsym, sym_cap)).  Needs
 * BTGPU_FLAG_SYMBOLS; note the LE quirk Q6: for kind AA nsym counts from the reduced length. */
int btgpu_poll_symbols(btgpu_handle *h, btgpu_hit *out, uint8_t *symbols, int sym_cap, int *sym_len, int max_hits);
/* What classic_packet::try_clock(clock) yields for clock = 0..63 (lib/packet_impl.cc:1046-1063): the UAP
 * that satisfies the HEC and the packet type, from the FEC-1/3-decoded, unwhitened header; fec13_ok = 0
 * where unfec13 (:367-383) reports failure (try_clock then returns 0 and leaves type/UAP alone). */
typedef struct btgpu_header {
    uint8_t uap[64];
    uint8_t type[64];
    int32_t fec13_ok;
    int32_t reserved;
} btgpu_header;
/* btgpu_poll_symbols plus the header sweep of each record (zeroed for kind AA); needs BTGPU_FLAG_HEADERS */
int btgpu_poll_headers(btgpu_handle *h, btgpu_hit *out, btgpu_header *hdr, uint8_t *symbols, int sym_cap,
                       int *sym_len, int max_hits);
int btgpu_pending(const btgpu_handle *h);
/* Wait for every enqueued batch and move its records to the poll queue (BTGPU_FLAG_ASYNC). */
int btgpu_flush(btgpu_handle *h);

int btgpu_last_timing(const btgpu_handle *h, btgpu_timing *out);

/* ---- introspection for parity tests: copies an intermediate of the LAST batch ----
 * what: 0 channel-bank output Y (complex64 [channel][g]; DIRECT channelizer or BTGPU_FLAG_DEBUG_Y),
 *       1 demodulated stream d (float32, time-major [g][nch]: pass channel to get a strided copy),
 *       2 E_on per window (float64, [slot][channel]), 3 E_off per window, 4 snr per window,
 *       5 noise-bank output (complex64; DIRECT squelch only),
 *       10 the exact stage's tasks ({int32 window = slot * nch + channel index, int32 rows, float64 snr}), 11 their exact
 *       demodulated rows (float32 [task][1416]; rows 1 .. rows-1 of a task are valid).
 * channel is a classic channel number (ignored for 2..4); returns elements copied. */
long btgpu_debug_fetch(btgpu_handle *h, int what, int channel, size_t first, size_t count, void *out);

/* Host-side copies of the integer tables the kernels use (header-byte distance tables of
 * le_packet::sniff_aa under the reference's names, "derived/classic_first18", "derived/le_whiten16"):
 * returns bytes written or <0.  For the digest test against the reference's literals. */
int btgpu_debug_lut(const char *name, void *out, int cap_bytes);

/* ---- parity entry: the access-code search on a captured SYMBOL stream ----
 * Runs the window kernel's search (classic_packet::sniff_ac + check_ac, lib/packet_impl.cc:247-268,
 * :471-510) straight on `n` sliced symbols, one per byte (0/1) -- the format of the reference's
 * samples/channel37.dem -- without the float front end.  policy 0: every qualifying offset;
 * policy 1: a stream scan that moves on by 68 symbols after a hit (the sniffer's loop,
 * lib/multi_sniffer_impl.cc:107-127, applied to the whole stream).  Records: offset = absolute
 * symbol offset, lap, ac_errors, nsym = n - offset; ordered by offset.  Returns the number of
 * records (the first `cap` are written) or a negative error.  No handle needed. */
long btgpu_debug_scan_symbols(const uint8_t *symbols, size_t n, int device, int policy, btgpu_hit *out, long cap);

/* ---- hop reversal: the piconet's complete hopping sequence on the GPU ----
 * Replaces basic_rate_piconet_impl::init_hop_reversal's precalc + address_precalc + gen_hops
 * (lib/piconet_impl.cc:96-255: a 128 MiB table, one entry per 625 us slot, index = CLK27..1),
 * init_candidates (:285-302), winnow(offset, channel) (:305-321) and hop(clock) (:279-282).
 * address = (UAP << 24 | LAP) & 0xfffffff; afh = the piconet's d_afh flag. */
typedef struct btgpu_hopseq btgpu_hopseq;
int  btgpu_hopseq_create(uint32_t address, int afh, int device, btgpu_hopseq **out);
void btgpu_hopseq_destroy(btgpu_hopseq *h);
/* candidate CLK1-27 values whose hop matches `channel` among those with the known CLK1-6 bits;
 * returns the number of candidates (or a negative error) */
int  btgpu_hopseq_init_candidates(btgpu_hopseq *h, int channel, int known_clock_bits, int aliased);
/* keep the candidates whose hop `offset` slots later is `channel`; returns how many remain */
int  btgpu_hopseq_winnow(btgpu_hopseq *h, int offset, int channel, int aliased);
/* the surviving candidates in ascending order (up to cap); returns their total number */
int  btgpu_hopseq_candidates(btgpu_hopseq *h, uint32_t *out, int cap);
int  btgpu_hopseq_lookup(btgpu_hopseq *h, const uint32_t *index, int n, uint8_t *channel);
long btgpu_hopseq_fetch(btgpu_hopseq *h, size_t first, size_t count, uint8_t *out);   /* table slice */

#ifdef __cplusplus
}
#endif
#endif /* BTGPU_H */
