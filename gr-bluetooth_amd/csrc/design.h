// design.h -- host-side derivation of everything the reference's multi_block constructor
// computes (lib/multi_block.cc:40-120, :299-342) plus the constant tables the HIP kernels
// need.  Pure host C++ (no HIP), shared by the C ABI and the GNU Radio block mirror.
#pragma once
#include <cstdint>
#include <vector>

#include "btgpu.h"

namespace btgpu {

constexpr int kMmseTaps = 8;
constexpr int kMmseSteps = 128;
constexpr int kSymbolsPerSlot = 625;
constexpr int kSymbolsShortAC = 68;      // SYMBOLS_PER_BASIC_RATE_SHORTENED_ACCESS_CODE
constexpr int kSymbolsLeAA = 40;         // SYMBOLS_PER_LOW_ENERGY_PREAMBLE_AA
constexpr int kSymbolsHistorySniffer = 3125;

struct FilterBank {
    int ntaps = 0;                       // true filter length
    int blk = 1;                         // block length of the summation order: the bank's decimation (hop)
    int ntp = 0;                         // padded with exact zeros to whole blocks: ceil(ntaps / blk) * blk
    int nch = 0;
    std::vector<float> taps;             // [nch][ntp][2] (re, im), time-reversed, zero padded
    std::vector<double> foff;            // [nch] centre offset in Hz
    int rot_period = 0;                  // 0: not periodic (device computes the phase)
    std::vector<float> rot;              // [nch][rot_period][2] derotation factors per output index
};

struct AcTables {
    uint64_t a0_lo;                      // AC(LAP=0) bits 0..63 (bit i = i-th symbol on air)
    uint32_t a0_hi;                      //            bits 64..67
    uint64_t byte_lo[3][256];            // XOR of the affine columns selected by LAP byte b
    uint32_t byte_hi[3][256];
    uint64_t btbb_pcol[24];              // BCH(64,30) parity column of LAP bit k alone (libbtbb-style
                                         // single-error correction of a LAP bit), sync-word bit order
};

struct ClassicWhitening {                // classic_packet_impl::unwhiten (lib/packet_impl.cc:513-526)
    uint32_t first18[64];                // first 18 whitening bits for CLK1-6 = c (bit i = i-th), for the header
};

struct LeTables {                       // le_packet::sniff_aa (lib/packet_impl.cc:1452-1527)
    uint8_t hdr[4][256];                 // min Hamming distance to the valid header-byte sets:
                                         // 0 access LSB, 1 access MSB, 2 data LSB, 3 data MSB
    uint16_t whiten16[40];               // first 16 whitening bits per LE channel index (bit i = i-th)
    int8_t index_of_channel[79];         // classic channel number -> LE channel index, -1 = none
};

struct Design {
    btgpu_config cfg{};
    btgpu_design d{};
    double samples_per_slot_d = 0;       // d_samples_per_slot (double in the reference)
    std::vector<float> h_channel, h_noise;
    FilterBank channel, noise;
    float demod_gain = 0;
    float gain_mu = 0, mu0 = 0, omega_relative_limit = 0, omega0 = 0, gain_omega = 0, omega_mid = 0;
    float mmse[(kMmseSteps + 1) * kMmseTaps];
    float atan_tab[257];
    AcTables ac;
    LeTables le;
    ClassicWhitening wh;
    int blocks_per_window = 0;           // ddc_out / (slot/decim)
    int tail = 0;                        // ddc_out % (slot/decim)
    int outs_per_slot = 0;               // slot / decim; segmented: ddc_out (rows from one window to the next)
    bool segmented = false;              // 625 * sps is not a multiple of decim: every window on its own grid
};

// returns BTGPU_OK or a negative error code; never throws
int make_design(const btgpu_config &cfg, Design &out);

// own implementation of the Bluetooth access code (classic_packet::acgen equivalent)
uint64_t sync_word(uint32_t lap);
void access_code_68(uint32_t lap, uint64_t &lo, uint32_t &hi);   // bits 0..67, air order
void access_code_bytes(uint32_t lap, uint8_t ac[9]);             // 72 bits, MSB-first packing

int firdes_ntaps(double fs, double transition_width);
std::vector<float> firdes_low_pass_hann(double gain, double fs, double cutoff, double transition_width);

}  // namespace btgpu

// ---- polyphase (M = 100 bins of 1 MHz) fast path -----------------------------------------
namespace btgpu {

constexpr int kPfbM = 100;
constexpr int kNoiseMargin = 4096;      // samples before a segment the staged noise path may read

struct PfbBank {
    bool available = false;
    int M = 0;                          // bins: fs / 1 MHz (100: the 10 x 10 FFT kernel; 4..50: the small-M kernel)
    int D = 0;                          // hop (input samples per output instant)
    int L = 0;                          // prototype length
    int Q = 0;                          // taps per polyphase branch, ceil(L / 100)
    int S = 0;                          // 2*D / 100: branch-window slide per two output instants
    bool real_taps = false;
    std::vector<float> taps;            // [Q*M][2]  a[j] = proto[L-1-j] * exp(-j 2 pi delta j / M)
    std::vector<float> twiddle;         // M = 100: [100][2] exp(-j 2 pi m1 p2 / 100) at index m1*10 + p2
    std::vector<float> dftw;            // M < 100: [M][nch][2] exp(-j 2 pi p m_c / M), the DFT restricted to the channels' bins
    std::vector<int> binpos;            // [nch] position of the channel's bin in the in-place 10x10 FFT output
    std::vector<int> binnat;            // [nch] the bin itself, m in 0..99 (channel epilogue: Y[t][m])
    int rot_period = 0;
    std::vector<float> krot;            // [nch][rot_period][2]  C_m * exp(-j 2 pi f D t / fs)
    std::vector<float> rho;             // [nch][2]  exp(-j 2 pi f D / fs): y[t] conj(y[t-1]) = Y[t] conj(Y[t-1]) rho
    bool rho_real = false;              // every rho is +-1 (integer-MHz offsets at D = 50)
    bool natural = false;               // M = 8 = nch with channel c in bin c (bin of channel 0 folded into the taps)
};

// Lane -> task table of the second DFT pass over `rows` instants (tasks (row, m1), row-contiguous
// 80-byte reads at LDS pitch kPfbUst): each half-wave gets two tasks of every residue
// (row + m1) mod 16, placed so that both the 16-lane groups of ds_read_b128 / ds_write_b128 and the
// contiguous 16-lane groups of ds_write_b64 see sixteen different residues -> no bank conflicts.
// Entry = row << 4 | m1, 0xffff = idle lane.  `sweeps` * `lanes` entries.
std::vector<uint16_t> make_dft_pass2_map(int rows, int lanes, int sweeps);

// The 100-bin kernel's tap table: branch-major and padded so that a lane fetches the Q taps of its branch with
// 16-byte loads.  Real taps: [M][(Q+3)&~3] floats; complex: [M][(Q+1)&~1][2].  (b.taps itself is [Q*M][2],
// tap-major: what the small-M kernel stages through LDS.)
std::vector<float> pack_branch_major(const PfbBank &b);

struct NoiseStage {
    bool available = false;
    int R = 0;                          // stage-1 hop = 5 * decimation
    int L1 = 0;                         // B-spline prototype length
    int L3 = 0;                         // stage-2 taps at the stage-1 output rate
    int Jm = 0;                         // quadrature edge half-width (stage-2 outputs)
    int pad = 0;                        // composite length excess / 2
    int nw = 0;                         // quadrature weights per slot = outs + 2*Jm
    int outs = 0;                       // stage-2 outputs per slot (slot / R)
    std::vector<float> h3;              // [L3]
    std::vector<double> weights;        // [nw]
    double fit_l1_error = 0;            // ||composite - h_noise||_1 / ||h_noise||_1
    PfbBank pfb;                        // stage 1 as a polyphase bank (100 Msps)
    FilterBank direct;                  // stage 1 as a direct-form bank (any rate): B-spline prototype, hop R
};

struct FastPath {
    PfbBank channel;
    NoiseStage noise;
};

int make_fast_path(const Design &des, FastPath &fp);

// Exact rows (exact.hip.h): can the demodulated rows of the direct-form channel bank be computed ONCE on the shared output grid and
// serve every window that overlaps them?  The reference restarts its rotator per window (lib/multi_block.cc:180-205 through
// freq_xlating_fir_filter_ccf [EXT]); the grid's rotator differs from a window's own by the factor rot[window start], and the
// quadrature demodulator y[t] conj(y[t-1]) is bit for bit indifferent to a common factor of exactly +-1.  True when the bank is
// periodic, the windows share one grid, D is one the kernel is built for and every window start of every channel meets +-1.
constexpr int kExactSlotRows = 1250;    // rows per slot that exact_rows_kernel's tiling is laid out for (kernels.hip.h kExSlotRows: eleven tiles per slot)
bool exact_rows_available(const Design &des);

// direct-form bank builder shared by the reference-filter banks and the staged squelch
void build_direct_bank(FilterBank &b, const std::vector<float> &h, int low_ch, int nch, double extra_hz,
                       double center_freq, double fs, int decim);

}  // namespace btgpu
