// hopseq.hip.h -- hop reversal on the GPU: the complete basic-rate
// hopping sequence of one piconet (2^27 entries, one per 625 us slot, 128 MiB in HBM) and the
// CLK1-27 candidate lists that basic_rate_piconet winnows with it.
//   gen_hops_kernel          lib/piconet_impl.cc:214-255 (gen_hops) with perm5 (:179-211) as bit
//                            operations instead of the reference's 512 KiB lookup table
//   init_candidates_kernel   lib/piconet_impl.cc:285-302
//   winnow_kernel            lib/piconet_impl.cc:305-321
// All three are embarrassingly parallel: one lane per pair of sequence entries / per probe / per
// candidate; the table is written once with coalesced 2-byte stores and probed at random afterwards.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace btgpu {

constexpr uint32_t kSequenceLength = 134217728u;      // include/gr_bluetooth/piconet.h:83

struct HopAddress {                                   // address_precalc (lib/piconet_impl.cc:149-167)
    int a1, b, c1, d1, e, afh;
};

// 5-bit permutation: 14 butterflies, control bit i swaps z bits (i1[i], i2[i]); applied from
// control bit 13 down to 0 (perm5, :179-211)
__device__ __forceinline__ int perm5_bits(int z, int p_high, int p_low)
{
    const uint32_t p = (uint32_t)p_low | ((uint32_t)p_high << 9);
    // index pairs packed 3 bits each, stage 13 first
    const int i1[14] = {0, 2, 1, 3, 0, 1, 0, 3, 1, 0, 2, 1, 0, 1};
    const int i2[14] = {1, 3, 2, 4, 4, 3, 2, 4, 4, 3, 4, 3, 3, 2};
#pragma unroll
    for (int i = 13; i >= 0; i--) {
        const int x = ((z >> i1[i]) ^ (z >> i2[i])) & 1;      // bits differ?
        const int sw = x & (int)((p >> i) & 1u);
        z ^= (sw << i1[i]) | (sw << i2[i]);
    }
    return z;
}

__device__ __forceinline__ int hop_bank(int k) { return (k * 2) % 79; }        // precalc (:131-146)

// one lane = sequence entries 2n and 2n + 1 (clock bit 1 = 0 / 1 share x, a, c, d, f)
__global__ __launch_bounds__(256) void gen_hops_kernel(HopAddress ad, uint8_t *__restrict__ sequence)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;           // 0 .. 2^26 - 1
    if (n >= kSequenceLength / 2) return;
    const uint32_t index = 2 * n;
    const int x = (int)(n & 31u);
    const int k9 = (int)((index >> 6) & 0x1ffu);
    const int j = (int)((index >> 15) & 31u);
    const int i = (int)((index >> 20) & 31u);
    const int a = ad.a1 ^ i, c = ad.c1 ^ j, d = ad.d1 ^ k9;
    const int f = (int)(16u * (index >> 6));
    const int perm_in = ((x + a) & 31) ^ ad.b;
    const int h0 = hop_bank((perm5_bits(perm_in, c, d) + ad.e + f) % 79);
    int h1 = h0;
    if (!ad.afh) h1 = hop_bank((perm5_bits(perm_in, c ^ 0x1f, d) + ad.e + f + 32) % 79);
    ((uchar2 *)sequence)[n] = make_uchar2((unsigned char)h0, (unsigned char)h1);
}

__device__ __forceinline__ int aliased_channel(int ch) { return ((ch + 24) % 25) + 26; }   // :520-523

// probes i = known_clock_bits + 64 p; matches are appended (order restored on the host when asked for)
__global__ __launch_bounds__(256) void init_candidates_kernel(const uint8_t *__restrict__ sequence, int channel,
                                                             int known_clock_bits, int aliased,
                                                             uint32_t *__restrict__ cand, unsigned int *__restrict__ count)
{
    const uint32_t pidx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)known_clock_bits + 64u * pidx;
    if (i >= kSequenceLength) return;
    int obs = sequence[i];
    if (aliased) obs = aliased_channel(obs);
    if (obs == channel) cand[atomicAdd(count, 1u)] = i;
}

__global__ __launch_bounds__(256) void winnow_kernel(const uint8_t *__restrict__ sequence, const uint32_t *__restrict__ in,
                                                    unsigned int n_in, uint32_t offset, int channel, int aliased,
                                                    uint32_t *__restrict__ out, unsigned int *__restrict__ count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_in) return;
    const uint32_t cnd = in[t];
    int obs = sequence[(cnd + offset) % kSequenceLength];
    if (aliased) obs = aliased_channel(obs);
    if (obs == channel) out[atomicAdd(count, 1u)] = cnd;
}

__global__ void hop_lookup_kernel(const uint8_t *__restrict__ sequence, const uint32_t *__restrict__ index, int n,
                                  uint8_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = sequence[index[t] % kSequenceLength];
}

}  // namespace btgpu
