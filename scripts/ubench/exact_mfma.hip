// exact_mfma.hip -- micro-benchmark and numerics proof for gr-bluetooth_amd/csrc/exact.hip.h (run on an MI355X):
//   1. v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain, bit for bit, on N random operand blocks (wide exponent range, signed
//      zeros, subnormals) -- the premise of the summation order DESIGN.md fixes;
//   2. exact_rows_kernel<50>'s de-rotated outputs == a plain C restatement of that order (the oracle's ddc_run) on random input;
//   3. its rate: rows/s, TFLOP/s of useful multiply-adds, for a given number of busy channels per tile.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../gr-bluetooth_amd/csrc -I../../include -o /tmp/exact_mfma exact_mfma.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "exact.hip.h"

using namespace btgpu;

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(2); } } while (0)

__host__ __device__ inline uint32_t mix(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// a float with a random sign, 23 random mantissa bits and an exponent spread over +-20 around 1; 1 in 64 a signed zero, 1 in 64 a subnormal
__host__ __device__ inline float rnd_float(uint32_t h)
{
    const uint32_t sel = h & 63u, sign = (h >> 6) & 1u, man = (h >> 9) & 0x7fffffu;
    uint32_t e = 127u - 20u + ((h >> 7) % 41u);
    if (sel == 0) return sign ? -0.0f : 0.0f;
    uint32_t bits = (sign << 31) | (sel == 1 ? 0u : e << 23) | man;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(bits);
#else
    memcpy(&f, &bits, 4);
#endif
    return f;
}

// one wave = one 32 x 32 block with K = 2 * steps; A[i][k] = rnd(blk, i, k), B[k][j] = rnd(blk, 1000 + j, k)
__global__ void mfma_chain_check(int steps, uint32_t seed, unsigned long long *mismatch, float *first_bad)
{
    const int lane = threadIdx.x & 63;
    const uint32_t blk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + seed * 0x10001u;
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int s = 0; s < steps; s++) {
        const int k = 2 * s + (lane >> 5);
        const float a = rnd_float(mix(blk, lane & 31, k)), b = rnd_float(mix(blk, 1000 + (lane & 31), k));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const int col = lane & 31;
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        float ref = 0.f;
        for (int k = 0; k < 2 * steps; k++) ref = fmaf(rnd_float(mix(blk, row, k)), rnd_float(mix(blk, 1000 + col, k)), ref);
        if (__float_as_uint(ref) != __float_as_uint(acc[i])) {
            if (atomicAdd(mismatch, 1ull) == 0) { first_bad[0] = ref; first_bad[1] = acc[i]; first_bad[2] = (float)row; first_bad[3] = (float)col; }
        }
    }
}

int main(int argc, char **argv)
{
    const int per_tile = argc > 1 ? atoi(argv[1]) : 4;           // busy channels per tile in the timing run
    const int S = argc > 2 ? atoi(argv[2]) : 2304;               // slots of the timing run
    // ---- 1. MFMA == fmaf chain ----
    {
        unsigned long long *d_mis; float *d_bad;
        CK(hipMalloc(&d_mis, 8)); CK(hipMalloc(&d_bad, 16)); CK(hipMemset(d_mis, 0, 8));
        const int steps = 50, blocks = 2560, waves = 4;          // 2560 * 4 blocks * 32 * 32 * 100 products
        unsigned long long total = 0;
        for (uint32_t seed = 1; seed <= 10; seed++) {
            hipLaunchKernelGGL(mfma_chain_check, dim3(blocks), dim3(64 * waves), 0, 0, steps, seed, d_mis, d_bad);
            total += (unsigned long long)blocks * waves * 32 * 32 * 2 * steps;
        }
        CK(hipDeviceSynchronize());
        unsigned long long mis; float bad[4];
        CK(hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
        printf("mfma_f32_32x32x2f32 vs fmaf chain (K = 100): %llu products, %llu outputs differ", total, mis);
        if (mis) printf(" (first: chain %a mfma %a at row %g col %g)", bad[0], bad[1], bad[2], bad[3]);
        printf("\n");
        if (mis) return 1;
    }
    // ---- 2. exact_rows_kernel<50> vs the C restatement ----
    constexpr int D = 50;
    const int nch = 79, ntaps = 667, ntp = 672, Qr = 2;
    std::vector<float> taps((size_t)nch * ntp * 2, 0.f);
    for (int c = 0; c < nch; c++) for (int j = 0; j < ntaps; j++) {
        taps[((size_t)c * ntp + j) * 2] = 1e-3f * (float)((int)(mix(7, c, j) % 2001u) - 1000);
        taps[((size_t)c * ntp + j) * 2 + 1] = 1e-3f * (float)((int)(mix(8, c, j) % 2001u) - 1000);
    }
    std::vector<float> tapsA(exact_taps_floats(nch, D));
    exact_pack_taps(taps.data(), nch, ntp, D, tapsA.data());
    std::vector<float> rot((size_t)nch * Qr * 2);
    for (int c = 0; c < nch; c++) { rot[(c * Qr) * 2] = 1.f; rot[(c * Qr) * 2 + 1] = 0.f; rot[(c * Qr + 1) * 2] = (c & 1) ? -1.f : 1.f; rot[(c * Qr + 1) * 2 + 1] = 0.f; }
    std::vector<float> atab(257);
    for (int i = 0; i <= 255; i++) atab[i] = (float)atan((double)i / 255.0);
    atab[256] = atab[255];
    auto run = [&](long long G, const std::vector<uint32_t> &bitmap, size_t x_len, const float2 *d_x, float *d_d, float *d_dcol, float2 *d_y, int reps, float *ms) {
        float *d_tapsA, *d_atab; float2 *d_rot; uint32_t *d_bm;
        CK(hipMalloc(&d_tapsA, tapsA.size() * 4)); CK(hipMemcpy(d_tapsA, tapsA.data(), tapsA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_atab, 257 * 4)); CK(hipMemcpy(d_atab, atab.data(), 257 * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_rot, rot.size() * 4)); CK(hipMemcpy(d_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_bm, bitmap.size() * 4)); CK(hipMemcpy(d_bm, bitmap.data(), bitmap.size() * 4, hipMemcpyHostToDevice));
        ExactParams p{};
        p.x_len = (long long)x_len; p.first0 = 0; p.G = G; p.tapsA = d_tapsA; p.rot = d_rot; p.Qr = Qr; p.atan_tab = d_atab; p.gain = 1.0f;
        p.bitmap = d_bm; p.ntiles = exact_ntiles(G); p.dbg = getenv("UB_DBG") ? atoi(getenv("UB_DBG")) : 0; p.d = d_d; p.drow = 80; p.dcol = d_dcol; p.ydbg = d_y; p.ystride = G; p.nch = nch;
        const size_t lds = exact_lds_bytes(D);
        CK(hipFuncSetAttribute((const void *)exact_rows_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(exact_rows_kernel<D>, dim3(p.ntiles), dim3(kExThreads), lds, 0, p, d_x);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int grid = getenv("UB_GRID") ? atoi(getenv("UB_GRID")) : p.ntiles;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(exact_rows_kernel<D>, dim3(grid), dim3(kExThreads), lds, 0, p, d_x);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (ms) { CK(hipEventElapsedTime(ms, e0, e1)); *ms /= (float)(reps > 0 ? reps : 1); }
        CK(hipFree(d_tapsA)); CK(hipFree(d_atab)); CK(hipFree(d_rot)); CK(hipFree(d_bm));
    };
    {
        const long long G = 1000;
        const size_t x_len = (size_t)G * D + 700;
        std::vector<float> x(x_len * 2);
        for (size_t i = 0; i < x.size(); i++) x[i] = 1e-2f * (float)((int)(mix(9, (uint32_t)i, 0) % 2001u) - 1000);
        std::vector<uint32_t> bm((size_t)exact_ntiles(G) * kExWords, 0);
        const int chans[5] = {0, 1, 31, 40, 78};
        for (int t = 0; t < exact_ntiles(G); t++) for (int c : chans) bm[(size_t)t * kExWords + c / 32] |= 1u << (c % 32);
        float2 *d_x, *d_y; float *d_d;
        CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_y, (size_t)nch * G * 8)); CK(hipMemset(d_y, 0, (size_t)nch * G * 8));
        CK(hipMalloc(&d_d, (size_t)G * 80 * 4)); CK(hipMemset(d_d, 0, (size_t)G * 80 * 4));
        run(G, bm, x_len, d_x, d_d, nullptr, d_y, 0, nullptr);
        std::vector<float> y((size_t)nch * G * 2);
        CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
        long long bad = 0, checked = 0;
        for (int c : chans) for (long long g = 0; g < G; g++) {
            float yr = 0.f, yi = 0.f;
            for (int q = 0; q < kExQB; q++) {
                float gr = 0.f, gi = 0.f;
                for (int r = 0; r < D; r++) {
                    const int j = q * D + r;
                    const float tr = j < ntp ? taps[((size_t)c * ntp + j) * 2] : 0.f, ti = j < ntp ? taps[((size_t)c * ntp + j) * 2 + 1] : 0.f;
                    const size_t a = (size_t)g * D + j;
                    const float vr = a < x_len ? x[2 * a] : 0.f, vi = a < x_len ? x[2 * a + 1] : 0.f;
                    gr = fmaf(tr, vr, gr); gr = fmaf(-ti, vi, gr);
                    gi = fmaf(ti, vr, gi); gi = fmaf(tr, vi, gi);
                }
                yr = q ? yr + gr : gr; yi = q ? yi + gi : gi;
            }
            const float rr = rot[((size_t)c * Qr + g % Qr) * 2], ri = rot[((size_t)c * Qr + g % Qr) * 2 + 1];
            const float ox = fmaf(-yi, ri, yr * rr), oy = fmaf(yi, rr, yr * ri);
            checked++;
            if (memcmp(&ox, &y[((size_t)c * G + g) * 2], 4) || memcmp(&oy, &y[((size_t)c * G + g) * 2 + 1], 4)) {
                if (!bad) printf("first difference: channel %d row %lld: want (%a, %a) got (%a, %a)\n", c, g, ox, oy, y[((size_t)c * G + g) * 2], y[((size_t)c * G + g) * 2 + 1]);
                bad++;
            }
        }
        printf("exact_rows_kernel<50> vs the C restatement of its order: %lld outputs checked, %lld differ\n", checked, bad);
        CK(hipFree(d_x)); CK(hipFree(d_y)); CK(hipFree(d_d));
        if (bad && !getenv("UB_DBG")) return 1;
    }
    // ---- 3. rate ----
    {
        const long long G = (long long)S * 1250;
        const size_t x_len = (size_t)G * D + 700;
        float2 *d_x; float *d_d, *d_dcol;
        CK(hipMalloc(&d_x, x_len * 8));
        {
            std::vector<float> x(x_len * 2);
            for (size_t i = 0; i < x.size(); i++) x[i] = 1e-2f * (float)((int)(mix(9, (uint32_t)i, 0) % 2001u) - 1000);
            CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&d_d, (size_t)G * 80 * 4)); CK(hipMalloc(&d_dcol, ((size_t)G / 25 + 1) * 2000 * 4));
        const int only_mode = argc > 3 ? atoi(argv[3]) : -1;
        for (int mode = 0; mode < 3; mode++) {
            if (only_mode >= 0 && mode != only_mode) continue;
            // 0: per_tile channels in every tile; 1: the same number of (channel, tile) pairs bunched into a quarter of the tiles; 2: one channel per tile
            std::vector<uint32_t> bm((size_t)exact_ntiles(G) * kExWords, 0);
            long long pairs = 0;
            for (int t = 0; t < exact_ntiles(G); t++) {
                int n = mode == 0 ? per_tile : mode == 1 ? ((t & 3) == 0 ? 4 * per_tile : 0) : 1;
                for (int i = 0; i < n && i < nch; i++) { const int c = (int)((mix(3, t, i) % 79u)); bm[(size_t)t * kExWords + c / 32] |= 1u << (c % 32); }
                for (int w = 0; w < kExWords; w++) pairs += __builtin_popcount(bm[(size_t)t * kExWords + w]);
            }
            float ms = 0.f;
            run(G, bm, x_len, d_x, d_d, d_dcol, nullptr, 5, &ms);
            const double rows = (double)pairs * kExSlotRows / kExSlotTiles, fl = rows * 2.0 * 4.0 * 667.0;
            printf("mode %d: %lld (channel, tile) pairs, %.2f M rows: %.3f ms = %.2f G rows/s, %.1f TFLOP/s of useful multiply-adds, input %.2f TB/s\n", mode, pairs,
                   rows * 1e-6, ms, rows / ms * 1e-6, fl / ms * 1e-9, (double)exact_ntiles(G) * (mode == 1 ? 0.25 : 1.0) * kExCols * D * 8.0 / ms * 1e-9);
        }
        CK(hipFree(d_x)); CK(hipFree(d_d)); CK(hipFree(d_dcol));
    }
    // ---- 4. D = 4 (8 Msps, configs[1]): the rate where a channel's tile is four matrix instructions per wave ----
    {
        constexpr int D4 = 4;
        const int nch4 = 8, ntaps4 = 53, ntp4 = 56, S4 = argc > 4 ? atoi(argv[4]) : 16384;
        std::vector<float> taps4((size_t)nch4 * ntp4 * 2, 0.f);
        for (int c = 0; c < nch4; c++) for (int j = 0; j < ntaps4; j++) {
            taps4[((size_t)c * ntp4 + j) * 2] = 1e-3f * (float)((int)(mix(17, c, j) % 2001u) - 1000);
            taps4[((size_t)c * ntp4 + j) * 2 + 1] = 1e-3f * (float)((int)(mix(18, c, j) % 2001u) - 1000);
        }
        std::vector<float> tapsA4(exact_taps_floats(nch4, D4));
        exact_pack_taps(taps4.data(), nch4, ntp4, D4, tapsA4.data());
        const long long G = (long long)S4 * 1250;
        const size_t x_len = (size_t)G * D4 + 700;
        std::vector<float> xh(x_len * 2);
        for (size_t i = 0; i < xh.size(); i++) xh[i] = 1e-2f * (float)((int)(mix(19, (uint32_t)i, 0) % 2001u) - 1000);
        float2 *d_x; float *d_da, *d_tapsA, *d_atab; float2 *d_rot; uint32_t *d_bm;
        CK(hipMalloc(&d_x, x_len * 8)); CK(hipMemcpy(d_x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_da, (size_t)G * 8 * 4)); CK(hipMemset(d_da, 0, (size_t)G * 8 * 4));
        CK(hipMalloc(&d_tapsA, tapsA4.size() * 4)); CK(hipMemcpy(d_tapsA, tapsA4.data(), tapsA4.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_atab, 257 * 4)); CK(hipMemcpy(d_atab, atab.data(), 257 * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_rot, rot.size() * 4)); CK(hipMemcpy(d_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> bm((size_t)exact_ntiles(G) * kExWords, 0);
        long long pairs = 0;
        for (int t = 0; t < exact_ntiles(G); t++) { const uint32_t m = (mix(5, t, 0) % 100u) < 77u ? 0xffu : (mix(6, t, 0) & 0xffu); bm[(size_t)t * kExWords] = m; pairs += __builtin_popcount(m); }   // 77 % of the tiles with all eight channels (the bench capture's share), the rest at random
        CK(hipMalloc(&d_bm, bm.size() * 4)); CK(hipMemcpy(d_bm, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
        ExactParams p{};
        p.x_len = (long long)x_len; p.first0 = 0; p.G = G; p.tapsA = d_tapsA; p.rot = d_rot; p.Qr = Qr; p.atan_tab = d_atab; p.gain = 1.0f;
        p.bitmap = d_bm; p.ntiles = exact_ntiles(G); p.drow = 8; p.dcol = nullptr; p.nch = nch4; p.d = d_da;
        p.dbg = getenv("UB_DBG") ? atoi(getenv("UB_DBG")) : 0;
        const size_t lds = exact_lds_bytes(D4);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0.f;
        ExactRowsKernel k = exact_rows_pick(D4);
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3(p.ntiles), dim3(kExThreads), lds, 0, p, d_x);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, dim3(p.ntiles), dim3(kExThreads), lds, 0, p, d_x);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5.f;
        const double rows = (double)pairs * kExSlotRows / kExSlotTiles, fl = rows * 8.0 * ntaps4;
        printf("D = 4, 8 channels, %lld (channel, tile) pairs, %.1f M rows: %.3f ms = %.1f G rows/s, %.1f TFLOP/s of useful multiply-adds\n", pairs, rows * 1e-6, ms, rows / ms * 1e-6, fl / ms * 1e-9);
    }
    return 0;
}
