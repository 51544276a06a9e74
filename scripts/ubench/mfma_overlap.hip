// mfma_overlap.hip -- do an MFMA-chain wave and a load/LDS/VALU wave on the SAME SIMD overlap?  (exact.hip.h's matrix / epilogue waves)
// Per CU one workgroup: waves 0-3 run dependent v_mfma_f32_32x32x2_f32 chains, waves 4-7 a loop of global loads, LDS traffic and VALU work.
// Three launches: matrix waves only, other waves only, both.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, const float4 *src, int iters, int roles, int kind)
{
    __shared__ float4 buf[2048];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        if (!(roles & 1)) return;
        f32x16 acc;
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
        float a = 1.0f + lane * 1e-9f, b = 1.0f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 50; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        for (int i = 0; i < 16; i++) s += acc[i];
    } else {
        if (!(roles & 2)) return;
        const int t = threadIdx.x - 256;
        for (int it = 0; it < iters; it++) {
            if (kind & 1) {                                          // 13 KB from the L2 into LDS
                float4 v0 = src[(size_t)(it & 63) * 832 + t], v1 = src[(size_t)(it & 63) * 832 + 256 + t], v2 = src[(size_t)(it & 63) * 832 + 512 + t];
                buf[t] = v0; buf[256 + t] = v1; buf[512 + t] = v2;
            }
            if (kind & 2) {                                          // LDS reads + dependent adds (the diagonal sums)
                float y = 0.f;
                const float *g = (const float *)buf;
#pragma unroll
                for (int q = 0; q < 28; q++) y = y + g[q * 136 + lane + q];
                s += y;
            }
            if (kind & 4) {                                          // ~150 VALU instructions (the demodulator)
                float z = s + 1.0f;
#pragma unroll
                for (int q = 0; q < 150; q++) z = fmaf(z, 1.0000001f, 1e-7f);
                s += z;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *d; float4 *src; hipMalloc(&d, 512 * 4096 * 4); hipMalloc(&src, 64 * 832 * 16); hipMemset(src, 0, 64 * 832 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int wgs = 1; wgs <= 2; wgs++)
        for (int kind = 1; kind <= 7; kind = kind == 1 ? 2 : kind == 2 ? 4 : kind == 4 ? 7 : 8) {
            float ms[4] = {0, 0, 0, 0};
            for (int roles = 1; roles <= 3; roles++) {
                hipLaunchKernelGGL(k, dim3(256 * wgs), dim3(512), 0, 0, d, src, 4, roles, kind);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256 * wgs), dim3(512), 0, 0, d, src, iters, roles, kind);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[roles], e0, e1);
            }
            printf("%d workgroup(s) per CU, other work kind %d (1 L2->LDS copy, 2 LDS sums, 4 VALU): matrix alone %.3f ms, other alone %.3f ms, both %.3f ms (max %.3f, sum %.3f)\n",
                   wgs, kind, ms[1], ms[2], ms[3], ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2]);
        }
    return 0;
}
