// mfma_shadow.hip -- how many independent instructions of the SAME wave issue for free between two dependent v_mfma_f32_32x32x2_f32?
// (mfma_overlap.hip: instructions of OTHER waves of the SIMD do not -- their time adds to the matrix chain's.)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K, int KIND, int NACC = 1>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float buf[4096];
    const int lane = threadIdx.x & 63;
    buf[threadIdx.x] = 1.0f; buf[threadIdx.x + 256] = 2.0f;
    __syncthreads();
    f32x16 acc, acc2;
    for (int i = 0; i < 16; i++) { acc[i] = 0.f; acc2[i] = 0.f; }
    float a = 1.0f + lane * 1e-9f, b = 1.0f;
    float z0 = 1.0f, z1 = 2.0f, z2 = 3.0f, z3 = 4.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (NACC == 2 && (u & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; q++) {
                if (KIND == 0) {                                     // independent VALU
                    if ((q & 3) == 0) z0 = fmaf(z0, 1.0000001f, 1e-7f); else if ((q & 3) == 1) z1 = fmaf(z1, 1.0000001f, 1e-7f);
                    else if ((q & 3) == 2) z2 = fmaf(z2, 1.0000001f, 1e-7f); else z3 = fmaf(z3, 1.0000001f, 1e-7f);
                } else {                                             // LDS reads
                    z0 += buf[(lane + 64 * q + 16 * u) & 4095];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = z0 + z1 + z2 + z3;
    for (int i = 0; i < 16; i++) s += acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int KIND, int NACC = 1> static void run(float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    hipLaunchKernelGGL((k<K, KIND, NACC>), dim3(256), dim3(256), 0, 0, d, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, KIND, NACC>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d accumulator(s), %s x %2d between MFMAs: %.1f ns per MFMA = %.1f cycles at 2.4 GHz\n", NACC, KIND ? "ds_read" : "v_fma  ", K, ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.4);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<4, 0>(d); run<6, 0>(d); run<8, 0>(d); run<12, 0>(d); run<16, 0>(d);
    run<1, 1>(d); run<2, 1>(d); run<4, 1>(d); run<8, 1>(d);
    run<0, 0, 2>(d); run<2, 0, 2>(d); run<4, 0, 2>(d); run<8, 0, 2>(d); run<12, 0, 2>(d); run<16, 0, 2>(d); run<2, 1, 2>(d); run<4, 1, 2>(d); run<8, 1, 2>(d);
    return 0;
}
