// mfma_rate.hip -- issue rate of v_mfma_f32_32x32x2_f32 chains (one accumulator per wave, dependent, as exact_rows_kernel runs them)
// and of two / four independent accumulators, at 1 .. 5 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void chain(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; k++) for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < NACC; k++) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; k++) for (int i = 0; i < 16; i++) s += acc[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int nacc = 1; nacc <= 4; nacc *= 2)
        for (int wg_per_cu = 1; wg_per_cu <= 5; wg_per_cu++) {          // 4 waves per workgroup: wg_per_cu waves per SIMD
            const int grid = 256 * wg_per_cu;
            auto k = nacc == 1 ? chain<1> : nacc == 2 ? chain<2> : chain<4>;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 10, 1.0f, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfma_per_simd = (double)iters * 16 * nacc * wg_per_cu;
            printf("acc %d, %d waves/SIMD: %.3f ms, %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz, %.1f TFLOP/s\n", nacc, wg_per_cu, ms,
                   ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, mfma_per_simd * 1024 * 4096.0 / ms * 1e-9);
        }
    return 0;
}
