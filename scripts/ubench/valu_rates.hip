// valu_rates.hip -- gfx950 issue-rate probes behind the VALU roofline of DESIGN.md:
// cycles per wave64 instruction per SIMD for v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32,
// v_mul_f32, v_rcp_f32, v_cndmask, at 1 / 2 / 4 waves per SIMD (s_memtime cycles of one wave
// divided by the instructions all waves of its SIMD issued in that time).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long *cyc, float *sink, int iters)
{
    v2f acc[8];
    float s[8];
    for (int k = 0; k < 8; k++) { acc[k] = (v2f){1.0f + k, 2.0f + k}; s[k] = 1.0f + k; }
    v2f x = {1.0001f, 0.9999f}, y = {1e-6f, -1e-6f};
    float xs = 1.0001f, ys = 1e-6f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[k]) : "v"(xs), "v"(ys));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(x), "v"(y));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(x));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(y));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[k]) : "v"(xs));
            if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[k]) : "v"(ys));
            if (MODE == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[k]) : "v"(ys));
            if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(s[k]) : "v"(ys));
        }
    }
    const unsigned long long t1 = clock64();
    float r = 0.f;
    for (int k = 0; k < 8; k++) r += acc[k].x + acc[k].y + s[k];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (r == 123.456f) sink[0] = r;
}

template <int MODE>
void run(const char *name)
{
    unsigned long long *d_c; float *d_s;
    hipMalloc(&d_c, 4096 * sizeof(unsigned long long)); hipMalloc(&d_s, 4);
    const int iters = 4096;
    printf("%-14s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;          // wps waves on each of the CU's 4 SIMDs
        probe<MODE><<<256, threads>>>(d_c, d_s, iters);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        probe<MODE><<<256, threads>>>(d_c, d_s, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> c(256);
        hipMemcpy(c.data(), d_c, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : c) avg += (double)v; avg /= 256;
        // one wave issued iters*8 instructions while wps waves shared its SIMD
        printf("  wps=%d: %.2f cyc/inst/SIMD (%.3f ms, %.2f GHz eff)", wps, avg / (iters * 8.0 * wps), ms, avg / (ms * 1e6));
    }
    printf("\n");
    hipFree(d_c); hipFree(d_s);
}

// 8-byte coalesced loads from an L2-resident span (the access pattern of a polyphase branch filter
// that reads its input straight from global memory): GB/s per CU by loads in flight
__global__ __launch_bounds__(256) void gload8(const float2 *__restrict__ x, int span, int reps, float *sink)
{
    float2 acc = {0.f, 0.f};
    const int base = (blockIdx.x * 977) % (span - 4096);
    for (int r = 0; r < reps; r++) {
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = x[base + ((r * 131 + k * 100 + threadIdx.x) & 4095) + (k & 1)];
#pragma unroll
        for (int k = 0; k < 16; k++) { acc.x += v[k].x; acc.y += v[k].y; }
    }
    if (acc.x == 123.456f) sink[0] = acc.y;
}

int main()
{
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<3>("v_pk_add_f32");
    run<4>("v_mul_f32"); run<7>("v_add_f32"); run<8>("v_max_f32"); run<9>("v_mov_b32"); run<6>("v_cndmask_b32"); run<5>("v_rcp_f32");
    const int span = 1 << 20;               // 8 MB of float2: L2 / MALL resident
    float2 *d_x; float *d_s; hipMalloc(&d_x, span * sizeof(float2)); hipMalloc(&d_s, 4);
    hipMemset(d_x, 0, span * sizeof(float2));
    for (int wg : {256 * 2, 256 * 4, 256 * 6}) {
        const int reps = 2000;
        gload8<<<wg, 256>>>(d_x, span, 10, d_s); hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        gload8<<<wg, 256>>>(d_x, span, reps, d_s);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)wg * 256 * reps * 16 * 8;
        printf("gload8 %d WG/CU: %.1f GB/s total, %.1f B/clk/CU at 2.4 GHz\n", wg / 256, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}
