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
    unsigned long long msk = 0x5555555555555555ull + blockIdx.x, mo[2] = {0, 0};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int k8 = 0; k8 < 64; k8++) { const int k = k8 & 7;
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
            if (MODE == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s[k]) : "v"(xs), "s"(msk));
            if (MODE == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(s[k]), "v"(xs) : "vcc");
            if (MODE == 12) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mo[k & 1]) : "v"(s[k]), "v"(xs));
            if (MODE == 13) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(s[k]) : "v"(xs), "v"(ys));
            if (MODE == 14) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[k]) : "v"(xs), "v"(ys));
            if (MODE == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 17) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 19) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(acc[k]) : "v"(x));
            if (MODE == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(xs), "v"(ys) : "vcc");
            if (MODE == 21) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(s[k]) : "v"(xs));
            if (MODE == 22) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 23) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(s[k]));
            if (MODE == 24) asm volatile("v_rndne_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 25) asm volatile("v_floor_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 26) asm volatile("v_fract_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 27) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[k]) : "v"(xs), "v"(ys));
            if (MODE == 28) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[k]) : "v"(x));
            if (MODE == 29) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[k]) : "v"(x));
            if (MODE == 30) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(x), "v"(y));
            if (MODE == 31) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[k]) : "v"(ys));
            if (MODE == 32) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(acc[k]) : "v"(x));
            if (MODE == 33) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 34) asm volatile("v_sin_f32 %0, %0" : "+v"(s[k]));
            if (MODE == 35) asm volatile("s_nop 0");
        }
    }
    const unsigned long long t1 = clock64();
    float r = 0.f;
    for (int k = 0; k < 8; k++) r += acc[k].x + acc[k].y + s[k];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (r == 123.456f || (mo[0] ^ mo[1]) == 0x1234567ull) sink[0] = r;
}

template <int MODE>
void run(const char *name)
{
    unsigned long long *d_c; float *d_s;
    hipMalloc(&d_c, 4096 * sizeof(unsigned long long)); hipMalloc(&d_s, 4);
    const int iters = 1 << 15;
    printf("%-14s", name);
    for (int wps : {1, 2, 4}) {
        if (wps == 4 && MODE >= 10) continue;
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
        printf("  wps=%d: %6.2f memtime-cyc/inst/SIMD, %6.3f Ginst/s/SIMD (%.2f ms)", wps, avg / (iters * 8.0 * wps), iters * 8.0 * wps / (ms * 1e6), ms);
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
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_pk_mul_f32");
    run<3>("v_pk_add_f32");
    run<4>("v_mul_f32");
    run<5>("v_rcp_f32");
    run<6>("v_cndmask_vcc");
    run<7>("v_add_f32");
    run<8>("v_max_f32");
    run<9>("v_mov_b32");
    run<10>("v_cndmask_sgpr");
    run<11>("v_cmp_lt_f32");
    run<12>("v_cmp_e64_sgpr");
    run<13>("v_bfi_b32");
    run<14>("v_med3_f32");
    run<15>("v_and_b32");
    run<16>("v_xor_b32");
    run<17>("v_lshl_add_u32");
    run<18>("v_add_u32");
    run<19>("v_lshl_add_u64");
    run<20>("v_mad_u64_u32");
    run<21>("v_mul_lo_u32");
    run<22>("v_cvt_i32_f32");
    run<23>("v_cvt_f32_i32");
    run<24>("v_rndne_f32");
    run<25>("v_floor_f32");
    run<26>("v_fract_f32");
    run<27>("v_max3_f32");
    run<28>("v_mul_f64");
    run<29>("v_add_f64");
    run<30>("v_fma_f64");
    run<31>("v_mov_dpp_shr1");
    run<32>("v_pk_mul_opsel");
    run<33>("v_sqrt_f32");
    run<34>("v_sin_f32");
    run<35>("s_nop0");
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
