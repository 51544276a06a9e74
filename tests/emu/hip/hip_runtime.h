// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, not product code.
//
// A stand-in for <hip/hip_runtime.h> that lets the kernel sources under gr-bluetooth_amd/csrc be
// compiled for the HOST (clang++, x86) and executed thread by thread: the build container has no
// GPU, and the index arithmetic of the LDS-tiled kernels (tile spans, bank-aware layouts, lane ->
// task tables, halo handling) is exactly what goes wrong first.  Every thread of a workgroup is
// a ucontext fiber; __syncthreads() parks the fiber until all live fibers of the block arrive,
// so LDS hazards that a missing barrier would cause show up as wrong results here too.  Only
// what the kernels use is provided.  The product is built by hipcc for gfx950 only; nothing
// under gr-bluetooth_amd/ includes this file.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)emu::dyn_lds;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned int x, y; };
struct alignas(16) uint4 { unsigned int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace emu {
extern thread_local dim3 t_idx, b_idx, b_dim, g_dim;
extern char dyn_lds[160 * 1024];
void barrier();
void wave_barrier();
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace emu

#define threadIdx emu::t_idx
#define blockIdx emu::b_idx
#define blockDim emu::b_dim
#define gridDim emu::g_dim
static inline void __syncthreads() { emu::barrier(); }
static inline unsigned long long clock64() { return 0ULL; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned sh)
{
    sh &= 31;
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}
static inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh)
{
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
}
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline void sincospi(double x, double *s, double *c) { *s = std::sin(M_PI * x); *c = std::cos(M_PI * x); }
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }   // fibers: one OS thread
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
// __shfl_down through a per-workgroup exchange buffer and two workgroup barriers (write, read): exact for kernels
// whose shuffles sit in workgroup-uniform control flow or are followed only by the exit of the lanes that skip them
// (the emulator's barrier ignores exited lanes) -- energy_kernel, block_sum_kernel, noise_stage2_kernel.
namespace emu { extern char xchg[1024][16]; }
template <class T> static inline T __shfl_down(T v, int off, int width = 64)
{
    static_assert(sizeof(T) <= 16, "exchange slot");
    const unsigned tid = threadIdx.x;
    std::memcpy(emu::xchg[tid], &v, sizeof(T));
    emu::barrier();
    T r = v;
    const unsigned lane = tid % (unsigned)width;
    if (lane + (unsigned)off < (unsigned)width && tid + (unsigned)off < blockDim.x) std::memcpy(&r, emu::xchg[tid + off], sizeof(T));
    emu::barrier();
    return r;
}
template <class T> static inline T __shfl_up(T v, int off, int width = 64)
{
    static_assert(sizeof(T) <= 16, "exchange slot");
    const unsigned tid = threadIdx.x;
    std::memcpy(emu::xchg[tid], &v, sizeof(T));
    emu::wave_barrier();
    T r = v;
    const unsigned lane = tid % (unsigned)width;
    if (lane >= (unsigned)off) std::memcpy(&r, emu::xchg[tid - off], sizeof(T));
    emu::wave_barrier();
    return r;
}
static inline unsigned long long __ballot(int pred)
{
    const unsigned tid = threadIdx.x, w0 = tid & ~63u;
    emu::xchg[tid][0] = pred ? 1 : 0;
    emu::barrier();
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && w0 + i < blockDim.x; i++) if (emu::xchg[w0 + i][0]) m |= 1ull << i;
    emu::barrier();
    return m;
}
// v_mfma_f32_32x32x2_f32 (exact.hip.h): D = A[32 x 2] * B[2 x 32] + C per wave, lane l holding A[l & 31][l >> 5], B[l >> 5][l & 31] and
// the sixteen C/D elements (row (i & 3) + 8 (i >> 2) + 4 (l >> 5), column l & 31).  On the device it is bit for bit the k-ordered
// fmaf chain below (scripts/ubench/exact_mfma.hip checks that on the hardware).  Operands cross the exchange buffer like the
// shuffles': every lane of the workgroup must arrive (workgroup-uniform control flow).
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int)
{
    const unsigned tid = threadIdx.x, w0 = tid & ~63u, lane = tid & 63u;
    std::memcpy(emu::xchg[tid], &a, 4); std::memcpy(emu::xchg[tid] + 4, &b, 4);
    emu::wave_barrier();
    const unsigned col = lane & 31u;
    for (unsigned i = 0; i < 16; i++) {
        const unsigned row = (i & 3u) + 8u * (i >> 2) + 4u * (lane >> 5);
        float acc = c[i];
        for (unsigned k = 0; k < 2; k++) {
            float av, bv;
            std::memcpy(&av, emu::xchg[w0 + row + 32u * k], 4); std::memcpy(&bv, emu::xchg[w0 + col + 32u * k] + 4, 4);
            acc = std::fmaf(av, bv, acc);
        }
        c[i] = acc;
    }
    emu::wave_barrier();
    return c;
}
