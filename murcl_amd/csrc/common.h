// Shared device helpers for the murcl_amd gfx950 kernels (CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MURCL_DTYPE_F32 0
#define MURCL_DTYPE_BF16 1

#define MURCL_CHECK_LAUNCH() ((int)hipGetLastError())

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// 4 consecutive elements <-> f32x4
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    u32x2 u = *(const u32x2*)p;
    return f32x4{bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1])};
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    *(u32x2*)p = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

// LDS byte offset of a __shared__ object (flat LDS address: low 32 bits are the offset).
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(uintptr_t)p; }

// LDS-DMA: each lane copies 16 B from its own global address to  lds_dst + lane*16.
// lds_dst must be wave-uniform.  Invisible to hipcc's waitcnt bookkeeping by design: the
// caller counts these with s_waitcnt vmcnt(N) and a barrier before any ds_read of the bytes
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Raw workgroup barrier that does NOT drain vmcnt (LDS-DMA tiles stay in flight across it):
// own LDS traffic retired first, compiler fenced on both sides.
#define LDS_BARRIER()                                       \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (observed round-robin
// dispatch; speed only, never correctness).  Gives each XCD a contiguous run of tiles so
// neighbouring tiles that share operand panels hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exp via v_exp_f32 (2^x). abs error ~1e-7 relative to range.
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
