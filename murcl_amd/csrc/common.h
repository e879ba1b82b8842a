// Shared device helpers for the murcl_amd gfx950 kernels (CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/murcl_amd.h"      // every extern "C" definition is checked against its public declaration

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MURCL_DTYPE_F32 0
#define MURCL_DTYPE_BF16 1
#define MURCL_DTYPE_F32X3 2      // GEMM entry points only: f32 tensors, products as a 3-term bf16 split on the bf16 matrix pipe

#define MURCL_CHECK_LAUNCH() ((int)hipGetLastError())

extern "C" int murcl_cu_budget(void);      // runtime.hip: CUs the persistent launches size their one round of workgroups for (256 by default)

// hipFuncSetAttribute applies to the current device only: call sites remember which devices they have prepared
// (one process per GPU is the norm, but a process that drives several devices must raise the LDS limit on each).
struct MurclOncePerDevice {
    unsigned long long seen = 0;
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (seen & bit) return false;
        seen |= bit;
        return true;
    }
};

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {          // one v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
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

// Counter-based dropout keep mask (csrc/clam.hip, csrc/elementwise.hip): element i takes byte (i & 7) of
// splitmix64(seed + (i / 8) * C): kept (value `scale`) when the byte is below `thresh` = keep probability in 1/256ths.
__device__ __forceinline__ unsigned long long murcl_splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long murcl_drop_word(unsigned long long seed, long group8) {
    return murcl_splitmix64(seed + (unsigned long long)group8 * 0xD1342543DE82EF95ull);
}

// 8 consecutive elements <-> float[8] (16-byte accesses for bf16, two for f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) {
    const u32x4 u = *(const u32x4*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(u[e]); v[2 * e + 1] = bf_hi(u[e]); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) {
    *(u32x4*)p = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
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
// Same with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset: no per-op 64-bit VALU add.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(bs), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(bs), "s"(dst) : "memory");
}
// Variant for bases the compiler already knows to be wave-uniform (kernel args, blockIdx / readfirstlane'd
// values and scalar arithmetic on them): no v_readfirstlane round trip.
__device__ __forceinline__ void glds16_u(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// the same with the non-temporal cache policy: a stream that is read once should not displace what the NEXT kernel will
// read from the 256 MiB Infinity Cache (e.g. the activations this kernel is writing)
#ifndef GLDS_STREAM_POLICY
#define GLDS_STREAM_POLICY " nt"        // cache-policy suffix of the streaming LDS-DMA load (A/B: " sc1", " sc0 sc1", " nt sc1" ...)
#endif
__device__ __forceinline__ void glds16_u_nt(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" GLDS_STREAM_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
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
    // tanh(x) = 1 - 2/(2^(2x log2 e) + 1): one v_exp_f32 + one v_rcp_f32 (1 ulp each), no IEEE
    // division sequence.  Saturates correctly: e=inf -> 1, e=0 -> -1.
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// DPP cross-lane moves (VALU, no LDS round trip).  CTRL: quad_perm = sel0|sel1<<2|sel2<<4|sel3<<6,
// 0x140 row_mirror, 0x141 row_half_mirror.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// all 16 lanes of each row end with the row's sum / max
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float rdlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48) with the gfx950 row/half swaps - VALU only
__device__ __forceinline__ float quarters_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// the whole wave's sum in every lane, VALU only (4 DPP steps inside the 16-lane rows, then the two gfx950 swaps): __shfl_xor goes
// through the LDS crossbar (ds_bpermute) for the 16- and 32-lane steps, a dependent chain of LDS round trips per reduction
__device__ __forceinline__ float wave_sum_valu(float v) { return quarters_sum(row16_sum(v)); }
// sum over aligned groups of G lanes (G a power of two, wave-uniform), every lane of a group ends with its group's sum
__device__ __forceinline__ float group_sum(float v, int G) {
    if (G == 64) return wave_sum_valu(v);
    if (G == 32) {
        v = row16_sum(v);
        auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    if (G == 16) return row16_sum(v);
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_sum_shfl(float v, int G) {          // the plain butterfly, for A/B builds
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// reductions over lanes 0..31 (rows 0 and 1), result wave-uniform
__device__ __forceinline__ float half_wave_sum(float v) { v = row16_sum(v); return rdlane(v, 0) + rdlane(v, 16); }
__device__ __forceinline__ float half_wave_max(float v) { v = row16_max(v); return fmaxf(rdlane(v, 0), rdlane(v, 16)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
