#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const uint16_t* g, float* out, const float* fin, uint16_t* bo){
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  int l=threadIdx.x;
  // asm glds
  unsigned keep; unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const uint16_t* src = g + l*8;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(ldsaddr) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (l&15)*4));
  out[l]=v[0]+v[1]+v[2]+v[3];
  __bf16 b=(__bf16)fin[l]; bo[l]=*(uint16_t*)&b;
  f32x4 c={0,0,0,0}; c=__builtin_amdgcn_mfma_f32_16x16x4f32(fin[l],fin[l+64],c,0,0,0); out[64+l]=c[0];
}
