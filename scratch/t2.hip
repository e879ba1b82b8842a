#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2 b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return __builtin_bit_cast(uint32_t, b);
}
__global__ void k(const float* a, uint32_t* o) { int i = threadIdx.x; o[i] = pack_bf2(a[2*i], a[2*i+1]); }
