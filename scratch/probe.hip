#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k_axpy(const float* x, float* y, float a, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]=a*x[i]+y[i];}
// one wave: C[16x16] = A[16x32] * B[32x16] via bf16 mfma; A row-major [16][32], Bt row-major [16][32] (B^T)
__global__ void k_mfma(const uint16_t* A, const uint16_t* Bt, float* C){
  int l=threadIdx.x; bf16x8 a,b;
  for(int j=0;j<8;j++){a[j]=A[(l&15)*32+8*(l>>4)+j]; b[j]=Bt[(l&15)*32+8*(l>>4)+j];}
  f32x4 c={0,0,0,0};
  c=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)*4+r)*16+(l&15)]=c[r];
}
// fp32 mfma 32x32x2: A[32][2] row-major, B[2][32]
__global__ void k_mfma32(const float* A,const float* B,float* C){
  int l=threadIdx.x; float a=A[(l&31)*2+(l>>5)], b=B[(l>>5)*32+(l&31)];
  f32x16 c; for(int i=0;i<16;i++)c[i]=0;
  c=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,c,0,0,0);
  for(int r=0;r<16;r++) C[((r&3)+8*(r>>2)+4*(l>>5))*32+(l&31)]=c[r];
}
extern "C" int probe_axpy(const float*x,float*y,float a,int n,hipStream_t s){hipLaunchKernelGGL(k_axpy,dim3((n+255)/256),dim3(256),0,s,x,y,a,n);return (int)hipGetLastError();}
extern "C" int probe_mfma(const uint16_t*A,const uint16_t*Bt,float*C,hipStream_t s){hipLaunchKernelGGL(k_mfma,dim3(1),dim3(64),0,s,A,Bt,C);return (int)hipGetLastError();}
extern "C" int probe_mfma32(const float*A,const float*B,float*C,hipStream_t s){hipLaunchKernelGGL(k_mfma32,dim3(1),dim3(64),0,s,A,B,C);return (int)hipGetLastError();}
