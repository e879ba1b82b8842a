// Lab equipment (tools/_abl/lib/probes.so): what does a grid-wide barrier INSIDE a kernel cost on this part, against the ~4.5 us floor
// of a dependent launch?  (the persistent bag-level head / one-kernel PPO epoch were rejected on round-1
// numbers - NT-Xent 33 us with one barrier against 20 us without - and never re-measured.)
// `phases` times: every workgroup writes `payload` floats, all workgroups meet at a counter barrier (release / acquire at agent scope:
// the compiler's L2 write-back + invalidate, which is what makes another XCD's data visible), then every workgroup reads and checks
// the payload a workgroup on ANOTHER XCD wrote in this phase.  Spins are bounded: a barrier that does not complete sets err[1] and the
// kernel runs out instead of hanging the GPU.
#include <hip/hip_runtime.h>

__device__ __forceinline__ bool gb_wait(int* counter, int target, int* err) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20) || __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return true;
}

// two-level form: the workgroups of one XCD meet at that XCD's counter (8 addresses take arrivals in parallel; same-address atomics at
// agent scope serialise at ~29 ns each, which is what makes the flat barrier linear in the workgroup count), the last one to arrive
// there reports to the global counter, everybody polls the global one.  counter[0] global, counter[16 * (1 + xcc)] per XCD.
__device__ __forceinline__ void gb_wait2(int* counter, int phase, int G, int* err) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        int* mine = counter + 16 * (1 + xcc);
        // workgroups per XCD under the b % 8 placement: G / 8 (+1 for the first G % 8)
        const int per = G / 8 + ((int)((blockIdx.x) % 8) < G % 8 ? 1 : 0);
        if (__hip_atomic_fetch_add(mine, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == (phase + 1) * per - 1)
            __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        const int groups = G < 8 ? G : 8;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * groups) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20) || __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void grid_barrier_kernel(int* counter, float* buf, int payload, int phases, int* err, int do_barrier) {
    const int G = gridDim.x, b = blockIdx.x;
    int bad = 0;
    for (int ph = 0; ph < phases; ++ph) {
        float* mine = buf + ((size_t)(ph & 1) * G + b) * payload;
        for (int i = threadIdx.x; i < payload; i += 256) mine[i] = (float)(ph * 1024 + b) + (float)(i & 7);
        if (do_barrier) {
            __syncthreads();
            if (do_barrier == 2) gb_wait2(counter, ph, G, err);
            else gb_wait(counter, (ph + 1) * G, err);
        }
        const int o = (b + 37) % G;                                    // (37 is odd: another XCD under the b % 8 placement)
        const float* theirs = buf + ((size_t)(ph & 1) * G + o) * payload;
        for (int i = threadIdx.x; i < payload; i += 256) bad += (theirs[i] != (float)(ph * 1024 + o) + (float)(i & 7));
    }
    if (bad) atomicAdd(err, bad);
}

extern "C" int murcl_debug_grid_barrier(int grid, int payload_floats, int phases, int do_barrier, int* counter, float* buf, int* err,
                                        hipStream_t stream) {
    hipLaunchKernelGGL(grid_barrier_kernel, dim3(grid), dim3(256), 0, stream, counter, buf, payload_floats, phases, err, do_barrier);
    return (int)hipGetLastError();
}
