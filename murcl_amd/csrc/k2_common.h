// Shared pieces of the K2 streaming kernels (attn_pool.hip forward, attn_pool_bwd.hip backward): tile geometry,
// LDS-DMA tile issue, division-free work-list walker, MFMA wrappers.
#pragma once
#include "common.h"

#define K2_L 512
#define K2_D 128
#define K2_TR 16                  // rows per tile (one MFMA row tile)
#define K2_NSLOT 4                // ring slots per workgroup; three tiles in flight while one is consumed
#define K2_PSTRIDE (K2_L + 4)      // floats per chunk partial in the forward workspace: [L] sum p.H, then (m, l), 2 spare - rows 16-byte aligned

// NW waves per workgroup, each owning a DW = 128/NW column slice of Wa in registers:
//   bf16: NW = 4 (DW = 32, 128 VGPRs of weights), TWO workgroups per CU - they desynchronise naturally, so
//         one workgroup's barrier / LDS latencies are covered by the other's MFMA and VALU work;
//   f32 : NW = 8 (DW = 16, 128 VGPRs of weights), one workgroup per CU (parity path).
template <typename T, int NWO = 0> struct K2 {      // NWO: waves per workgroup override (0 = the per-dtype default)
    static constexpr int NW = NWO ? NWO : ((sizeof(T) == 2) ? 4 : 8);
    static constexpr int DW = K2_D / NW;                     // D columns per wave: 32 / 16
    static constexpr int NJ = DW / 16;                       // MFMA column tiles per wave: 2 / 1
    static constexpr int ROWB = K2_L * (int)sizeof(T);       // bytes per row in HBM: 1024 / 2048
    static constexpr int PADB = ROWB + 16;                   // LDS row stride: +16 B rotates rows over the 16 bank slots
    static constexpr int TR = K2_TR;
    static constexpr int SLOT = TR * PADB;                   // 16.25 KiB / 32.25 KiB
    static constexpr int GT = TR * ROWB / (NW * 1024);       // LDS-DMA instructions per wave per tile: 4 / 4
    static constexpr int NKK = ROWB / 64;                    // MFMA k-steps: 16 / 32
    static constexpr int PC = K2_L / NW;                     // pooled columns per wave: 128 / 64
    static constexpr int NPJ = PC / 16;                      // pooling MFMA column tiles per wave: 8 / 4
    static constexpr int WG_PER_CU = (sizeof(T) == 2) ? 2 : 1;
    static constexpr int MIN_WAVES = WG_PER_CU * NW / 4;    // waves per SIMD the launch bound must leave room for
};

// LDS carve (bytes): ring | spart [NW waves][16 rows] f32 | pbuf [NW waves][16] u32 | sbuf [<= chunk rows] f32
#define K2_MAX_CHUNK 1024
template <typename T, int NWO = 0> struct K2Lds {
    static constexpr int OFF_SPART = K2_NSLOT * K2<T, NWO>::SLOT;
    static constexpr int OFF_PBUF = OFF_SPART + 2 * K2<T, NWO>::NW * 16 * 4;    // spart is double-buffered
    static constexpr int OFF_SBUF = OFF_PBUF + K2<T, NWO>::NW * 32 * 4;         // pbuf: two tiles' weights per wave (paired forward)
    static constexpr int BYTES = OFF_SBUF + K2_MAX_CHUNK * 4;
};

template <typename T> struct WFrag;
template <> struct WFrag<bf16_t> { typedef bf16x8 type; };
template <> struct WFrag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 k2_mma(typename WFrag<T>::type a, typename WFrag<T>::type b, f32x4 c);
template <> __device__ __forceinline__ f32x4 k2_mma<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 k2_mma<float>(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

// Tile issue shared by forward and backward: LDS-DMA instruction ii = j*NW + wave copies 1 KiB =
// one row (bf16) / half a row (f32); global base is wave-uniform (SGPRs), per-lane offset = lane*16.
#ifndef K2_LOAD_NT
#define K2_LOAD_NT 0              // 1: H tiles are loaded with the non-temporal policy (read once per pass: they should not displace the rows
#endif                            // the pass has not reached yet from the Infinity Cache).  Measured in the step: K2 forward 54.3 -> 58.5 us - off
#define K2_GLDS(base, voff, dst) do { if (K2_LOAD_NT) glds16_u_nt(base, voff, dst); else glds16_u(base, voff, dst); } while (0)
template <typename T, int NWO = 0>
__device__ __forceinline__ void k2_issue_tile(const T* bag_base, int row0, int N, unsigned slot_lds, int wave, int lane) {
    typedef K2<T, NWO> C_;
    const unsigned voff = lane * 16;
    if (row0 + C_::TR <= N) {
        // whole tile inside the bag (all but the last tile of a ragged bag): one scalar base, constant strides
        const char* base = (const char*)bag_base + (size_t)row0 * C_::ROWB;
#pragma unroll
        for (int j = 0; j < C_::GT; ++j) {
            const int ii = j * C_::NW + wave;
            const int row = (sizeof(T) == 2) ? ii : (ii >> 1);
            const int half = (sizeof(T) == 2) ? 0 : (ii & 1);
            K2_GLDS(base + (size_t)(row * C_::ROWB + half * 1024), voff, slot_lds + row * C_::PADB + half * 1024);
        }
    } else {
#pragma unroll
        for (int j = 0; j < C_::GT; ++j) {
            const int ii = j * C_::NW + wave;
            const int row = (sizeof(T) == 2) ? ii : (ii >> 1);
            const int half = (sizeof(T) == 2) ? 0 : (ii & 1);
            const int grow = min(row0 + row, N - 1);              // rows past N: clamped, masked by the caller
            K2_GLDS((const char*)bag_base + (size_t)grow * C_::ROWB + half * 1024, voff,
                    slot_lds + row * C_::PADB + half * 1024);
        }
    }
}

// One LDS-DMA piece (j of GT) of a tile: the loops that spread a tile's issue over their MFMA k-groups use this form.
template <typename T, int NWO = 0>
__device__ __forceinline__ void k2_issue_piece(const T* bag_base, int row0, int N, unsigned slot_lds, int wave, int lane, int j) {
    typedef K2<T, NWO> C_;
    const int ii = j * C_::NW + wave;
    const int row = (sizeof(T) == 2) ? ii : (ii >> 1);
    const int half = (sizeof(T) == 2) ? 0 : (ii & 1);
    const int grow = min(row0 + row, N - 1);                      // rows past N: clamped, masked by the caller
    K2_GLDS((const char*)bag_base + (size_t)grow * C_::ROWB + half * 1024, lane * 16, slot_lds + row * C_::PADB + half * 1024);
}

// Walks a workgroup's tile sequence without integer division in the loop: item = blockIdx + k*gridDim,
// tile-in-item `tin`; (bag, chunk) are re-derived only when the item changes.
struct K2Pos {
    int tin, item, bag, ch, last;      // last >= 0: walk the items from the end (item' = last - item)
    __device__ __forceinline__ void locate(int S) { const int it = last >= 0 ? last - item : item; bag = it / S; ch = it - bag * S; }
    __device__ __forceinline__ void init(int first_item, int S, int last_item = -1) { tin = 0; item = first_item; last = last_item; locate(S); }
    __device__ __forceinline__ void next(int tiles_per_item, int stride, int S) {
        if (++tin == tiles_per_item) { tin = 0; item += stride; locate(S); }
    }
};
