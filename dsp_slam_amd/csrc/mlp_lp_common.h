// Shared by mlp_lp_kernel.hip (the f16 / bf16 prepass: forward only) and mlp_lpj_kernel.hip (the 16-bit forward + input-gradient kernel of the
// low-precision compute mode): operand types, the v_mfma_f32_16x16x32 wrapper, packing, the weight-ring state.
#pragma once
#include "dsp_internal.h"
#include "mlp_common.h"

namespace dsp {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

constexpr int LP_KQ = 4;               // 32-k steps per chunk (a chunk spans 128 slab rows)
constexpr int LP_RT = 4;               // 16-row tiles per 64-row output group
constexpr int LP_FRAG_BYTES = 1024;    // one A fragment: 16 rows x 32 k, 16 B per lane
constexpr int LP_NCH = 4;              // chunks per output group of a hidden layer: 16 steps of 32 k = 512 slab rows
constexpr int LP_NOG = 8;              // 64-row output groups per layer
constexpr int LP_ZERO_BYTES = WIDTH * 4;

template <bool BF>
__device__ __forceinline__ f32x4 lp_mfma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// two fp32 -> one register holding two 16-bit values (element 0 in the low half), round to nearest even:
// v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32
template <bool BF>
__device__ __forceinline__ unsigned lp_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    if constexpr (BF)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
    else
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}

// relu + round + pack of two accumulators.  f16: round first, then ONE v_pk_max_f16 on the pair -- max(round(x), 0) == round(max(x, 0)),
// rounding keeps the sign (a -0 that survives multiplies to a zero product).  bf16 has no packed max on gfx950: relu in fp32, then pack.
template <bool BF>
__device__ __forceinline__ unsigned lp_relu_pack(float lo, float hi) {
    if constexpr (BF) {
        return lp_pack<BF>(relu1(lo), relu1(hi));
    } else {
        unsigned r;
        asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(lp_pack<BF>(lo, hi)));
        return r;
    }
}

template <bool BF>
__device__ __forceinline__ float lp_round(float x) {
    if constexpr (BF)
        return (float)(__bf16)x;
    else
        return (float)(_Float16)x;
}

struct LpRing {            // weight-stream state, all wave-uniform
    int issue_pos, issue_slot, rd_slot, total_chunks;
    const char* wbase;     // this wave's source address inside chunk 0
    unsigned lane_off;     // lane * 16: the per-lane part of every source address
    const char* isrc;      // ... inside the chunk being issued
    unsigned ring0, idst;  // LDS byte addresses: ring start + this wave's quarter; destination of the chunk being issued
    char* ring_ptr;
    unsigned ring_lane;    // LDS byte address of the ring start + lane * 16: base of this lane's A-fragment reads
};

__device__ __forceinline__ void lp_issue_next(LpRing& rg) {
    rg.issue_pos = (rg.issue_pos + 1 == rg.total_chunks) ? 0 : rg.issue_pos + 1;
    rg.issue_slot = (rg.issue_slot + 1 == LP_NBUF) ? 0 : rg.issue_slot + 1;
    rg.isrc = rg.wbase + (size_t)rg.issue_pos * CHUNK_BYTES;
    rg.idst = rg.ring0 + rg.issue_slot * CHUNK_BYTES;
}

// rows 64 g + 16 rt + 4 gq + r of a fp32 table, in accumulator (D) order: dst[rt][r]
__device__ __forceinline__ void lp_load_rows(const float* tab, int g, int gq, f32x4 (&dst)[LP_RT]) {
#pragma unroll
    for (int rt = 0; rt < LP_RT; ++rt) dst[rt] = *reinterpret_cast<const f32x4*>(tab + 64 * g + 16 * rt + 4 * gq);
}

}  // namespace dsp
