// Shared by mlp_kernel.hip (throughput form) and mlp_split_kernel.hip (latency form): MFMA / LDS-DMA helpers and the
// LDS carve-up constants.
#pragma once
#include "dsp_internal.h"

namespace dsp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NBUF = 6;                       // LDS ring depth (chunks)
constexpr int BIAS_ROWS_MAX = 11;             // 7 hidden biases, final-layer weights, 3 first-layer xyz columns
constexpr int BIAS_BYTES = BIAS_ROWS_MAX * WIDTH * 4;      // 22 KiB
constexpr int CODEBIAS_BYTES = 2 * WIDTH * 4;              // per-object code contribution of layer 0 and of the latent_in layer
constexpr int MASK_SLOTS = 8;
constexpr int MASK_BYTES = MASK_SLOTS * 8 * 256 * 2;       // [slot][og][tid] u16 = 32 KiB
// A operands are read from LDS in groups of PAIR_READS k-steps, one s_waitcnt per group, two groups ahead of their MFMAs: every instruction
// between two MFMAs costs matrix-pipe time even in the MFMA's shadow (measured: one read + one wait per k-step 0.885 of peak, groups of 2:
// 0.903, groups of 4: 0.898; profiles/r06_removed_experiments.md)
constexpr int PAIR_READS = 2;
constexpr int GLDS_PER_CHUNK = 4;             // per wave: 4 x 1 KiB pieces of a 16 KiB chunk

#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// LDS-DMA of one wave's quarter (4 KiB) of a weight chunk: four global_load_lds_dwordx4, each 64 lanes x 16 B from a
// global address to LDS at M0 + lane*16.  The instruction's immediate offset moves BOTH the global source and
// the LDS destination (measured: tools/probes/hw_probe.hip), so one address and one M0 value serve all four.
// Invisible to hipcc's s_waitcnt bookkeeping by design: completion is counted by hand (vmcnt) below.
// Addressing: the stream position is wave-uniform, so it travels as an SGPR pair (saddr) and the per-lane part is the constant
// 32-bit offset lane*16 -- half the address registers the 64-bit per-lane form sends through the address unit per instruction
// (measured: K1 0.862 -> 0.883 of peak against the per-lane form).
// M0 is written and NOT restored: hipcc treats M0 as reserved and sets it itself right before any instruction of its own that needs
// it; in these kernels it emits none (gfx9+ LDS instructions do not read M0), which tests/test_cabi_symbols.py checks on the built
// code objects (every M0 reference must be one of these writes).  Two SALU instructions fewer per piece, in the MFMA stream.
__device__ __forceinline__ void glds_quarter(const char* gsrc_uniform, unsigned lane_off, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:3072"
        :
        : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst)
        : "memory");
}

// The four pieces of a chunk share one LDS destination base: M0 is set once (glds_set_dst, at the barrier that frees the slot) and the
// pieces are bare loads whose immediate offset moves source and destination alike -- 5 instructions per chunk and wave in the MFMA stream
// (a form in which every piece writes M0 itself: 12; profiles/r06_removed_experiments.md).  M0 has to survive the k-steps in between;
// nothing else in these kernels writes it (tests/test_cabi_symbols.py checks the code objects).
__device__ __forceinline__ void glds_set_dst(unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds_dst) : "memory");
}
template <int PIECE>
__device__ __forceinline__ void glds_piece_m0(const char* gsrc_uniform, unsigned lane_off, unsigned lds_dst) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane_off), "s"(gsrc_uniform), "n"(PIECE * 1024) : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// relu as ONE instruction: fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max before the real one
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Always-on guard of the low-precision pre-classification (DESIGN.md "Prepass").  The prepass is exact as long as
// |sdf_lp - sdf_fp32| < lp_delta for every sample it classifies.  The fp32 kernels re-decode every sample the prepass left inside the
// widened band, plus a stratified sample of the ones it classified (k_band_*: guard samples), and each of those still holds its prepass value
// where the fp32 value is about to be stored: compare them.  A difference of half the object's margin or more is a TRIP (counted per
// wave, per object): the host then re-runs the batch with the prepass off (batch_run) -- results never depend on a margin that the
// workload itself has shown to be thin.  The largest difference seen is kept as well (dsp_stats.prepass_guard_max_err).
// `active`: this lane stores a sample's fp32 sdf; old_lp: what the slot held (1.0f = never decoded by the prepass: no comparison).
// tol: half the object's margin, 0.5f * float(guard word 0 of the object) -- passed in so that a caller can fetch it early
__device__ __forceinline__ void prepass_guard_tol(const MlpArgs& a, int obj, bool active, float old_lp, float y, float tol) {
    float err = 0.f;
    if (active && old_lp != 1.0f) {
        err = fabsf(old_lp - y);
        if (!(err < 1.0f)) err = 1.0f;                  // NaN / inf prepass value
    }
    float m = err;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (m > 0.f && (threadIdx.x & 63) == 0) {
        unsigned* w = a.guard + (size_t)obj * a.guard_stride;
        atomicMax(w + 2, __float_as_uint(m));
        if (!(m < tol)) atomicAdd(w + 1, 1u);
    }
}

// the same with the tolerance read here (callers that have no early point to fetch it from)
__device__ __forceinline__ void prepass_guard(const MlpArgs& a, int obj, bool active, float old_lp, float y) {
    prepass_guard_tol(a, obj, active, old_lp, y, 0.5f * __uint_as_float(a.guard[(size_t)obj * a.guard_stride]));
}

}  // namespace dsp
