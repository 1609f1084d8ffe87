// DeepSDF decoder forward + input gradient in f16 / bf16 MFMA for gfx950 (MI355X): the jacobian kernel of the LOW-PRECISION COMPUTE MODE
// (dsp_batch_set_compute(DSP_COMPUTE_F16), include/dsp_gn.h) -- an opt-in, non-parity fast path: BASELINE.json's north_star names "fp32/bf16
// GEMMs", SURVEY.md section 8(d) allows it "reported separately, never mixed into the fp32 fraction".  The default path (mlp_kernel.hip, fp32
// v_mfma_f32_16x16x4_f32) is untouched by it.
//
// Replaces get_batch_sdf_jacobian (reconstruct/loss_utils.py:82-103) with 16-bit matrix operands and fp32 accumulation -- the precision class
// the reference's authors ran in (PyTorch 1.10 on Ampere multiplies in TF32: 10-bit mantissas like f16's, fp32 accumulation).
//
// Structure: mlp_lp_kernel.hip's (one workgroup = 4 waves = one 128-point tile, a wave = 32 points as two 16-point column blocks, activation
// slabs X / Y in registers as packed 16-bit pairs, v_mfma_f32_16x16x32, weights through an 8-slot LDS ring by LDS-DMA), as TWO kernels over the
// same tile list (one kernel holding both sweeps wants more than the 512 registers a lane has: 197 spilled, measured):
//   mlp_lpj_fwd_kernel   the prepass kernel's eight passes, bit for bit (same stream, same order: the sdf equals mlp_lp_kernel's), and every pass
//                        also emits its relu mask: 128 bits per lane and column block (bit = accumulator > 0), 64 KiB per tile in global memory
//                        (the LDS is full: ring 128 KiB + tables).  Writes the sdf into the point's output row.
//   mlp_lpj_bwd_kernel   builds the backward sweep's input slab from the last hidden layer's mask -- S w_last where the accumulator was positive
//                        (S = 16: keeps small gradient entries out of f16's subnormals; d tanh = 1 - y^2 and 1 / S multiply the result at the end,
//                        in fp32) -- and runs eight passes over the TRANSPOSED weights (a stream of its own, packed by pack_decoder_lpj_host in the
//                        same slot order): the epilogue ANDs each accumulator with its mask bit instead of the relu, rounds and packs.  The
//                        latent_in layer's pass also yields the gradient of the re-injected [xyz | code] rows (unmasked; kept as packed pairs,
//                        20 registers), the first layer's pass leaves d sdf / d [code | xyz] in the accumulators: + the kept rows,
//                        x (1 - y^2) / S, stored as the fp32 kernel stores it (68 floats per point: d/dcode[64], d/dxyz[3], sdf).
// Masks travel through memory that the kernels also stream weights through by hand-counted vmcnt.  The forward kernel's mask stores are
// compiler-managed: they add to the counter, so a counted wait can only wait LONGER than needed (outstanding <= k still implies that at most k
// DMA pieces are in flight).  The backward kernel fetches a layer's masks by LDS-DMA a pass ahead of their use into the (otherwise unused) bias
// area and reads them back from the LDS: a compiler-managed global load would put an s_waitcnt vmcnt(0) in front of the first use and drain
// the weight ring once per pass (measured: profiles/r06_lp_compute.md).
#include "dsp_internal.h"
#include "mlp_common.h"
#include "mlp_lp_common.h"

namespace dsp {

constexpr float LPJ_SEED_SCALE = 16.f;
constexpr int LPJ_SKIP_T0 = 27;            // first 16-row tile that may hold re-injected input rows of the latent_in layer (27: 64-D codes, 29: 32-D)
constexpr int LPJ_SKIP_TILES = 32 - LPJ_SKIP_T0;
constexpr int LPJ_MASK_TILE = 8 * 4 * 2 * 64;      // uint4 per 128-point tile: [layer 8][wave 4][column block 2][lane 64]

// relu-mask bit of one accumulator, shifted into `bits` (bits = 2 * bits + (x > 0)): v_cmp + v_addc (mlp_kernel.hip).  Element i of the 32
// pushed into a word ends at bit 31 - i.
__device__ __forceinline__ void lpj_push_bit(unsigned& bits, float x) {
    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(x) : "vcc");
}
// x where bit (31 - i) of word is set, else +0: one v_bfe_i32 (0 / all ones) + one v_and
__device__ __forceinline__ float lpj_keep(float x, unsigned word, int i) {
    return __int_as_float(__float_as_int(x) & __builtin_amdgcn_sbfe((int)word, 31u - (unsigned)i, 1u));
}

// one layer's masks of one wave (two column blocks, 2 x 1 KiB) from global memory into its staging buffer by LDS-DMA.  Writes M0: only between
// a chunk's last DMA piece and the next chunk's glds_set_dst, i.e. outside lpj_pass
__device__ __forceinline__ void lpj_mask_dma(const char* gsrc_uniform, unsigned lane_off, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:1024"
        :
        : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst)
        : "memory");
}

// KIND of a pass: 0 forward (relu, mask bits out), 1 forward LAST hidden layer (mask bits out + the final dot product; no slab), 2 backward (mask
// bits applied; with `cap`, the latent_in layer's pass, the re-injected rows of its input are kept as well), 3 backward FIRST layer (two output
// groups, accumulators kept)
//
// One half (two accumulators of one 16-row tile and column block) of an epilogue unit.  T, blk, half are compile-time after unrolling.
template <bool BF, int KIND>
__device__ __forceinline__ void lpj_half(int T, int blk, int half, float e0, float e1, u32x4 (&out)[32], unsigned (&mw)[2][4], const float* dp,
                                         int gq, float (&part)[2], unsigned (&skip)[LPJ_SKIP_TILES][2][2], bool cap) {
    const int i0 = 4 * (T & 7) + 2 * half;                  // element index of e0 inside its mask word (T >> 3)
    unsigned packed;
    if constexpr (KIND == 0) {
        lpj_push_bit(mw[blk][T >> 3], e0);
        lpj_push_bit(mw[blk][T >> 3], e1);
        packed = lp_relu_pack<BF>(e0, e1);
    } else if constexpr (KIND == 1) {
        const f32x2 w = *reinterpret_cast<const f32x2*>(dp + 16 * T + 4 * gq + 2 * half);
        lpj_push_bit(mw[blk][T >> 3], e0);
        lpj_push_bit(mw[blk][T >> 3], e1);
        part[blk] = fmaf(relu1(e0), w.x, part[blk]);
        part[blk] = fmaf(relu1(e1), w.y, part[blk]);
        return;                                               // nothing reads the last hidden layer's slab
    } else {
        if (T >= LPJ_SKIP_T0) {          // (compile-time) rows that are re-injected input in the latent_in layer's pass: kept unmasked there (cap is wave-uniform)
            const unsigned raw = lp_pack<BF>(e0, e1);
            skip[T - LPJ_SKIP_T0][blk][half] = cap ? raw : skip[T - LPJ_SKIP_T0][blk][half];
        }
        packed = lp_pack<BF>(lpj_keep(e0, mw[blk][T >> 3], i0), lpj_keep(e1, mw[blk][T >> 3], i0 + 1));
    }
    out[2 * (T >> 1) + blk][2 * (T & 1) + half] = packed;
}

// One dense layer pass over this wave's 32 points (mlp_lp_kernel.hip: lp_pass, with the epilogues above).  NOG output groups of 64 rows.
template <bool BF, int NCH, int KIND, int NOG>
__device__ __forceinline__ void lpj_pass(const LpPass pd, u32x4 (&in)[32], u32x4 (&out)[32], f32x4 (&acc)[2][LP_RT][2], u32x4 (&abuf)[2][LP_RT], LpRing& rg,
                                         const u32x4 (&xb)[2], const float* bp, const float* dp, int gq, float (&part)[2],
                                         unsigned (&mw)[2][4], unsigned (&skip)[LPJ_SKIP_TILES][2][2], bool cap = false) {
    constexpr bool FWD = KIND <= 1;
    // ---- prologue (forward only): the xyz B operands at their fixed step, padding cleared (lp_pass) ----
    if constexpr (FWD) {
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        if (NCH == 1) {
#pragma unroll
            for (int ks = 0; ks < LP_KQ; ++ks)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) in[2 * ks + blk] = ks == 0 ? xb[blk] : zero;
        } else {
            const bool latf = pd.kind == 2;
            constexpr int KX = LP_KQ * NCH - 1;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) in[2 * KX + blk] = latf ? xb[blk] : in[2 * KX + blk];
#pragma unroll
            for (int t = 1; t <= 3; ++t) {
                const int T = 2 * KX - t;
                const bool z = latf && pd.npad >= t;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    in[2 * (T >> 1) + blk][2 * (T & 1) + 0] = z ? 0u : in[2 * (T >> 1) + blk][2 * (T & 1) + 0];
                    in[2 * (T >> 1) + blk][2 * (T & 1) + 1] = z ? 0u : in[2 * (T >> 1) + blk][2 * (T & 1) + 1];
                }
            }
        }
    }
    if constexpr (KIND <= 1) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int w = 0; w < 4; ++w) mw[blk][w] = 0u;
    }
    f32x4 bias[LP_RT];
    lp_load_rows(bp, 0, gq, bias);

#pragma unroll
    for (int g = 0; g < NOG; ++g) {
        const int par = g & 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int nx_slot = (rg.rd_slot + 1 == LP_NBUF) ? 0 : rg.rd_slot + 1;
            typedef const __attribute__((address_space(3))) char* lds_cptr;
            unsigned cb_a = rg.ring_lane + (unsigned)rg.rd_slot * CHUNK_BYTES, nb_a = rg.ring_lane + (unsigned)nx_slot * CHUNK_BYTES;
            asm volatile("" : "+v"(cb_a), "+v"(nb_a));
            const lds_cptr cbp = (lds_cptr)(size_t)cb_a, nbp = (lds_cptr)(size_t)nb_a;
#pragma unroll
            for (int kq = 0; kq < LP_KQ; ++kq) {
                const int ks = LP_KQ * c + kq;
                if (kq == LP_KQ / 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GLDS_PER_CHUNK * (LP_NBUF - 3)) : "memory");
                constexpr int NM = 2 * LP_RT;
                // the previous output group's epilogue, one (row tile, column block) unit per step, its two halves dealt over the step's last four gaps
                const bool epi = KIND != 3 && NCH > 1 && g > 0 && ks >= 1 && ks <= 8;
                const int ert = (ks - 1) >> 1, eblk = (ks - 1) & 1;
                constexpr int E0 = NM - 4;
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int rt = m >> 1, blk = m & 1;
                    if (m == 0) __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0), vmcnt / expcnt untouched
                    if (m < 2) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int f = 2 * m + q;
                            const lds_cptr src = (kq + 1 < LP_KQ) ? cbp + ((kq + 1) * LP_RT + f) * LP_FRAG_BYTES : nbp + f * LP_FRAG_BYTES;
                            abuf[(kq + 1) & 1][f] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(src);
                        }
                    }
                    acc[par][rt][blk] = lp_mfma<BF>(abuf[kq & 1][rt], in[2 * ks + blk], ks == 0 ? bias[rt] : acc[par][rt][blk]);
                    if (kq == LP_KQ / 2 && m == 1) { glds_set_dst(rg.idst); glds_piece_m0<0>(rg.isrc, rg.lane_off, rg.idst); }
                    if (kq == LP_KQ / 2 && m == 2) glds_piece_m0<1>(rg.isrc, rg.lane_off, rg.idst);
                    if (kq == LP_KQ / 2 + 1 && m == 1) glds_piece_m0<2>(rg.isrc, rg.lane_off, rg.idst);
                    if (kq == LP_KQ / 2 + 1 && m == 2) { glds_piece_m0<3>(rg.isrc, rg.lane_off, rg.idst); lp_issue_next(rg); }
                    if (NCH == 1) {          // first layer (four steps in all): one row tile's two units behind the last MFMA of that row tile in step 1
                        if (g > 0 && ks == 1 && blk == 1) {
#pragma unroll
                            for (int b2 = 0; b2 < 2; ++b2) {
                                const f32x4 v = acc[par ^ 1][rt][b2];
                                lpj_half<BF, KIND>(4 * (g - 1) + rt, b2, 0, v.x, v.y, out, mw, dp, gq, part, skip, cap);
                                lpj_half<BF, KIND>(4 * (g - 1) + rt, b2, 1, v.z, v.w, out, mw, dp, gq, part, skip, cap);
                            }
                        }
                    } else if (epi) {
                        const int T = 4 * (g - 1) + ert;
                        if (m == E0 + 0) { e0 = acc[par ^ 1][ert][eblk].x; e1 = acc[par ^ 1][ert][eblk].y; asm volatile("" : "+v"(e0), "+v"(e1)); }
                        if (m == E0 + 1) lpj_half<BF, KIND>(T, eblk, 0, e0, e1, out, mw, dp, gq, part, skip, cap);
                        if (m == E0 + 2) { e0 = acc[par ^ 1][ert][eblk].z; e1 = acc[par ^ 1][ert][eblk].w; asm volatile("" : "+v"(e0), "+v"(e1)); }
                        if (m == E0 + 3) lpj_half<BF, KIND>(T, eblk, 1, e0, e1, out, mw, dp, gq, part, skip, cap);
                    }
                    if (ks == LP_KQ * NCH - 1 && m == 1 && g + 1 < NOG) lp_load_rows(bp, g + 1, gq, bias);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            rg.rd_slot = nx_slot;
        }
    }
    if constexpr (KIND != 3) {        // the last group's epilogue has no MFMAs of its own pass to hide behind
#pragma unroll
        for (int rt = 0; rt < LP_RT; ++rt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 v = acc[(NOG - 1) & 1][rt][blk];
                lpj_half<BF, KIND>(4 * (NOG - 1) + rt, blk, 0, v.x, v.y, out, mw, dp, gq, part, skip, cap);
                lpj_half<BF, KIND>(4 * (NOG - 1) + rt, blk, 1, v.z, v.w, out, mw, dp, gq, part, skip, cap);
            }
    }
}

template <bool BF>
__device__ __forceinline__ f32x2 lpj_unpack(unsigned p) {
    if constexpr (BF) {
        const b2 v = __builtin_bit_cast(b2, p);
        return (f32x2){(float)v[0], (float)v[1]};
    } else {
        const h2 v = __builtin_bit_cast(h2, p);
        return (f32x2){(float)v[0], (float)v[1]};
    }
}

// what both kernels set up: LDS carve-up, the weight ring primed with the first chunks, the first A fragments
#define LPJ_PROLOGUE(EXTRA_LDS_INIT)                                                                                           \
    constexpr int WAVE_PTS = LP_WAVE_PTS, WAVE_BYTES = CHUNK_BYTES / 4;                                                         \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                                 \
    const int tid = threadIdx.x;                                                                                                \
    const int lane = tid & 63;                                                                                                  \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                                  \
    const int gq = lane >> 4;                                                                                                   \
    const int pl = lane & 15;                                                                                                   \
    float* bias_l = reinterpret_cast<float*>(smem);                                                                             \
    float* cb_l = reinterpret_cast<float*>(smem + BIAS_BYTES);                                                                  \
    float* zero_l = reinterpret_cast<float*>(smem + BIAS_BYTES + CODEBIAS_BYTES);                                               \
    char* ring_ptr = smem + BIAS_BYTES + CODEBIAS_BYTES + LP_ZERO_BYTES;                                                        \
    const int n_tiles = *a.n_tiles;                                                                                             \
    if ((int)blockIdx.x >= n_tiles) return;                                                                                     \
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[0] = clock64(); a.clk[1] = wall_clock64(); }                              \
    for (int i = tid; i < a.n_bias_rows * WIDTH; i += 256) bias_l[i] = a.bias_tab[i];                                           \
    for (int i = tid; i < WIDTH; i += 256) { zero_l[i] = 0.f; EXTRA_LDS_INIT; }                                                 \
    __syncthreads();                                                                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                            \
    LpRing rg;                                                                                                                  \
    rg.issue_pos = 0; rg.issue_slot = 0; rg.rd_slot = 0; rg.total_chunks = a.total_chunks;                                      \
    rg.wbase = reinterpret_cast<const char*>(a.wstream) + wave * WAVE_BYTES;                                                    \
    rg.lane_off = lane * 16;                                                                                                    \
    rg.isrc = rg.wbase;                                                                                                         \
    rg.ring0 = lds_addr(ring_ptr) + wave * WAVE_BYTES;                                                                          \
    rg.idst = rg.ring0;                                                                                                         \
    rg.ring_ptr = ring_ptr;                                                                                                     \
    rg.ring_lane = lds_addr(ring_ptr) + lane * 16;                                                                              \
    _Pragma("unroll") for (int i = 0; i < LP_NBUF - 1; ++i) {                                                                   \
        glds_quarter(rg.isrc, rg.lane_off, rg.idst);                                                                            \
        lp_issue_next(rg);                                                                                                      \
    }                                                                                                                           \
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GLDS_PER_CHUNK * (LP_NBUF - 2)) : "memory");                          \
    u32x4 abuf[2][LP_RT];                                                                                                       \
    _Pragma("unroll") for (int rt = 0; rt < LP_RT; ++rt) {                                                                      \
        abuf[0][rt] = *reinterpret_cast<const u32x4*>(ring_ptr + lane * 16 + rt * LP_FRAG_BYTES);                               \
        abuf[1][rt] = (u32x4){0u, 0u, 0u, 0u};                                                                                  \
    }                                                                                                                           \
    u32x4 X[32], Y[32];                                                                                                         \
    f32x4 acc[2][LP_RT][2];                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 32; ++i) { X[i] = (u32x4){0u, 0u, 0u, 0u}; Y[i] = (u32x4){0u, 0u, 0u, 0u}; }          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
        _Pragma("unroll") for (int rt = 0; rt < LP_RT; ++rt)                                                                    \
            _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) acc[i][rt][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};

// ---- forward with mask export: tile t's masks at mask_buf[t * 4096 + ((slot * 4 + wave) * 2 + blk) * 64 + lane], slot = layer ----
template <bool BF>
__global__ __launch_bounds__(256, 1) void mlp_lpj_fwd_kernel(const LpjArgs a) {
    LPJ_PROLOGUE((void)0)
    (void)cb_l;
    const float* wl = bias_l + a.wlast_row * WIDTH;
    unsigned mw[2][4];
    unsigned skip[LPJ_SKIP_TILES][2][2];       // (unused by the forward kinds: folds away)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int4 td = a.tiles[tile];
        bool valid[2];
        int pidx[2];
        float4 pt[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int local = wave * WAVE_PTS + 16 * blk + pl;
            valid[blk] = local < td.y;
            pidx[blk] = td.x + (valid[blk] ? local : 0);
            pt[blk] = a.pts[pidx[blk]];
            if (!valid[blk]) pt[blk] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        reinterpret_cast<float4*>(cb_l)[tid] = reinterpret_cast<const float4*>(a.code_bias + (size_t)td.z * a.code_bias_stride)[tid];
        __syncthreads();

        // split-precision xyz operands (mlp_lp_kernel.hip)
        u32x4 xb[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float xp[4][3];
            const float xyz[3] = {pt[blk].x, pt[blk].y, pt[blk].z};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xp[0][c] = 0.f;
                xp[1][c] = lp_round<BF>(xyz[c]);
                xp[2][c] = lp_round<BF>(xyz[c] - xp[1][c]);
                xp[3][c] = lp_round<BF>(xyz[c] - xp[1][c] - xp[2][c]);
            }
            float kv[32];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const int u = kk >> 4, k16 = kk & 15, t = k16 / 3;
                const int ent = (t < 5) ? LP_XYZ_TERMS[BF ? 1 : 0][u][t] : 0;
                kv[kk] = ent ? xp[ent >> 2][k16 % 3] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned v0 = lp_pack<BF>(kv[2 * q], kv[2 * q + 1]), v1 = lp_pack<BF>(kv[8 + 2 * q], kv[8 + 2 * q + 1]);
                const unsigned v2 = lp_pack<BF>(kv[16 + 2 * q], kv[16 + 2 * q + 1]), v3 = lp_pack<BF>(kv[24 + 2 * q], kv[24 + 2 * q + 1]);
                xb[blk][q] = gq == 0 ? v0 : (gq == 1 ? v1 : (gq == 2 ? v2 : v3));
            }
        }

        float part[2] = {0.f, 0.f};
        uint4* msc = a.mask_buf + (size_t)tile * LPJ_MASK_TILE + wave * 128 + lane;
        auto bias_of = [&](const LpPass& pd) { return pd.bias_row == -2 ? cb_l + WIDTH : (pd.bias_row == -3 ? cb_l : bias_l + pd.bias_row * WIDTH); };
        auto store_masks = [&](int slot) {
            msc[slot * 512] = make_uint4(mw[0][0], mw[0][1], mw[0][2], mw[0][3]);
            msc[slot * 512 + 64] = make_uint4(mw[1][0], mw[1][1], mw[1][2], mw[1][3]);
        };
        // the prepass kernel's passes: first layer Y -> X, then X -> Y / Y -> X pairs, the last hidden layer reads X (eight hidden layers: the host
        // offers this kernel for that depth only)
        lpj_pass<BF, 1, 0, LP_NOG>(a.pass[0], Y, X, acc, abuf, rg, xb, bias_of(a.pass[0]), zero_l, gq, part, mw, skip);
        store_masks(0);
        for (int ps = 1; ps < 7; ps += 2) {
            lpj_pass<BF, LP_NCH, 0, LP_NOG>(a.pass[ps], X, Y, acc, abuf, rg, xb, bias_of(a.pass[ps]), zero_l, gq, part, mw, skip);
            store_masks(ps);
            lpj_pass<BF, LP_NCH, 0, LP_NOG>(a.pass[ps + 1], Y, X, acc, abuf, rg, xb, bias_of(a.pass[ps + 1]), zero_l, gq, part, mw, skip);
            store_masks(ps + 1);
        }
        lpj_pass<BF, LP_NCH, 1, LP_NOG>(a.pass[7], X, Y, acc, abuf, rg, xb, bias_of(a.pass[7]), wl, gq, part, mw, skip);
        store_masks(7);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            part[blk] += __shfl_xor(part[blk], 16);
            part[blk] += __shfl_xor(part[blk], 32);
        }
        const int sb = gq & 1;       // lane group 0 stores the point of column block 0, lane group 1 that of column block 1
        const float y = tanhf((sb ? part[1] : part[0]) + a.b_last);
        if (gq < 2 && (sb ? valid[1] : valid[0])) a.out_grad[(size_t)((sb ? pidx[1] : pidx[0]) + td.w) * GRAD_STRIDE + 67] = y;
        // stores and LDS-DMA share vmcnt and may retire out of order: drain before counting again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = clock64(); a.clk[3] = wall_clock64(); }
}

// ---- backward from the exported masks ----
template <bool BF>
__global__ __launch_bounds__(256, 1) void mlp_lpj_bwd_kernel(const LpjArgs a) {
    // S x the final layer's weights live in the (otherwise unused) per-object code-bias area of the LDS carve-up
    LPJ_PROLOGUE(cb_l[i] = LPJ_SEED_SCALE * a.bias_tab[a.wlast_row * WIDTH + i])
    const float* wls_l = cb_l;
    unsigned mw[2][4];
    unsigned skip[LPJ_SKIP_TILES][2][2];
    u32x4 xb[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
    float part[2] = {0.f, 0.f};
    // mask staging: two 8 KiB buffers in the bias area (the backward sweep adds no bias), [wave][column block][lane] uint4 each
    const unsigned stage0 = lds_addr(bias_l) + wave * 2048;
    const uint4* stage_l = reinterpret_cast<const uint4*>(bias_l) + wave * 128 + lane;
    const char* mbase = reinterpret_cast<const char*>(a.mask_buf) + wave * 2048;
    lpj_mask_dma(mbase + (size_t)blockIdx.x * (LPJ_MASK_TILE * 16) + 7 * 8192, rg.lane_off, stage0);
    lpj_mask_dma(mbase + (size_t)blockIdx.x * (LPJ_MASK_TILE * 16) + 6 * 8192, rg.lane_off, stage0 + 8192);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int4 td = a.tiles[tile];
        bool valid[2];
        int prow[2];
        float y[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int local = wave * WAVE_PTS + 16 * blk + pl;
            valid[blk] = local < td.y;
            prow[blk] = td.x + (valid[blk] ? local : 0) + td.w;
            y[blk] = valid[blk] ? a.out_grad[(size_t)prow[blk] * GRAD_STRIDE + 67] : 0.f;      // the forward kernel's sdf
        }
        // next tile of this workgroup (the last one fetches its own masks again: no branch around the DMA)
        const int ntile = tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile;
        const char* msrc = mbase + (size_t)tile * (LPJ_MASK_TILE * 16), *nsrc = mbase + (size_t)ntile * (LPJ_MASK_TILE * 16);
        // masks of layer `slot`: staged by fetch_masks a pass earlier, read back by the lanes that the DMA wrote them for
        auto fetch_masks = [&](const char* src, int slot, int buf) { lpj_mask_dma(src + slot * 8192, rg.lane_off, stage0 + buf * 8192); };
        auto load_masks = [&](int buf) {
            const uint4 m0 = stage_l[buf * 512], m1 = stage_l[buf * 512 + 64];
            mw[0][0] = m0.x; mw[0][1] = m0.y; mw[0][2] = m0.z; mw[0][3] = m0.w;
            mw[1][0] = m1.x; mw[1][1] = m1.y; mw[1][2] = m1.z; mw[1][3] = m1.w;
        };
#pragma unroll
        for (int t = 0; t < LPJ_SKIP_TILES; ++t)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) { skip[t][b2][0] = 0u; skip[t][b2][1] = 0u; }
        // the sweep's input slab: S w_last where the last hidden layer's accumulator was positive (its mask, slot 7), in the slab's slot order --
        // registers 2 (T & 1), 2 (T & 1) + 1 of Y[2 (T >> 1) + blk] hold rows 16 T + 4 gq + {0, 1}, {2, 3}
        load_masks(0);
#pragma unroll
        for (int T = 0; T < 32; ++T) {
            const f32x4 ws = *reinterpret_cast<const f32x4*>(wls_l + 16 * T + 4 * gq);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const unsigned w = mw[blk][T >> 3];
                const int i0 = 4 * (T & 7);
                Y[2 * (T >> 1) + blk][2 * (T & 1) + 0] = lp_pack<BF>(lpj_keep(ws.x, w, i0), lpj_keep(ws.y, w, i0 + 1));
                Y[2 * (T >> 1) + blk][2 * (T & 1) + 1] = lp_pack<BF>(lpj_keep(ws.z, w, i0 + 2), lpj_keep(ws.w, w, i0 + 3));
            }
        }
        // layers 7 .. 0, straight-line (pass 7 - L of the table, masks of layer L - 1): eight hidden layers with the latent_in layer fourth -- DeepSDF's
        // geometry, the only one the host offers this kernel for.  No loop and no branch: a join with both slabs live costs the compiler a hundred
        // spilled registers (mlp_lp_kernel.hip).
#define LPJ_BWD(KIND_, NOG_, P_, IN_, OUT_, CAP_) lpj_pass<BF, LP_NCH, KIND_, NOG_>(a.pass[P_], IN_, OUT_, acc, abuf, rg, xb, zero_l, zero_l, gq, part, mw, skip, CAP_)
        // staging: buffer 0 holds layer 7's masks and buffer 1 layer 6's when the tile starts (fetched during the previous tile's pass 6 / 7, or
        // ahead of the loop); the pass that reads buffer b fetches the masks of the pass after it into the other buffer, whose last reader is a
        // pass behind.  A fetch is two LDS-DMA pieces older than the 32 (8) chunks the pass then issues: the ring's counted waits (at most 20
        // pieces outstanding) retire it within five chunks.
        // Passes 1 .. 6 are ONE loop body of two passes run three times (the latent_in layer's pass differs from its neighbours by a uniform flag):
        // four pass bodies of ~25 KB instead of eight (203 -> 100 KB of code per tile against a 64 KB instruction cache): 8 % faster, measured.
        load_masks(1); fetch_masks(msrc, 5, 0); LPJ_BWD(2, LP_NOG, 0, Y, X, false);
        for (int it = 0; it < 3; ++it) {
            load_masks(0); fetch_masks(msrc, 4 - 2 * it, 1); LPJ_BWD(2, LP_NOG, 2 * it + 1, X, Y, it == 1);
            load_masks(1); fetch_masks(it < 2 ? msrc : nsrc, it < 2 ? 3 - 2 * it : 7, 0); LPJ_BWD(2, LP_NOG, 2 * it + 2, Y, X, false);
        }
        fetch_masks(nsrc, 6, 1); LPJ_BWD(3, 2, 7, X, Y, false);
#undef LPJ_BWD
        // acc[0][j][blk]: rows 16 j + 4 gq + r of d / d code through the first layer (j < 4); acc[1][0][blk]: lane group 3, registers 1..3 =
        // d / d xyz through the first layer (rows 77..79 of the pass).  + the rows the latent_in layer's pass kept; x (1 - y^2) / S.
        // 64-D codes: xyz in tile 27, code in 28..31; 32-D: xyz in 29, code in 30..31.  Chosen by mask arithmetic (a ternary on the register array
        // becomes a run-time index, which sends the array to scratch memory)
        const unsigned m64 = a.lat_tile == LPJ_SKIP_T0 ? 0xffffffffu : 0u, m32 = ~m64;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const float sc = (1.f - y[blk] * y[blk]) * (1.f / LPJ_SEED_SCALE);
            float* orow = a.out_grad + (size_t)prow[blk] * GRAD_STRIDE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // code tile j of the kept rows: tile lat_tile + 1 + j while it exists
                const unsigned s0 = (skip[1 + j][blk][0] & m64) | (j < 2 ? skip[(3 + j) % LPJ_SKIP_TILES][blk][0] & m32 : 0u);
                const unsigned s1 = (skip[1 + j][blk][1] & m64) | (j < 2 ? skip[(3 + j) % LPJ_SKIP_TILES][blk][1] & m32 : 0u);
                const f32x2 k0 = lpj_unpack<BF>(s0), k1 = lpj_unpack<BF>(s1);
                const f32x4 g4 = acc[0][j][blk];
                if (valid[blk])
                    *reinterpret_cast<float4*>(orow + 16 * j + 4 * gq) = make_float4((g4.x + k0.x) * sc, (g4.y + k0.y) * sc, (g4.z + k1.x) * sc, (g4.w + k1.y) * sc);
            }
            const unsigned x0 = (skip[0][blk][0] & m64) | (skip[2][blk][0] & m32), x1 = (skip[0][blk][1] & m64) | (skip[2][blk][1] & m32);
            const f32x2 k0 = lpj_unpack<BF>(x0), k1 = lpj_unpack<BF>(x1);
            const f32x4 gx = acc[1][0][blk];
            if (valid[blk] && gq == 3) *reinterpret_cast<float4*>(orow + 64) = make_float4((gx.y + k0.y) * sc, (gx.z + k1.x) * sc, (gx.w + k1.y) * sc, y[blk]);
        }
        // stores and LDS-DMA share vmcnt and may retire out of order: drain before counting again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = clock64(); a.clk[3] = wall_clock64(); }
}

template __global__ void mlp_lpj_fwd_kernel<false>(const LpjArgs);
template __global__ void mlp_lpj_fwd_kernel<true>(const LpjArgs);
template __global__ void mlp_lpj_bwd_kernel<false>(const LpjArgs);
template __global__ void mlp_lpj_bwd_kernel<true>(const LpjArgs);

static size_t mlp_lpj_lds_bytes() { return BIAS_BYTES + CODEBIAS_BYTES + LP_ZERO_BYTES + LP_NBUF * CHUNK_BYTES; }

hipError_t mlp_lpj_prepare_device() {
    const void* fns[4] = {reinterpret_cast<const void*>(&mlp_lpj_fwd_kernel<false>), reinterpret_cast<const void*>(&mlp_lpj_fwd_kernel<true>),
                          reinterpret_cast<const void*>(&mlp_lpj_bwd_kernel<false>), reinterpret_cast<const void*>(&mlp_lpj_bwd_kernel<true>)};
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_lpj_lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// which: 0 = forward with mask export (args: the prepass stream and pass table), 1 = backward from the masks (args: the transposed stream)
hipError_t launch_mlp_lpj(int which, bool bf16, const LpjArgs& args, int n_blocks, hipStream_t stream) {
    const size_t lds = mlp_lpj_lds_bytes();
    if (which == 0) {
        if (bf16) hipLaunchKernelGGL((mlp_lpj_fwd_kernel<true>), dim3(n_blocks), dim3(256), lds, stream, args);
        else hipLaunchKernelGGL((mlp_lpj_fwd_kernel<false>), dim3(n_blocks), dim3(256), lds, stream, args);
    } else {
        if (bf16) hipLaunchKernelGGL((mlp_lpj_bwd_kernel<true>), dim3(n_blocks), dim3(256), lds, stream, args);
        else hipLaunchKernelGGL((mlp_lpj_bwd_kernel<false>), dim3(n_blocks), dim3(256), lds, stream, args);
    }
    return hipGetLastError();
}

}  // namespace dsp
