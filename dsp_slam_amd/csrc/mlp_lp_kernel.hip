// DeepSDF decoder forward in f16 / bf16 MFMA for gfx950 (MI355X): the PREPASS that classifies ray samples.
//
// Not a replacement for the fp32 decoder (mlp_kernel.hip) -- an exact filter in front of it.  The render term only looks at
// clamp(sdf, -th, th) (reconstruct/loss_utils.py:40-48, reconstruct/loss.py:84-96): occupancy is exactly 0 for sdf >= th
// and exactly 1 for sdf <= -th.  A sample whose low-precision sdf is farther than a calibrated margin delta from the band
// |sdf| < th is therefore classified for good, bit-exactly, and only the samples inside the widened band are decoded in
// fp32 (dsp_gn.hip, "prepass").
//
// Structure (DESIGN.md "K0"): one workgroup = 4 waves = one 128-point tile, one wave per SIMD.  A wave owns 32 points as TWO column
// blocks of 16 and keeps its [512 rows x 32 points] activation slab in registers as packed 16-bit pairs (128 registers; two slabs X, Y
// ping-pong between layers).  Round 5: the matrix instruction is v_mfma_f32_16x16x32_{f16,bf16} (rounds 2-4: 32x32x16).  A register-only
// probe with live (random) operand data holds 2.13 GHz on the 16x16x32 form and 1.78 GHz on the 32x32x16 form (same FLOP per cycle:
// tools/probes/mfma16_probe.hip, profiles/r05_k0_clock.md) -- the chip's clock under dense 16-bit MFMA follows the switching power, and the
// 16x16 form switches less per FLOP.  One A fragment (16 output rows x 32 k, one ds_read_b128 per lane) feeds both column blocks: the
// LDS duty is what it was (1 KiB per 32 matrix-pipe cycles and wave).
//
// Lane maps: A lane l = row l & 15, k slots 8 (l >> 4) + e; B lane l = point l & 15, the same k slots (the pairing of A and B slots is all
// that matters); D lane l = point l & 15, rows 4 (l >> 4) + r of the 16-row tile.  After v_cvt_pk the D registers of row tiles 2q, 2q + 1
// ARE the B operand of the next layer's 32-k step q if that layer's weights are packed with the slot order
// krow(q, gq, e) = 32 q + 16 (e >> 2) + 4 gq + (e & 3)  (pack_decoder_lp_host; tests/lp_emulator.py pins it on the CPU).
// Weights stream through an 8-slot LDS ring of 16 KiB chunks (4 steps of 32 k x 4 row tiles x 1 KiB A fragment) by LDS-DMA, one piece
// behind an MFMA, counted vmcnt + one s_barrier per chunk -- the protocol of mlp_kernel.hip at 4x the chunk rate.  The relu / v_cvt_pk
// epilogue of output group g-1 is interleaved with the MFMAs of group g (two accumulator sets).  xyz enters layer 0 and the latent_in
// layer through one 32-k step of split-precision products (LP_XYZ_TERMS), the code through the fp32 per-object bias (k_code_bias),
// biases are the fp32 C operand of each tile's first MFMA, and the final 512 -> 1 layer + tanh is an fp32 VALU dot product on the
// un-rounded accumulators of the last hidden layer.
#include "dsp_internal.h"
#include "mlp_common.h"
#include "mlp_lp_common.h"

namespace dsp {

// relu + round + pack the accumulators of row tiles [rt0, rt1) of output group g into the next layer's input slab, and the same values
// times the rows of `dp` into the final-layer dot product (LAST: the last hidden layer writes no slab):
// row tile T = 4 g + rt is half (T & 1) of the next layer's 32-k step T >> 1: registers 2 (T & 1), 2 (T & 1) + 1 of out[2 (T >> 1) + blk].
// (g, rt0, rt1 are compile-time constants after the callers' loops are unrolled; the loop bounds here are literal so that every register
// index folds.)
template <bool BF, bool LAST>
__device__ __forceinline__ void lp_epilogue(int g, int rt0, int rt1, const f32x4 (&acc)[LP_RT][2], const float* dp, int gq, u32x4 (&out)[32], float (&part)[2],
                                            int blk0 = 0, int blk1 = 2) {
#pragma unroll
    for (int rt = 0; rt < LP_RT; ++rt) {
        if (rt >= rt0 && rt < rt1) {
            if (LAST) {
                // last hidden layer: nothing reads its slab; only the final 512 -> 1 layer's dot product on the un-rounded values
                const f32x4 w = *reinterpret_cast<const f32x4*>(dp + 64 * g + 16 * rt + 4 * gq);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    if (blk >= blk0 && blk < blk1) {
                        part[blk] = fmaf(relu1(acc[rt][blk].x), w.x, part[blk]);
                        part[blk] = fmaf(relu1(acc[rt][blk].y), w.y, part[blk]);
                        part[blk] = fmaf(relu1(acc[rt][blk].z), w.z, part[blk]);
                        part[blk] = fmaf(relu1(acc[rt][blk].w), w.w, part[blk]);
                    }
                }
            } else {
                const int T = 4 * g + rt;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    if (blk >= blk0 && blk < blk1) {
                        out[2 * (T >> 1) + blk][2 * (T & 1) + 0] = lp_relu_pack<BF>(acc[rt][blk].x, acc[rt][blk].y);
                        out[2 * (T >> 1) + blk][2 * (T & 1) + 1] = lp_relu_pack<BF>(acc[rt][blk].z, acc[rt][blk].w);
                    }
                }
            }
        }
    }
}

// One dense layer: out = relu(W . in + bias) for this wave's 32 points; part += relu(.) . dot row.  `in` / `out` are the two
// register slabs, indexed [2 ks + blk]: 32-k step ks, column block blk.  A pass is 8 output groups x NCH chunks of straight-line code:
// everything that differs between layers is data (bias / dot-row pointers, prologue selects) -- hipcc answers run-time control flow inside
// this body with hundreds of register moves at every join.  NCH = 1 for the first layer (its K is the xyz step only), LP_NCH for the others.
// NBLK = 2: the wave's 32 points as two column blocks (128-point tiles, the throughput form); NBLK = 1: ONE column block of 16 points
// (64-point tiles: a detection-sized list -- ~117 tiles of 128 points on 256 CUs -- becomes ~235 tiles of half the length; the same
// arithmetic per point, so the same values).
template <bool BF, int NCH, bool LAST, int NBLK>
__device__ __forceinline__ void lp_pass(const LpPass pd, u32x4 (&in)[32], u32x4 (&out)[32], f32x4 (&acc)[2][LP_RT][2],
                                        u32x4 (&abuf)[2][LP_RT], LpRing& rg, const u32x4 (&xb)[2], const float* bp,
                                        const float* dp, int lane, int gq, float (&part)[2]) {
    // ---- prologue: place the xyz B operands at their fixed step -------------------------------------------------------
    // first layer: step 0 (the rest of its single chunk is padding); latent_in layer: the last step (15), behind the slab rows and
    // pd.npad padding 16-row tiles.  Padding meets zero A fragments: clear it so that no stale inf / nan of an earlier layer does.
    // (Selects, not branches: hipcc sinks the stores of two branches into one store through a pointer phi, which pins the whole slab
    // in scratch memory.)
    {
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        if (NCH == 1) {
#pragma unroll
            for (int ks = 0; ks < LP_KQ; ++ks)
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) in[2 * ks + blk] = ks == 0 ? xb[blk] : zero;
        } else {
            const bool lat = pd.kind == 2;
            constexpr int KX = LP_KQ * NCH - 1;          // the xyz step of the latent_in layer
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) in[2 * KX + blk] = lat ? xb[blk] : in[2 * KX + blk];
#pragma unroll
            for (int t = 1; t <= 3; ++t) {               // padding 16-row tile 2 KX - t = half (t & 1 ? 1 : 0) of step (2 KX - t) >> 1
                const int T = 2 * KX - t;
                const bool z = lat && pd.npad >= t;
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    in[2 * (T >> 1) + blk][2 * (T & 1) + 0] = z ? 0u : in[2 * (T >> 1) + blk][2 * (T & 1) + 0];
                    in[2 * (T >> 1) + blk][2 * (T & 1) + 1] = z ? 0u : in[2 * (T >> 1) + blk][2 * (T & 1) + 1];
                }
            }
        }
    }
    f32x4 bias[LP_RT];
    lp_load_rows(bp, 0, gq, bias);

#pragma unroll
    for (int g = 0; g < LP_NOG; ++g) {
        const int par = g & 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int nx_slot = (rg.rd_slot + 1 == LP_NBUF) ? 0 : rg.rd_slot + 1;
            // This lane's LDS byte address inside the chunk being read and inside the next one, each as ONE opaque 32-bit register: every
            // A-fragment read below is then `ds_read_b128 v, base offset:imm`.  Left to itself hipcc materialises a separate address for
            // every (slot, step, row tile), parks them in AGPRs and pays a v_accvgpr_read (often two) per ds_read.
            typedef const __attribute__((address_space(3))) char* lds_cptr;
            unsigned cb_a = rg.ring_lane + (unsigned)rg.rd_slot * CHUNK_BYTES, nb_a = rg.ring_lane + (unsigned)nx_slot * CHUNK_BYTES;
            asm volatile("" : "+v"(cb_a), "+v"(nb_a));
            const lds_cptr cbp = (lds_cptr)(size_t)cb_a, nbp = (lds_cptr)(size_t)nb_a;
#pragma unroll
            for (int kq = 0; kq < LP_KQ; ++kq) {
                const int ks = LP_KQ * c + kq;
                if (kq == LP_KQ / 2) {
                    // chunk q+1 has landed for this wave once <= LP_NBUF-3 younger chunks are in flight; the barrier
                    // publishes every wave's quarter and proves all reads of chunk q-1 retired (mlp_kernel.hip)
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GLDS_PER_CHUNK * (LP_NBUF - 3)) : "memory");
                }
                // One step = eight MFMAs of 16 cycles (row tile m >> 1, column block m & 1).  A 16-cycle MFMA leaves this one wave THREE issue
                // slots, so everything else is dealt out over the eight gaps and pinned there (sched_barrier after every MFMA):
                //   gap 0: ONE lgkmcnt(0) -- the four A fragments of this step were read in gaps 0, 1 of the previous step, seven MFMAs ago --
                //          then the reads of fragments 0, 1 of the NEXT step;   gap 1: fragments 2, 3;
                //   gaps 1, 2 of steps 2, 3: the chunk's four LDS-DMA pieces;
                //   gaps 4 .. 7: one (row tile, column block) unit of the PREVIOUS output group's relu / v_cvt_pk epilogue, two instructions
                //          a gap (steps 1 .. 8 carry the eight units).  hipcc left alone sinks the reads behind the sixth MFMA and bunches the
                //          epilogue behind one step: measured 0.65 duty against 0.74 for the 32x32x16 form.
                constexpr int NM = NBLK * LP_RT;                          // MFMAs per step
                const bool epi = NCH > 1 && g > 0 && ks >= 1 && ks <= 4 * NBLK;
                const int ert = NBLK == 2 ? (ks - 1) >> 1 : ks - 1, eblk = NBLK == 2 ? (ks - 1) & 1 : 0;       // this step's epilogue unit
                constexpr int E0 = NM - 4;                                // the unit's four micro-steps sit in the step's last four gaps
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int rt = NBLK == 2 ? m >> 1 : m, blk = NBLK == 2 ? m & 1 : 0;
                    if (m == 0) __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0), vmcnt / expcnt untouched
                    if (m < 2) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int f = 2 * m + q;        // fragment (= row tile) of step kq + 1
                            const lds_cptr src = (kq + 1 < LP_KQ) ? cbp + ((kq + 1) * LP_RT + f) * LP_FRAG_BYTES : nbp + f * LP_FRAG_BYTES;
                            abuf[(kq + 1) & 1][f] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(src);
                        }
                    }
                    acc[par][rt][blk] = lp_mfma<BF>(abuf[kq & 1][rt], in[2 * ks + blk], ks == 0 ? bias[rt] : acc[par][rt][blk]);
                    // refill of the slot freed by the barrier above: this wave's quarter of the chunk, four 1 KiB DMA pieces
                    if (kq == LP_KQ / 2 && m == 1) { glds_set_dst(rg.idst); glds_piece_m0<0>(rg.isrc, rg.lane_off, rg.idst); }
                    if (kq == LP_KQ / 2 && m == 2) glds_piece_m0<1>(rg.isrc, rg.lane_off, rg.idst);
                    if (kq == LP_KQ / 2 + 1 && m == 1) glds_piece_m0<2>(rg.isrc, rg.lane_off, rg.idst);
                    if (kq == LP_KQ / 2 + 1 && m == 2) { glds_piece_m0<3>(rg.isrc, rg.lane_off, rg.idst); lp_issue_next(rg); }
                    if (NCH == 1) {          // first layer: four steps in all, one row tile behind the last MFMA of each row tile of step 1
                        if (g > 0 && ks == 1 && blk == NBLK - 1) lp_epilogue<BF, LAST>(g - 1, rt, rt + 1, acc[par ^ 1], dp, gq, out, part, 0, NBLK);
                    } else if (epi && !LAST) {
                        const int T = 4 * (g - 1) + ert;
                        if (m == E0 + 0) { e0 = acc[par ^ 1][ert][eblk].x; e1 = acc[par ^ 1][ert][eblk].y; asm volatile("" : "+v"(e0), "+v"(e1)); }
                        if (m == E0 + 1) out[2 * (T >> 1) + eblk][2 * (T & 1) + 0] = lp_relu_pack<BF>(e0, e1);
                        if (m == E0 + 2) { e0 = acc[par ^ 1][ert][eblk].z; e1 = acc[par ^ 1][ert][eblk].w; asm volatile("" : "+v"(e0), "+v"(e1)); }
                        if (m == E0 + 3) out[2 * (T >> 1) + eblk][2 * (T & 1) + 1] = lp_relu_pack<BF>(e0, e1);
                    } else if (epi && m == E0) {      // last hidden layer (one pass in eight): the unit's dot-product terms in one piece
                        lp_epilogue<BF, LAST>(g - 1, ert, ert + 1, acc[par ^ 1], dp, gq, out, part, eblk, eblk + 1);
                    }
                    // the next group's bias: behind the second MFMA of the group's last step, six MFMAs ahead of the lgkmcnt(0) that follows
                    if (ks == LP_KQ * NCH - 1 && m == 1 && g + 1 < LP_NOG) lp_load_rows(bp, g + 1, gq, bias);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            rg.rd_slot = nx_slot;
        }
    }
    // the last group's epilogue has no MFMAs of its own pass to hide behind
    lp_epilogue<BF, LAST>(LP_NOG - 1, 0, LP_RT, acc[(LP_NOG - 1) & 1], dp, gq, out, part, 0, NBLK);
}

// Four waves, one per SIMD.  (Eight waves of one column block each -- two per SIMD, one wave's reads and epilogue in the slots the other's
// MFMAs leave -- were measured: duty 0.71 against 0.67, granted clock -150 MHz, slower; profiles/r05_k0_clock.md, r06_removed_experiments.md.)
template <bool BF, int NBLK>
__global__ __launch_bounds__(256, 1) void mlp_lp_kernel(const LpArgs a) {
    constexpr int NW = 4;
    constexpr int TILE = 16 * NBLK * NW, WAVE_PTS = 16 * NBLK;     // LP_TILE_PTS / LP_WAVE_PTS for NBLK = 2
    constexpr int NT = 64 * NW, WAVE_BYTES = CHUNK_BYTES / NW, PIECES = GLDS_PER_CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4;    // which 4-row block of each 16-row tile / which 8 of each step's 32 k slots this lane holds
    const int pl = lane & 15;    // this lane's point inside each of the wave's two 16-point column blocks

    float* bias_l = reinterpret_cast<float*>(smem);
    float* cb_l = reinterpret_cast<float*>(smem + BIAS_BYTES);
    float* zero_l = reinterpret_cast<float*>(smem + BIAS_BYTES + CODEBIAS_BYTES);     // a row of zeros: the "final-layer weights" of every layer but the last
    char* ring_ptr = smem + BIAS_BYTES + CODEBIAS_BYTES + LP_ZERO_BYTES;

    DirectList dl{0, 0, 0, 0};
    if (a.direct.kind) {
        dl = direct_list(a.direct, TILE);
        if (blockIdx.x == 0 && tid == 0) direct_commit(a.direct, dl);
    }
    const int n_tiles = a.direct.kind ? dl.n_tiles : *a.n_tiles;
    if ((int)blockIdx.x >= n_tiles) return;
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[0] = clock64(); a.clk[1] = wall_clock64(); }
    for (int i = tid; i < a.n_bias_rows * WIDTH; i += NT) bias_l[i] = a.bias_tab[i];
    for (int i = tid; i < WIDTH; i += NT) zero_l[i] = 0.f;
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    LpRing rg;
    rg.issue_pos = 0; rg.issue_slot = 0; rg.rd_slot = 0; rg.total_chunks = a.total_chunks;
    rg.wbase = reinterpret_cast<const char*>(a.wstream) + wave * WAVE_BYTES;   // wave-uniform; the lane part is rg.lane_off
    rg.lane_off = lane * 16;
    rg.isrc = rg.wbase;
    rg.ring0 = lds_addr(ring_ptr) + wave * WAVE_BYTES;
    rg.idst = rg.ring0;
    rg.ring_ptr = ring_ptr;
    rg.ring_lane = lds_addr(ring_ptr) + lane * 16;
#pragma unroll
    for (int i = 0; i < LP_NBUF - 1; ++i) {
        glds_quarter(rg.isrc, rg.lane_off, rg.idst);
        lp_issue_next(rg);
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PIECES * (LP_NBUF - 2)) : "memory");
    u32x4 abuf[2][LP_RT];        // A fragments of the current step and of the next one
#pragma unroll
    for (int rt = 0; rt < LP_RT; ++rt) {
        abuf[0][rt] = *reinterpret_cast<const u32x4*>(ring_ptr + lane * 16 + rt * LP_FRAG_BYTES);
        abuf[1][rt] = (u32x4){0u, 0u, 0u, 0u};
    }

    u32x4 X[32], Y[32];
    f32x4 acc[2][LP_RT][2];
#pragma unroll
    for (int i = 0; i < 32; ++i) { X[i] = (u32x4){0u, 0u, 0u, 0u}; Y[i] = (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rt = 0; rt < LP_RT; ++rt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) acc[i][rt][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wl = bias_l + a.wlast_row * WIDTH;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int4 td = a.direct.kind ? direct_tile(a.direct, dl, tile, TILE) : a.tiles[tile];
        // this lane's two points: one per 16-point column block of the wave
        bool valid[2];
        int pidx[2], src[2];
        float4 pt[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int local = wave * WAVE_PTS + 16 * blk + pl;
            valid[blk] = blk < NBLK && local < td.y;
            pidx[blk] = td.x + (valid[blk] ? local : 0);
            src[blk] = a.index ? a.index[pidx[blk]] : pidx[blk];
            pt[blk] = a.pts[src[blk]];
            if (!valid[blk]) pt[blk] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        reinterpret_cast<float4*>(cb_l)[tid] = reinterpret_cast<const float4*>(a.code_bias + (size_t)td.z * a.code_bias_stride)[tid];
        __syncthreads();

        // split-precision xyz operands (LP_XYZ_TERMS): k slot 16 u + 3 t + c of the xyz step carries part xpart(u, t) of coordinate c
        // (u = which of the table's two 16-slot halves); this lane holds slots 8 gq .. 8 gq + 7
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
        // slabs ping-pong: the first layer reads Y (its xyz step) and writes X, layer 1 reads X and writes Y, ...
        auto bias_of = [&](const LpPass& pd) { return pd.bias_row == -2 ? cb_l + WIDTH : (pd.bias_row == -3 ? cb_l : bias_l + pd.bias_row * WIDTH); };
        // Pass bodies: first layer (Y -> X), hidden layers X -> Y and Y -> X, and the LAST hidden layer, which reads X and writes no
        // slab (only the final layer's dot product on the un-rounded accumulators).  The pass count is even (pack_decoder_lp_host
        // refuses others: the prepass is then off), so the last layer's input is always in X and the loop has no conditional half --
        // a join there costs ~120 spilled registers per tile.
        const int n_mid = a.n_pass - 2;      // hidden layers between the first and the last one
        lp_pass<BF, 1, false, NBLK>(a.pass[0], Y, X, acc, abuf, rg, xb, bias_of(a.pass[0]), zero_l, lane, gq, part);
        for (int ps = 1; ps < n_mid; ps += 2) {       // n_mid is even (the host refuses odd pass counts): always both halves, no join
            lp_pass<BF, LP_NCH, false, NBLK>(a.pass[ps], X, Y, acc, abuf, rg, xb, bias_of(a.pass[ps]), zero_l, lane, gq, part);
            lp_pass<BF, LP_NCH, false, NBLK>(a.pass[ps + 1], Y, X, acc, abuf, rg, xb, bias_of(a.pass[ps + 1]), zero_l, lane, gq, part);
        }
        lp_pass<BF, LP_NCH, true, NBLK>(a.pass[a.n_pass - 1], X, Y, acc, abuf, rg, xb, bias_of(a.pass[a.n_pass - 1]), wl, lane, gq, part);
        // a point's 512 rows are spread over the four lane groups: lanes p, p + 16, p + 32, p + 48
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            part[blk] += __shfl_xor(part[blk], 16);
            part[blk] += __shfl_xor(part[blk], 32);
        }
        // lane group 0 stores the point of column block 0, lane group 1 that of column block 1
        const int sb = gq & 1;
        float y = tanhf((sb ? part[1] : part[0]) + a.b_last);
        // exactly 1.0f is the optimiser's "never decoded" placeholder (gn_kernels.hip: sample_write_ray): a prepass value never takes it.
        // (tanh saturates to 1.0f above ~9 -- or after an f16 overflow upstream.)  NaN stays NaN: the band kernels send it to the fp32 kernel.
        if (y >= 1.0f) y = 0x1.fffffep-1f;
        if (gq < NBLK && (sb ? valid[1] : valid[0])) a.out_sdf[a.index ? (sb ? src[1] : src[0]) : (sb ? pidx[1] : pidx[0]) + td.w] = y;
        // stores and LDS-DMA share vmcnt and may retire out of order: drain before counting again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = clock64(); a.clk[3] = wall_clock64(); }
}

template __global__ void mlp_lp_kernel<false, 2>(const LpArgs);
template __global__ void mlp_lp_kernel<true, 2>(const LpArgs);
template __global__ void mlp_lp_kernel<false, 1>(const LpArgs);
template __global__ void mlp_lp_kernel<true, 1>(const LpArgs);

static size_t mlp_lp_lds_bytes() { return BIAS_BYTES + CODEBIAS_BYTES + LP_ZERO_BYTES + LP_NBUF * CHUNK_BYTES; }

hipError_t mlp_lp_prepare_device() {
    const void* fns[4] = {reinterpret_cast<const void*>(&mlp_lp_kernel<false, 2>), reinterpret_cast<const void*>(&mlp_lp_kernel<true, 2>),
                          reinterpret_cast<const void*>(&mlp_lp_kernel<false, 1>), reinterpret_cast<const void*>(&mlp_lp_kernel<true, 1>)};
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_lp_lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// tile_pts: LP_TILE_PTS (128: two column blocks per wave) or LP_TILE_PTS_SMALL (64: one); the tile list must have been built for it
hipError_t launch_mlp_lp(bool bf16, const LpArgs& args, int n_blocks, hipStream_t stream, int tile_pts) {
    const bool small = tile_pts == LP_TILE_PTS_SMALL;
    const size_t lds = mlp_lp_lds_bytes();
    if (small) {
        if (bf16) hipLaunchKernelGGL((mlp_lp_kernel<true, 1>), dim3(n_blocks), dim3(256), lds, stream, args);
        else hipLaunchKernelGGL((mlp_lp_kernel<false, 1>), dim3(n_blocks), dim3(256), lds, stream, args);
    } else {
        if (bf16) hipLaunchKernelGGL((mlp_lp_kernel<true, 2>), dim3(n_blocks), dim3(256), lds, stream, args);
        else hipLaunchKernelGGL((mlp_lp_kernel<false, 2>), dim3(n_blocks), dim3(256), lds, stream, args);
    }
    return hipGetLastError();
}

}  // namespace dsp
