// DeepSDF decoder forward / forward+input-gradient for gfx950 (MI355X), fp32 MFMA.
//
// Replaces, for batches of object-frame points:
//   decode_sdf              reconstruct/loss_utils.py:51-79   (mlp_kernel<false>)
//   get_batch_sdf_jacobian  reconstruct/loss_utils.py:82-103  (mlp_kernel<true>)
//   Decoder.forward         deep_sdf/deep_sdf_decoder.py:75-110
//
// Design (DESIGN.md "K1/K2"):  one workgroup = 4 waves = one 64-point tile, one wave per SIMD with the
// whole 512-entry register file.  Each wave owns 16 points and keeps its [512 rows x 16 points]
// activation slab IN REGISTERS for the whole network: v_mfma_f32_16x16x4_f32 produces D[row][pt] with
// lane (pt = l&15, g = l>>4) holding rows 4g..4g+3 of each 16-row tile -- which is exactly the B-operand
// layout (k = l>>4) of the next layer if that layer's weights are packed with the matching k
// permutation.  So activations never touch LDS or HBM; only the weights stream through a 6-deep LDS
// ring filled by LDS-DMA (global_load_lds_dwordx4) with counted vmcnt waits and one s_barrier per
// 16 KiB chunk (64 MFMAs per wave).  ReLU masks for the backward sweep live in LDS (32 KiB).
#include "dsp_internal.h"
#include "mlp_common.h"

namespace dsp {

// relu-mask bit of one pre-activation, shifted into `bits` (bits = 2 * bits + (x > 0)): v_cmp + v_addc instead of the
// v_cmp / v_cndmask / v_lshl_or chain hipcc emits for `bits |= (x > 0) << k`.  Feed elements 15 down to 0 and element k ends at bit k.
__device__ __forceinline__ void push_mask_bit(unsigned& bits, float x) {
    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(x) : "vcc");
}

#define FOR16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// MODE 0: forward only.  MODE 1: forward, relu masks kept and exported (with nothing else to do) for every sample whose
// |sdf| < th -- the candidates of the render term.  MODE 2: forward + backward (input gradient).  MODE 3: backward only,
// from the masks and sdf a MODE 1 launch exported for the same point and code (the render rows of the jacobian: their
// forward pass is not repeated).
template <int MODE>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const MlpArgs a) {
    constexpr bool BWD = MODE >= 2;        // runs the backward sweep
    constexpr bool MASKS = MODE != 0;      // relu masks live in LDS
    constexpr bool DOFWD = MODE != 3;      // runs the forward sweep
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;    // MFMA k index of this lane group == 4-row block inside a 16-row tile
    const int pl = lane & 15;   // point of this lane inside the wave

    float* bias_l = reinterpret_cast<float*>(smem);
    float* cb_l = reinterpret_cast<float*>(smem + BIAS_BYTES);   // [0..511] layer-0 code bias, [512..1023] latent_in code bias
    unsigned short* mask_l = reinterpret_cast<unsigned short*>(smem + BIAS_BYTES + CODEBIAS_BYTES);
    char* ring_ptr = smem + BIAS_BYTES + CODEBIAS_BYTES + (MASKS ? MASK_BYTES : 0);
    const unsigned ring0 = lds_addr(ring_ptr);

    const int n_tiles = *a.n_tiles;
    const int tile0 = a.tile_begin ? *a.tile_begin : 0;
    if (tile0 + (int)blockIdx.x >= n_tiles) return;
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[0] = clock64(); a.clk[1] = wall_clock64(); }
    for (int i = tid; i < a.n_bias_rows * WIDTH; i += 256) bias_l[i] = a.bias_tab[i];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- weight stream state (all wave-uniform) ---------------------------------------------------
    int issue_pos = 0, issue_slot = 0, rd_slot = 0;
    const char* wbase = reinterpret_cast<const char*>(a.wstream) + wave * 4096;   // wave-uniform; the lane part is lane_off
    const unsigned lane_off = lane * 16;
    const int total_chunks = a.total_chunks;
    // A chunk refill is four LDS-DMA instructions per wave.  Each costs ~30-60 issue cycles (64 lane addresses through
    // the address unit), so in the main loop they are spread over four k-steps, one behind an MFMA each, instead of
    // stalling the matrix pipe for a whole burst (measured: the burst form cost 5 % of the kernel).
    const char* isrc = wbase;
    unsigned idst = ring0 + wave * 4096;
    auto issue_next = [&]() {   // advance to the next chunk of the stream / next ring slot
        issue_pos = (issue_pos + 1 == total_chunks) ? 0 : issue_pos + 1;
        issue_slot = (issue_slot + 1 == NBUF) ? 0 : issue_slot + 1;
        isrc = wbase + (size_t)issue_pos * CHUNK_BYTES;
        idst = ring0 + issue_slot * CHUNK_BYTES + wave * 4096;
    };
    auto issue = [&]() {
        glds_quarter(isrc, lane_off, idst);
        issue_next();
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue();
    // chunk 0 visible to every wave; from here on the barrier for chunk q+1 sits in the middle of chunk q
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GLDS_PER_CHUNK * (NBUF - 2)) : "memory");
    f32x4 abuf[2 * PAIR_READS];   // A operands run one group of PAIR_READS k-steps ahead of the MFMAs, across chunk / group / pass seams
    static_assert(PAIR_READS == 2, "the prologue below reads the first group");
    abuf[0] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16);
    abuf[1] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16 + 1024);

    float sin_[128];   // input slab of the current pass:  sin_[4t+r] = row 16t + 4g + r of point pl
    f32x4 acc[32];     // output slab being produced: the MFMA accumulators of all 32 row tiles
#pragma unroll
    for (int i = 0; i < 128; ++i) sin_[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = tile0 + blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int4 td = a.tiles[tile];
        const int local = wave * WAVE_PTS + pl;
        const bool valid = local < td.y;
        const int pidx = td.x + (valid ? local : 0);
        const int src = (MODE <= 1 && a.index) ? a.index[pidx] : pidx;
        float4 pt = a.pts[src];
        // prepass guard (mlp_common.h): the value this tile's result will replace is fetched NOW, with the point, so that its latency
        // hides behind the tile instead of stalling its last instructions (one register per lane for the tile's lifetime)
        float old_lp = 1.0f, guard_tol = 0.f;
        if (MODE <= 1 && a.guard) {
            if (valid && g == 0) old_lp = a.out_sdf[a.index ? src : pidx + td.w];
            guard_tol = 0.5f * __uint_as_float(a.guard[(size_t)td.z * a.guard_stride]);     // tile-uniform: the object's margin / 2
        }
        if (!valid) pt = make_float4(0.f, 0.f, 0.f, 0.f);
        // The shape code is the same for every point of the tile, so its contribution to layer 0 and to the latent_in
        // layer is a per-object bias vector (k_code_bias): stage both into LDS.  What is left of layer 0 is three
        // multiply-adds per row (W0[:, xyz] . p) -- done on the VALU below instead of 16 mostly-empty MFMA chunks.
        if (DOFWD) {
        reinterpret_cast<float4*>(cb_l)[tid] = reinterpret_cast<const float4*>(a.code_bias + (size_t)td.z * a.code_bias_stride)[tid];
        __syncthreads();
        {
            const float* w0 = bias_l + a.w0_row * WIDTH + 4 * g;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float pre[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * (4 * o + j);
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb_l + row + 4 * g);
                    const f32x4 wx = *reinterpret_cast<const f32x4*>(w0 + row);
                    const f32x4 wy = *reinterpret_cast<const f32x4*>(w0 + WIDTH + row);
                    const f32x4 wz = *reinterpret_cast<const f32x4*>(w0 + 2 * WIDTH + row);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[4 * j + r] = fmaf(wz[r], pt.z, fmaf(wy[r], pt.y, fmaf(wx[r], pt.x, c0[r])));
                }
                if (MASKS) {
                    unsigned bits = 0;
#pragma unroll
                    for (int k = 15; k >= 0; --k) push_mask_bit(bits, pre[k]);
                    mask_l[(0 * 8 + o) * 256 + tid] = (unsigned short)bits;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) sin_[16 * o + k] = relu1(pre[k]);
            }
        }
        }
        float skipc[16];
        float skipx[3];
        float y = 0.f;
        float gfirst = 0.f;   // d y / d (x|y|z by lane group) through the first layer
#pragma unroll
        for (int i = 0; i < 16; ++i) skipc[i] = 0.f;
        skipx[0] = skipx[1] = skipx[2] = 0.f;

        // seed of the backward sweep: d tanh * W_last, masked by the last hidden layer's relu
        auto seed_backward = [&](int slot) {
            const float* wl = bias_l + a.wlast_row * WIDTH + 4 * g;
            const float d = 1.f - y * y;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const unsigned bits = mask_l[(slot * 8 + o) * 256 + tid];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + 16 * (4 * o + j));
                    sin_[16 * o + 4 * j + 0] = ((bits >> (4 * j + 0)) & 1u) ? d * w4.x : 0.f;
                    sin_[16 * o + 4 * j + 1] = ((bits >> (4 * j + 1)) & 1u) ? d * w4.y : 0.f;
                    sin_[16 * o + 4 * j + 2] = ((bits >> (4 * j + 2)) & 1u) ? d * w4.z : 0.f;
                    sin_[16 * o + 4 * j + 3] = ((bits >> (4 * j + 3)) & 1u) ? d * w4.w : 0.f;
                }
            }
        };
        if (MODE == 3) {
            // masks and sdf of this point as the forward (MODE 1) launch of the same iteration exported them; each lane
            // only ever touches its own 64 mask words, so no barrier is needed
            const int sidx = __float_as_int(pt.w);
            const uint4* mp = reinterpret_cast<const uint4*>(a.mask_buf + ((size_t)sidx * 4 + g) * 64);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint4 w = valid ? mp[q] : make_uint4(0, 0, 0, 0);
                const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mask_l[(8 * q + 2 * e + 0) * 256 + tid] = (unsigned short)(ww[e] & 0xffffu);
                    mask_l[(8 * q + 2 * e + 1) * 256 + tid] = (unsigned short)(ww[e] >> 16);
                }
            }
            y = valid ? a.sdf_in[sidx] : 0.f;
            seed_backward(a.seed_slot);
        }

        for (int ps = 0; ps < a.n_pass; ++ps) {
            const PassDesc pd = a.pass[ps];
            // ---- pass prologue ------------------------------------------------------------------------------------
            if (pd.kind == 2) {
                // latent_in layer: xyz re-enters at rows 445..447 (lane group 3); the code part is in the bias (cb_l + 512)
                // (row 445 = tile 27, row 477 = tile 29 with 32-D codes; both are row 13 of their tile: lane group 3, registers 1..3)
                if (g == 3) {
                    if (a.lat_tile == 29) { sin_[117] = pt.x; sin_[118] = pt.y; sin_[119] = pt.z; }
                    else { sin_[109] = pt.x; sin_[110] = pt.y; sin_[111] = pt.z; }
                }
            } else if (BWD && pd.kind == 5) {
                // first layer, backward: d y / d xyz = W0[:, xyz]^T ga0 on the VALU (ga0 = the input slab of this pass,
                // which the pass epilogue overwrites); the MFMA pass itself only produces the 64 code rows
                const float* w0 = bias_l + a.w0_row * WIDTH + 4 * g;
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                for (int t = 0; t < 32; ++t) {
                    const f32x4 wx = *reinterpret_cast<const f32x4*>(w0 + 16 * t);
                    const f32x4 wy = *reinterpret_cast<const f32x4*>(w0 + WIDTH + 16 * t);
                    const f32x4 wz = *reinterpret_cast<const f32x4*>(w0 + 2 * WIDTH + 16 * t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gx = fmaf(wx[r], sin_[4 * t + r], gx);
                        gy = fmaf(wy[r], sin_[4 * t + r], gy);
                        gz = fmaf(wz[r], sin_[4 * t + r], gz);
                    }
                }
                gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
                gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
                gz += __shfl_xor(gz, 16); gz += __shfl_xor(gz, 32);
                gfirst = (g == 0) ? gx : (g == 1) ? gy : gz;
            }

            // All 32 output tiles of the layer are MFMA accumulators (acc[4*og+j], 128 AGPRs) with static indices: the
            // group / chunk / k-step loops are fully unrolled and skipped by uniform branches, so no register is ever
            // indexed dynamically and the MFMA stream runs from one output group into the next without an epilogue.
#pragma unroll
            for (int og = 0; og < 8; ++og) {
                if (og < pd.nog) {
                    f32x4 bias4[4];
                    if (pd.bias_row != -1) {
                        const float* bp = (pd.bias_row == -2 ? cb_l + WIDTH : bias_l + pd.bias_row * WIDTH) + 64 * og + 4 * g;
#pragma unroll
                        for (int j = 0; j < 4; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(bp + 16 * j);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (c < pd.nchunks) {
                            const int nx_slot = (rd_slot + 1 == NBUF) ? 0 : rd_slot + 1;
                            const char* cb = ring_ptr + rd_slot * CHUNK_BYTES + lane * 16;
                            const char* nb = ring_ptr + nx_slot * CHUNK_BYTES + lane * 16;
#pragma unroll
                            for (int s = 0; s < KSTEPS_PER_CHUNK; ++s) {
                                if (s == KSTEPS_PER_CHUNK / 2) {
                                    // Chunk q+1 has landed for this wave once <= NBUF-3 younger chunks are in flight; the
                                    // barrier publishes every wave's quarter and proves all reads of chunk q-1 retired
                                    // (every wave is inside chunk q), so its slot can be refilled right away.
                                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GLDS_PER_CHUNK * (NBUF - 3)) : "memory");
                                }
                                // A operands in groups of RG k-steps: one wait (lgkmcnt(0): the group read RG k-steps ago) and RG reads
                                // (k-steps s+RG .. s+2RG-1) every RG-th k-step -- 1 + 1/RG non-MFMA instructions per k-step instead of 2
                                constexpr int RG = PAIR_READS;
                                if ((s % RG) == 0) {
                                    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0), vmcnt / expcnt untouched
                                    __builtin_amdgcn_sched_barrier(0);      // keep it ahead of the k-step's first MFMA
#pragma unroll
                                    for (int q = 0; q < RG; ++q) {
                                        const int sp = s + RG + q;
                                        abuf[sp % (2 * RG)] = (sp < KSTEPS_PER_CHUNK)
                                                           ? *reinterpret_cast<const f32x4*>(cb + sp * 1024)
                                                           : *reinterpret_cast<const f32x4*>(nb + (sp - KSTEPS_PER_CHUNK) * 1024);
                                    }
                                }
                                const f32x4 av = abuf[s % (2 * PAIR_READS)];
                                const float b = sin_[16 * c + s];
                                acc[4 * og + 0] = MFMA16(av.x, b, (c == 0 && s == 0) ? bias4[0] : acc[4 * og + 0]);
                                // refill of the slot freed by the barrier above: one DMA piece per k-step, behind an MFMA
                                if (s == KSTEPS_PER_CHUNK / 2 + 0) { glds_set_dst(idst); glds_piece_m0<0>(isrc, lane_off, idst); };
                                if (s == KSTEPS_PER_CHUNK / 2 + 1) glds_piece_m0<1>(isrc, lane_off, idst);
                                if (s == KSTEPS_PER_CHUNK / 2 + 2) glds_piece_m0<2>(isrc, lane_off, idst);
                                if (s == KSTEPS_PER_CHUNK / 2 + 3) { glds_piece_m0<3>(isrc, lane_off, idst); issue_next(); }
                                acc[4 * og + 1] = MFMA16(av.y, b, (c == 0 && s == 0) ? bias4[1] : acc[4 * og + 1]);
                                acc[4 * og + 2] = MFMA16(av.z, b, (c == 0 && s == 0) ? bias4[2] : acc[4 * og + 2]);
                                acc[4 * og + 3] = MFMA16(av.w, b, (c == 0 && s == 0) ? bias4[3] : acc[4 * og + 3]);
                                // pin "read for k-step s+2, then the four MFMAs of k-step s": left alone, hipcc sinks
                                // every ds_read next to its use (one A buffer, lgkmcnt(0) before each MFMA quad)
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            rd_slot = nx_slot;
                        }
                    }
                }
            }

            // ---- layer epilogue: relu (+ mask store) or mask apply, accumulators -> next layer's input slab ----------
#pragma unroll
            for (int og = 0; og < 8; ++og) {
                if (og < pd.nog) {
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[4 * j + 0] = acc[4 * og + j].x; v[4 * j + 1] = acc[4 * og + j].y;
                        v[4 * j + 2] = acc[4 * og + j].z; v[4 * j + 3] = acc[4 * og + j].w;
                    }
                    if (pd.relu) {
                        if (MASKS) {
                            unsigned bits = 0;
#pragma unroll
                            for (int k = 15; k >= 0; --k) push_mask_bit(bits, v[k]);
                            mask_l[(pd.mask_slot * 8 + og) * 256 + tid] = (unsigned short)bits;
                        }
#pragma unroll
                        for (int k = 0; k < 16; ++k) v[k] = relu1(v[k]);
                    } else if (BWD && pd.mask_slot >= 0) {
                        if (pd.kind == 4) {
                            // latent_in layer: rows 445..447 / 448..511 of its input are the re-injected xyz / code, not
                            // relu outputs -- keep their gradients (unmasked) for the final d/d[code,xyz] sum
                            if (a.lat_tile != 29) {          // 64-D codes: xyz at rows 445..447, code at 448..511
                                if (og == 6) { skipx[0] = v[13]; skipx[1] = v[14]; skipx[2] = v[15]; }
                                if (og == 7) {
#pragma unroll
                                    for (int k = 0; k < 16; ++k) skipc[k] = v[k];
                                }
                            } else if (og == 7) {            // 32-D codes: xyz at rows 477..479, code at 480..511 (code index 16 (j - 2) + 4 g + r)
                                skipx[0] = v[5]; skipx[1] = v[6]; skipx[2] = v[7];
#pragma unroll
                                for (int k = 0; k < 8; ++k) skipc[k] = v[8 + k];
                            }
                        }
                        const unsigned bits = mask_l[(pd.mask_slot * 8 + og) * 256 + tid];
#pragma unroll
                        for (int k = 0; k < 16; ++k) v[k] = ((bits >> k) & 1u) ? v[k] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) sin_[16 * og + k] = v[k];
                }
            }

            if (DOFWD && ps == a.n_fwd - 1) {
                // final layer (512 -> 1) on the VALU + tanh  (deep_sdf_decoder.py:93,107-108)
                const float* wl = bias_l + a.wlast_row * WIDTH + 4 * g;
                float part = 0.f;
#pragma unroll
                for (int t = 0; t < 32; ++t) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + 16 * t);
                    part = fmaf(sin_[4 * t + 0], w4.x, part);
                    part = fmaf(sin_[4 * t + 1], w4.y, part);
                    part = fmaf(sin_[4 * t + 2], w4.z, part);
                    part = fmaf(sin_[4 * t + 3], w4.w, part);
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                y = tanhf(part + a.b_last);
                if (!BWD) {
                    if (a.guard) prepass_guard_tol(a, td.z, valid && g == 0, old_lp, y, guard_tol);
                    if (valid && g == 0) a.out_sdf[a.index ? src : pidx + td.w] = y;
                    if (MODE == 1 && valid && y > -a.th && y < a.th) {
                        // a candidate row of the render term (loss.py:88): export this lane's 64 mask words (8 layers x 8
                        // output groups) so that the jacobian launch can run the backward sweep without a second forward
                        uint4* mp = reinterpret_cast<uint4*>(a.mask_buf + ((size_t)src * 4 + g) * 64);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            unsigned ww[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                ww[e] = (unsigned)mask_l[(8 * q + 2 * e + 0) * 256 + tid] | ((unsigned)mask_l[(8 * q + 2 * e + 1) * 256 + tid] << 16);
                            mp[q] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
                        }
                    }
                } else {
                    seed_backward(pd.mask_slot);
                }
            }
        }

        if (BWD) {
            // sin_[0..15] now holds d y / d code through the first layer (rows 16t + 4g + r); gfirst the xyz part
            float* orow = a.out_grad + (size_t)(pidx + td.w) * GRAD_STRIDE;
            if (valid) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float4 o4;
                    o4.x = sin_[4 * t + 0] + skipc[4 * t + 0];
                    o4.y = sin_[4 * t + 1] + skipc[4 * t + 1];
                    o4.z = sin_[4 * t + 2] + skipc[4 * t + 2];
                    o4.w = sin_[4 * t + 3] + skipc[4 * t + 3];
                    *reinterpret_cast<float4*>(orow + 16 * t + 4 * g) = o4;
                }
            }
            const float s0 = __shfl(skipx[0], pl + 48);
            const float s1 = __shfl(skipx[1], pl + 48);
            const float s2 = __shfl(skipx[2], pl + 48);
            const float sk = (g == 0) ? s0 : (g == 1) ? s1 : s2;
            if (valid) orow[64 + g] = (g < 3) ? (gfirst + sk) : y;
            if (MODE == 2 && a.sdf_scatter && tile >= *a.scatter_tile_begin) {
                const bool sc = valid && g == 3;
                if (a.guard) prepass_guard(a, td.z, sc, sc ? a.sdf_scatter[__float_as_int(pt.w)] : 1.0f, y);
                if (sc) a.sdf_scatter[__float_as_int(pt.w)] = y;
            }
        }
        // stores and LDS-DMA share vmcnt and may retire out of order: drain before counting again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = clock64(); a.clk[3] = wall_clock64(); }
}

template __global__ void mlp_kernel<0>(const MlpArgs);
template __global__ void mlp_kernel<1>(const MlpArgs);
template __global__ void mlp_kernel<2>(const MlpArgs);
template __global__ void mlp_kernel<3>(const MlpArgs);

size_t mlp_lds_bytes(int mode) { return BIAS_BYTES + CODEBIAS_BYTES + (mode != 0 ? MASK_BYTES : 0) + NBUF * CHUNK_BYTES; }

// Opt every kernel variant into > 64 KiB of dynamic LDS on the current device (called by dsp_create).
hipError_t mlp_prepare_device() {
    const void* fns[4] = {reinterpret_cast<const void*>(&mlp_kernel<0>), reinterpret_cast<const void*>(&mlp_kernel<1>),
                          reinterpret_cast<const void*>(&mlp_kernel<2>), reinterpret_cast<const void*>(&mlp_kernel<3>)};
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_lds_bytes(i));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_mlp(int mode, const MlpArgs& args, int n_blocks, hipStream_t stream) {
    if (mode == 0)
        hipLaunchKernelGGL((mlp_kernel<0>), dim3(n_blocks), dim3(256), mlp_lds_bytes(0), stream, args);
    else if (mode == 1)
        hipLaunchKernelGGL((mlp_kernel<1>), dim3(n_blocks), dim3(256), mlp_lds_bytes(1), stream, args);
    else if (mode == 2)
        hipLaunchKernelGGL((mlp_kernel<2>), dim3(n_blocks), dim3(256), mlp_lds_bytes(2), stream, args);
    else
        hipLaunchKernelGGL((mlp_kernel<3>), dim3(n_blocks), dim3(256), mlp_lds_bytes(3), stream, args);
    return hipGetLastError();
}

}  // namespace dsp
