// DeepSDF decoder forward + input gradient, LATENCY form: one workgroup = 4 waves = ONE 16-point tile.
//
// mlp_kernel.hip gives every wave its own 16 points and all 512 output rows of each layer: a 64-point tile is a serial
// chain of 15 layer passes x 4096 MFMAs per wave (~0.9 ms forward + backward), whatever the number of tiles.  With one
// object in flight (SLAM's per-detection calls: estimate_pose_cam_obj, a single reconstruct_object) the jacobian launch
// has 30-60 such tiles for 256 CUs, so that chain IS the latency.  Here the four waves share the same 16 points and split
// the OUTPUT ROWS of every layer instead: wave w produces output groups 2w and 2w+1 (128 rows = 8 MFMA tiles, 1024 MFMAs
// per pass), and the layer's [512 x 16] result is exchanged through LDS (32 KiB, two barriers per layer) so that every
// wave again holds the full input slab of the next layer in registers.  Same k order, same bias seeding, same masks per
// output element as mlp_kernel<2>, hence bit-identical gradients (tests/test_gpu_configs.py::test_split_kernel_is_exact).
//
// Weights: waves no longer share A operands, so each wave streams its own rows through a private LDS ring (5 mini-chunks
// of 4 k-steps = 4 KiB; one LDS-DMA piece per k-step, counted vmcnt, no barrier).  The host lays a second copy of the
// weight stream out per wave (`wsplit`: for each pass, the chunks of the wave's two output groups in consumption order),
// so the stream position is a single wrapping pointer per wave.
//
// Replaces get_batch_sdf_jacobian (reconstruct/loss_utils.py:82-103) and, as mlp_split_kernel<0 / 1>, decode_sdf
// (loss_utils.py:51-79) for latency-sized batches.
//
// MODE 0: forward only.  MODE 1: forward, and the relu masks of samples with |sdf| < th are exported (mlp_kernel<1>'s layout) -- the tail
// tiles / small lists of a mask-exporting forward launch.  MODE 2: forward + backward; and, from tile *bwd_only_tile_begin on, BACKWARD
// ONLY from the masks and sdf a MODE 1 launch exported (round 4, "mixed" launch): the kept render rows skip their second forward sweep
// in the SAME launch as the surface points' forward + backward tiles, so the two kinds share the rounds over the CUs.  A wave's weight
// stream is [forward passes | backward passes]; a backward-only tile consumes the second part only, so where the stream wraps to is
// decided by the kind of the workgroup's NEXT tile (known at tile start; the prefetch wraps four mini-chunks before the tile ends).
#include "dsp_internal.h"
#include "mlp_common.h"

namespace dsp {

constexpr int SNB = 5;                                      // per-wave ring depth in mini-chunks
constexpr int MINI_BYTES = 4096;                            // 4 k-steps x 64 lanes x 16 B
constexpr int SPLIT_MASK_BYTES = MASK_SLOTS * 2 * 256 * 2;  // [slot][own group 0/1][tid] u16 = 8 KiB
constexpr int XCH_BYTES = 32 * 64 * 16;                     // one layer's output slab: 32 row tiles x 64 lanes x float4
constexpr int SPLIT_RING_BYTES = 4 * SNB * MINI_BYTES;      // 80 KiB

// MODE < 2 consumes the forward prefix of each wave's stream.
template <int MODE>
__global__ __launch_bounds__(256, 1) void mlp_split_kernel(const MlpArgs a) {
    constexpr bool BWD = MODE == 2;
    constexpr bool MASKS = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pl = lane & 15;

    float* bias_l = reinterpret_cast<float*>(smem);
    float* cb_l = reinterpret_cast<float*>(smem + BIAS_BYTES);
    unsigned short* mask_l = reinterpret_cast<unsigned short*>(smem + BIAS_BYTES + CODEBIAS_BYTES);
    f32x4* xch = reinterpret_cast<f32x4*>(smem + BIAS_BYTES + CODEBIAS_BYTES + SPLIT_MASK_BYTES);
    char* ring_ptr = smem + BIAS_BYTES + CODEBIAS_BYTES + SPLIT_MASK_BYTES + XCH_BYTES + wave * (SNB * MINI_BYTES);
    const unsigned ring0 = lds_addr(ring_ptr);

    DirectList dl{0, 0, 0, 0};
    if (a.direct.kind) dl = direct_list(a.direct, SPLIT_TILE_PTS);
    const int n_tiles = a.direct.kind ? dl.n_tiles : *a.n_tiles;
    // a list this short is the cluster kernel's (mlp_cluster_kernel.hip): both are launched -- unless that kernel lost a hand-off in this
    // run (error word set: it returns at entry from then on, and what it left of the launch that failed is recomputed here)
    const unsigned cl_err = (BWD && a.cl_err) ? *reinterpret_cast<const volatile unsigned*>(a.cl_err) : 0u;
    if (BWD && n_tiles < a.split_min_tiles && cl_err == 0u) return;
    // (this kernel takes the list; if the cluster kernel of THIS launch gave up on it, that kernel has recorded the list's counts already)
    const bool counted = BWD && n_tiles < a.split_min_tiles && cl_err == (a.cl_epoch_base | 1u);
    if (a.direct.kind && blockIdx.x == 0 && tid == 0 && !counted) direct_commit(a.direct, dl);
    if ((int)blockIdx.x >= n_tiles) return;
    for (int i = tid; i < a.n_bias_rows * WIDTH; i += 256) bias_l[i] = a.bias_tab[i];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- this wave's weight stream (wave-uniform state) -----------------------------------------------------------------
    const int total_minis = (BWD ? a.split_len[wave] : a.split_len_fwd[wave]) * (CHUNK_BYTES / MINI_BYTES);
    const int fwd_minis = a.split_len_fwd[wave] * (CHUNK_BYTES / MINI_BYTES);
    const char* wbase = reinterpret_cast<const char*>(a.wsplit) + (size_t)a.split_off[wave] * CHUNK_BYTES;   // wave-uniform
    const unsigned lane_off = lane * 16;
    // mixed launch: tiles from bo_begin on run the backward sweep only; the stream then starts (and wraps to) the backward part
    const int bo_begin = (BWD && a.bwd_only_tile_begin) ? *a.bwd_only_tile_begin : 0x7fffffff;
    int wrap_to = (BWD && (int)blockIdx.x >= bo_begin) ? fwd_minis : 0;       // where the stream (re)starts: the kind of the NEXT tile to be fed
    int issue_pos = wrap_to, issue_slot = 0, rd_slot = 0;
    const char* isrc = wbase + (size_t)issue_pos * MINI_BYTES;
    unsigned idst = ring0;
    auto issue_next = [&]() {
        issue_pos = (issue_pos + 1 == total_minis) ? wrap_to : issue_pos + 1;
        issue_slot = (issue_slot + 1 == SNB) ? 0 : issue_slot + 1;
        isrc = wbase + (size_t)issue_pos * MINI_BYTES;
        idst = ring0 + issue_slot * MINI_BYTES;
    };
#pragma unroll
    for (int i = 0; i < SNB - 1; ++i) { glds_quarter(isrc, lane_off, idst); issue_next(); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SNB - 2)) : "memory");     // mini-chunk 0 has landed
    f32x4 abuf[4];
    abuf[0] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16);
    abuf[1] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16 + 1024);

    float sin_[128];   // full input slab of the current pass (every wave holds all 512 rows of the 16 points)
    f32x4 acc[8];      // this wave's two output groups
#pragma unroll
    for (int i = 0; i < 128; ++i) sin_[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. the weight prefetch in flight
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // own rows -> LDS, all rows <- LDS.  `nog` output groups of the layer are live; the others keep their old registers.
    auto exchange = [&](const float (&own)[2][16], int nog) {
        lds_barrier();                                     // every wave is done reading the previous exchange
#pragma unroll
        for (int ol = 0; ol < 2; ++ol) {
            const int og = 2 * wave + ol;
            if (og < nog) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xch[(4 * og + j) * 64 + lane] = (f32x4){own[ol][4 * j + 0], own[ol][4 * j + 1], own[ol][4 * j + 2], own[ol][4 * j + 3]};
            }
        }
        lds_barrier();
#pragma unroll
        for (int og = 0; og < 8; ++og) {
            if (og < nog) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = xch[(4 * og + j) * 64 + lane];
                    sin_[16 * og + 4 * j + 0] = v.x; sin_[16 * og + 4 * j + 1] = v.y;
                    sin_[16 * og + 4 * j + 2] = v.z; sin_[16 * og + 4 * j + 3] = v.w;
                }
            }
        }
    };

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int4 td = a.direct.kind ? direct_tile(a.direct, dl, tile, SPLIT_TILE_PTS) : a.tiles[tile];
        const bool valid = pl < td.y;
        const int pidx = td.x + (valid ? pl : 0);
        const int src = (!BWD && a.index) ? a.index[pidx] : pidx;
        float4 pt = a.pts[src];
        if (!valid) pt = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool bo = BWD && tile >= bo_begin;                                   // backward sweep only, from exported masks (workgroup-uniform)
        if (BWD) wrap_to = ((int)(tile + gridDim.x) >= bo_begin) ? fwd_minis : 0;  // the prefetch wraps during this tile: to where the NEXT tile starts
        lds_barrier();        // previous tile's readers of cb_l are done
        reinterpret_cast<float4*>(cb_l)[tid] = reinterpret_cast<const float4*>(a.code_bias + (size_t)td.z * a.code_bias_stride)[tid];
        lds_barrier();
        float y = 0.f;
        if (bo) {
            // masks and sdf of this point as the forward (MODE 1 / mlp_kernel<1>) launch of the same iteration exported them: this lane's words
            // of the wave's two output groups for the eight mask slots.  Layout [sample][lane group][slot 8][group 8] u16; as u32 words,
            // word 4 slot + w holds groups 2w (low half) and 2w + 1.
            const int sidx = __float_as_int(pt.w);
            const unsigned* mp = reinterpret_cast<const unsigned*>(a.mask_buf + ((size_t)sidx * 4 + g) * 64) + wave;
#pragma unroll
            for (int sl = 0; sl < MASK_SLOTS; ++sl) {
                const unsigned w = valid ? mp[4 * sl] : 0u;
                mask_l[(sl * 2 + 0) * 256 + tid] = (unsigned short)(w & 0xffffu);
                mask_l[(sl * 2 + 1) * 256 + tid] = (unsigned short)(w >> 16);
            }
            y = valid ? a.sdf_in[sidx] : 0.f;
        } else {   // layer 0 on the VALU, redundantly in every wave (the slab is needed everywhere); masks of the own groups only
            const float* w0 = bias_l + a.w0_row * WIDTH + 4 * g;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                unsigned bits = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 16 * (4 * o + j);
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb_l + row + 4 * g);
                    const f32x4 wx = *reinterpret_cast<const f32x4*>(w0 + row);
                    const f32x4 wy = *reinterpret_cast<const f32x4*>(w0 + WIDTH + row);
                    const f32x4 wz = *reinterpret_cast<const f32x4*>(w0 + 2 * WIDTH + row);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = fmaf(wz[r], pt.z, fmaf(wy[r], pt.y, fmaf(wx[r], pt.x, c0[r])));
                        bits |= (pre > 0.f ? 1u : 0u) << (4 * j + r);
                        sin_[16 * o + 4 * j + r] = relu1(pre);
                    }
                }
                if (MASKS && (o >> 1) == wave) mask_l[(0 * 2 + (o & 1)) * 256 + tid] = (unsigned short)bits;
            }
        }
        float skipc[16];
        float skipx[3];
        float gfirst = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) skipc[i] = 0.f;
        skipx[0] = skipx[1] = skipx[2] = 0.f;

        // seed of the backward sweep on the own rows: d tanh * W_last, masked by the last hidden layer's relu (mask slot `slot`); then round the workgroup
        auto seed_backward = [&](int slot) {
            const float* wl = bias_l + a.wlast_row * WIDTH + 4 * g;
            const float d = 1.f - y * y;
            float seed[2][16];
#pragma unroll
            for (int ol = 0; ol < 2; ++ol) {
                const int og = 2 * wave + ol;
                const unsigned bits = mask_l[(slot * 2 + ol) * 256 + tid];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + 16 * (4 * og + j));
                    seed[ol][4 * j + 0] = ((bits >> (4 * j + 0)) & 1u) ? d * w4.x : 0.f;
                    seed[ol][4 * j + 1] = ((bits >> (4 * j + 1)) & 1u) ? d * w4.y : 0.f;
                    seed[ol][4 * j + 2] = ((bits >> (4 * j + 2)) & 1u) ? d * w4.z : 0.f;
                    seed[ol][4 * j + 3] = ((bits >> (4 * j + 3)) & 1u) ? d * w4.w : 0.f;
                }
            }
            exchange(seed, 8);
        };
        if (bo) seed_backward(a.seed_slot);       // (the mask words written above are this lane's own: no barrier needed before reading them back)

        for (int ps = bo ? a.n_fwd : 0; ps < a.n_pass; ++ps) {
            const PassDesc pd = a.pass[ps];
            if (pd.kind == 2) {
                // (row 445 = tile 27, row 477 = tile 29 with 32-D codes; both are row 13 of their tile: lane group 3, registers 1..3)
                if (g == 3) {
                    if (a.lat_tile == 29) { sin_[117] = pt.x; sin_[118] = pt.y; sin_[119] = pt.z; }
                    else { sin_[109] = pt.x; sin_[110] = pt.y; sin_[111] = pt.z; }
                }
            } else if (BWD && pd.kind == 5) {
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

            // ---- this wave's two output groups: 2 x nchunks x 16 k-steps, statically indexed accumulators -------------------
#pragma unroll
            for (int ol = 0; ol < 2; ++ol) {
                const int og = 2 * wave + ol;
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
#pragma unroll
                            for (int s = 0; s < KSTEPS_PER_CHUNK; ++s) {
                                const int m = s & 3;       // k-step inside the mini-chunk
                                const int nx_slot = (rd_slot + 1 == SNB) ? 0 : rd_slot + 1;
                                const char* cbp = ring_ptr + rd_slot * MINI_BYTES + lane * 16;
                                const char* nbp = ring_ptr + nx_slot * MINI_BYTES + lane * 16;
                                if (m == 2) {
                                    // the next mini-chunk is about to be read: it has landed once only the two mini-chunks
                                    // after it and the two pieces issued so far in this one are still in flight
                                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SNB - 3) + 2) : "memory");
                                }
                                static_assert(PAIR_READS == 2, "pairs");
                                // A operands in pairs, one lgkmcnt wait per pair (mlp_common.h): at m == 2 the pair belongs to the next
                                // mini-chunk, which the vmcnt wait above has just proven landed
                                if ((s & 1) == 0) {
                                    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0), vmcnt / expcnt untouched
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int q = 0; q < 2; ++q) {
                                        const int sp = m + 2 + q;
                                        abuf[(s + 2 + q) % 4] = (sp < 4) ? *reinterpret_cast<const f32x4*>(cbp + sp * 1024)
                                                                         : *reinterpret_cast<const f32x4*>(nbp + (sp - 4) * 1024);
                                    }
                                }
                                const f32x4 av = abuf[s % 4];
                                const float b = sin_[16 * c + s];
                                acc[4 * ol + 0] = MFMA16(av.x, b, (c == 0 && s == 0) ? bias4[0] : acc[4 * ol + 0]);
                                // refill of the slot behind the read pointer: one DMA piece per k-step, behind an MFMA
                                if (m == 0) { glds_set_dst(idst); glds_piece_m0<0>(isrc, lane_off, idst); };
                                if (m == 1) glds_piece_m0<1>(isrc, lane_off, idst);
                                if (m == 2) glds_piece_m0<2>(isrc, lane_off, idst);
                                if (m == 3) { glds_piece_m0<3>(isrc, lane_off, idst); issue_next(); }
                                acc[4 * ol + 1] = MFMA16(av.y, b, (c == 0 && s == 0) ? bias4[1] : acc[4 * ol + 1]);
                                acc[4 * ol + 2] = MFMA16(av.z, b, (c == 0 && s == 0) ? bias4[2] : acc[4 * ol + 2]);
                                acc[4 * ol + 3] = MFMA16(av.w, b, (c == 0 && s == 0) ? bias4[3] : acc[4 * ol + 3]);
                                __builtin_amdgcn_sched_barrier(0);
                                if (m == 3) rd_slot = nx_slot;
                            }
                        }
                    }
                }
            }

            // ---- layer epilogue on the own rows, then the exchange ------------------------------------------------------------
            float own[2][16];
#pragma unroll
            for (int ol = 0; ol < 2; ++ol) {
                const int og = 2 * wave + ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    own[ol][4 * j + 0] = acc[4 * ol + j].x; own[ol][4 * j + 1] = acc[4 * ol + j].y;
                    own[ol][4 * j + 2] = acc[4 * ol + j].z; own[ol][4 * j + 3] = acc[4 * ol + j].w;
                }
                if (og < pd.nog) {
                    if (pd.relu) {
                        unsigned bits = 0;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            bits |= (own[ol][k] > 0.f ? 1u : 0u) << k;
                            own[ol][k] = relu1(own[ol][k]);
                        }
                        if (MASKS) mask_l[(pd.mask_slot * 2 + ol) * 256 + tid] = (unsigned short)bits;
                    } else if (BWD && pd.mask_slot >= 0) {
                        if (pd.kind == 4) {      // latent_in layer: gradients of the re-injected xyz / code rows, unmasked (wave 3)
                            if (a.lat_tile != 29) {          // 64-D codes: xyz at rows 445..447, code at 448..511
                                if (og == 6) { skipx[0] = own[ol][13]; skipx[1] = own[ol][14]; skipx[2] = own[ol][15]; }
                                if (og == 7) {
#pragma unroll
                                    for (int k = 0; k < 16; ++k) skipc[k] = own[ol][k];
                                }
                            } else if (og == 7) {            // 32-D codes: xyz at rows 477..479, code at 480..511 (code index 16 (j - 2) + 4 g + r)
                                skipx[0] = own[ol][5]; skipx[1] = own[ol][6]; skipx[2] = own[ol][7];
#pragma unroll
                                for (int k = 0; k < 8; ++k) skipc[k] = own[ol][8 + k];
                            }
                        }
                        const unsigned bits = mask_l[(pd.mask_slot * 2 + ol) * 256 + tid];
#pragma unroll
                        for (int k = 0; k < 16; ++k) own[ol][k] = ((bits >> k) & 1u) ? own[ol][k] : 0.f;
                    }
                }
            }
            exchange(own, pd.nog);

            if (ps == a.n_fwd - 1) {
                // final layer (512 -> 1) + tanh, redundantly in every wave; then the backward seed on the own rows
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
                    if (wave == 0) {      // wave-uniform: the four waves of a tile hold the same 16 points
                        const bool st = g == 0 && valid;
                        if (a.guard) prepass_guard(a, td.z, st, st ? a.out_sdf[a.index ? src : pidx + td.w] : 1.0f, y);
                        if (st) a.out_sdf[a.index ? src : pidx + td.w] = y;
                    }
                    if (MODE == 1 && valid && y > -a.th && y < a.th) {
                        // a candidate row of the render term (loss.py:88): export this lane's mask words of the wave's two output groups,
                        // all eight slots, in mlp_kernel<1>'s layout -- the mixed jacobian launch runs this sample backward-only from them
                        unsigned* mp = reinterpret_cast<unsigned*>(a.mask_buf + ((size_t)src * 4 + g) * 64) + wave;
#pragma unroll
                        for (int sl = 0; sl < MASK_SLOTS; ++sl)
                            mp[4 * sl] = (unsigned)mask_l[(sl * 2 + 0) * 256 + tid] | ((unsigned)mask_l[(sl * 2 + 1) * 256 + tid] << 16);
                    }
                    continue;
                }
                seed_backward(pd.mask_slot);
            }
        }

        // d y / d code = first-layer rows (group 0, held by every wave after the last exchange) + the latent_in skip rows that
        // only wave 3 captured; it writes the row
        if (BWD && wave == 3) {
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
            if (a.sdf_scatter && tile >= (a.direct.kind ? dl.nt0 : *a.scatter_tile_begin)) {
                const bool sc = valid && g == 3;
                if (a.guard) prepass_guard(a, td.z, sc, sc ? a.sdf_scatter[__float_as_int(pt.w)] : 1.0f, y);
                if (sc) a.sdf_scatter[__float_as_int(pt.w)] = y;
            }
        }
        // stores and LDS-DMA share vmcnt and may retire out of order: drain before counting again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

size_t mlp_split_lds_bytes() { return BIAS_BYTES + CODEBIAS_BYTES + SPLIT_MASK_BYTES + XCH_BYTES + SPLIT_RING_BYTES; }

template __global__ void mlp_split_kernel<0>(const MlpArgs);
template __global__ void mlp_split_kernel<1>(const MlpArgs);
template __global__ void mlp_split_kernel<2>(const MlpArgs);

hipError_t mlp_split_prepare_device() {
    const void* fns[3] = {reinterpret_cast<const void*>(&mlp_split_kernel<0>), reinterpret_cast<const void*>(&mlp_split_kernel<1>),
                          reinterpret_cast<const void*>(&mlp_split_kernel<2>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_split_lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// mode 0 forward, 1 forward + relu-mask export, 2 forward + backward (+ backward-only tiles from args.bwd_only_tile_begin on)
hipError_t launch_mlp_split(int mode, const MlpArgs& args, int n_blocks, hipStream_t stream) {
    if (mode == 2)
        hipLaunchKernelGGL(mlp_split_kernel<2>, dim3(n_blocks), dim3(256), mlp_split_lds_bytes(), stream, args);
    else if (mode == 1)
        hipLaunchKernelGGL(mlp_split_kernel<1>, dim3(n_blocks), dim3(256), mlp_split_lds_bytes(), stream, args);
    else
        hipLaunchKernelGGL(mlp_split_kernel<0>, dim3(n_blocks), dim3(256), mlp_split_lds_bytes(), stream, args);
    return hipGetLastError();
}

}  // namespace dsp
