// DeepSDF decoder forward + input gradient, CLUSTER form: FOUR workgroups on four CUs = ONE 16-point tile.
//
// A detection of the size SLAM really hands over (<= 250 LiDAR points, <= 450 rays: src/LocalMapping_util.cc:109-110,179-180,
// configs/config_kitti.json:17) gives the jacobian launch 40-60 tiles of 16 points.  In the latency form (mlp_split_kernel.hip) a tile is
// one workgroup: 117 MFLOP through one CU's fp32 MFMA pipes = 191 us at best (measured 241), on 60 of 256 CUs, ten times in sequence --
// half of the per-detection latency, and nothing inside one CU can shorten it.  Here a tile's OUTPUT ROWS are split over a cluster of
// four workgroups (16 waves, two 16-row tiles each, 256 MFMAs per wave and layer pass instead of 1024) and the layer's [512 x 16] result
// is handed round the cluster through L2 / Infinity Cache after every pass:
//
//   own rows -> LDS (the three sibling waves on this CU) and, write-through (sc1), -> the cluster's exchange buffer in global memory, as
//   16-byte units that carry their own validity: three floats + the exchange counter (a lane's eight floats = three units).  No
//   s_waitcnt, no flag: a 16-byte store of one lane lands as a whole, so a unit whose tag is the current counter IS the current data;
//   waves 0..2 of every workgroup each poll the twelve units of ONE remote workgroup (sc1 loads: the producer stored write-through, so
//   no acquire fence is needed: MI355X_MICROARCH.md, "visibility") until every tag matches, and drop the eight row tiles into LDS;
//   barrier; every wave reads the whole slab from LDS, as in the latency form.  (Until the last day of round 4 the payload was
//   followed by s_waitcnt vmcnt(0) + a flag store, and the readers polled the flag before fetching: one store acknowledgement and one
//   L2 round trip more per hand-off, 16 hand-offs per tile.)
//
// Two exchange buffers alternate.  EVERY exchange is a barrier of the whole cluster (round 5): a wave that has nothing to publish in an
// exchange (the final hand-off of a tile, the forward passes of a narrower decoder) still stores ONE tagged "presence" unit, and every
// workgroup waits for a tagged unit of every sibling -- so a workgroup is never more than one exchange ahead of its slowest sibling by
// construction, and the buffer it is about to overwrite (the exchange before the previous one) has been read by everybody.  Counters,
// not flags: the epoch base of a launch comes from the host, so nothing has to be cleared between launches (the host clears the
// buffers when its 32-bit base wraps).
// Every spin is bounded IN TIME (cl_spin_ticks of the 100 MHz wall clock, 2 ms by default: ~500 healthy hand-offs): a wave that gives up
// raises the error word and keeps publishing (so that nobody waits for it).  The error word is the fallback's trigger ON THE DEVICE: the
// latency-form kernel is launched behind this one in every iteration anyway (it normally returns at once for a list this short) and
// takes the list when the word is set -- the same iteration's rows are recomputed one workgroup per tile, nothing is discarded or
// repeated by the host, and the cluster launches of the run's remaining iterations return at entry.  A shared GPU (SLAM's detectors
// run on it from another thread, Tracking_util.cc:31-57) that keeps a member from being scheduled therefore costs <= a few ms once.
// Same k order, same bias seeding, same relu masks per output element as mlp_kernel<2> / mlp_split_kernel<true>: bit-identical
// gradients (tests/test_gpu_round4.py::test_cluster_kernel_is_exact).
//
// Weights: the wave's rows only, as its own stream (`wcluster`: for every pass the chunks of its two row tiles, two k-steps per 16-byte
// lane element), through a private LDS ring of 5 x 4 KiB by LDS-DMA -- the latency form's protocol with "k-step" read as "pair of
// k-steps": one ds_read_b128 and one DMA piece per four MFMAs.
//
// Replaces get_batch_sdf_jacobian (reconstruct/loss_utils.py:82-103) for detection-sized launches.
#include "dsp_internal.h"
#include "mlp_common.h"

namespace dsp {

constexpr int CL_SNB = 5;                                   // per-wave ring depth in mini-chunks
constexpr int CL_MINI = 4096;                               // 4 k-step pairs x 64 lanes x 16 B = 8 k-steps of two row tiles
constexpr int CL_MASK_BYTES = MASK_SLOTS * 256 * 2;         // [slot][tid] u16: 2 tiles x 4 rows = 8 bits used
constexpr int CL_XCH_BYTES = 32 * 64 * 16;                  // one layer's output slab in LDS: 32 row tiles x 64 lanes x float4
constexpr int CL_XCH_G_BYTES = CL_XCH_UNITS * 64 * 16;      // ... in the global exchange buffer: 16 wave slots x 3 tagged units x 64 lanes x 16 B
constexpr int CL_RING_BYTES = 4 * CL_SNB * CL_MINI;

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void mlp_cluster_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int pl = lane & 15;

    DirectList dl{0, 0, 0, 0};
    if (a.direct.kind) dl = direct_list(a.direct, SPLIT_TILE_PTS);
    const int n_tiles = a.direct.kind ? dl.n_tiles : *a.n_tiles;
    // (a one-object batch without a list: this kernel records the counts when the list is its own -- an empty one included; if this launch
    // then loses a hand-off the latency form repeats the list but does not count it again: the error word names the launch)
    if (a.direct.kind && n_tiles <= a.cluster_max_tiles && blockIdx.x == 0 && tid == 0 &&
        (n_tiles <= 0 || *reinterpret_cast<const volatile unsigned*>(a.cl_err) == 0u)) direct_commit(a.direct, dl);
    // the launch sequence issues this kernel AND the latency form for the same list; the tile count (known on the device only) picks one
    if (n_tiles > a.cluster_max_tiles || n_tiles <= 0) return;
    // an earlier launch of this run lost a hand-off: the latency form takes every list for the rest of the run (see above)
    if (*reinterpret_cast<const volatile unsigned*>(a.cl_err) != 0u) return;
    // cluster = 4 workgroups 8 apart in launch order: workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md), so the members of a
    // cluster share an L2 -- a speed hint only, nothing below depends on it
    const int cl = ((int)blockIdx.x >> 5) * 8 + ((int)blockIdx.x & 7);
    const int rank = ((int)blockIdx.x >> 3) & 3;
    const int n_clusters = (int)gridDim.x >> 2;
    if (a.cl_tiles_done && blockIdx.x == 0 && tid == 0) atomicAdd(a.cl_tiles_done, (double)n_tiles);
    if (cl >= n_tiles) return;                  // (all four members agree)
    const int u = 4 * rank + wave;              // wave slot in the cluster: row tiles 2u, 2u+1 of every layer
    const int og = u >> 1, half = u & 1;        // ... = tiles 2 half, 2 half + 1 of 64-row output group og

    float* bias_l = reinterpret_cast<float*>(smem);
    float* cb_l = reinterpret_cast<float*>(smem + BIAS_BYTES);
    unsigned short* mask_l = reinterpret_cast<unsigned short*>(smem + BIAS_BYTES + CODEBIAS_BYTES);
    f32x4* xch = reinterpret_cast<f32x4*>(smem + BIAS_BYTES + CODEBIAS_BYTES + CL_MASK_BYTES);
    char* ring_ptr = smem + BIAS_BYTES + CODEBIAS_BYTES + CL_MASK_BYTES + CL_XCH_BYTES + wave * (CL_SNB * CL_MINI);
    const unsigned ring0 = lds_addr(ring_ptr);

    for (int i = tid; i < a.n_bias_rows * WIDTH; i += 256) bias_l[i] = a.bias_tab[i];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- this wave's weight stream (wave-uniform state) -----------------------------------------------------------------
    const int total_minis = a.cl_len[u];
    const char* wbase = reinterpret_cast<const char*>(a.wcluster) + (size_t)a.cl_off[u] * CL_MINI;
    const unsigned lane_off = lane * 16;
    int issue_pos = 0, issue_slot = 0, rd_slot = 0;
    const char* isrc = wbase;
    unsigned idst = ring0;
    auto issue_next = [&]() {
        issue_pos = (issue_pos + 1 == total_minis) ? 0 : issue_pos + 1;
        issue_slot = (issue_slot + 1 == CL_SNB) ? 0 : issue_slot + 1;
        isrc = wbase + (size_t)issue_pos * CL_MINI;
        idst = ring0 + issue_slot * CL_MINI;
    };
#pragma unroll
    for (int i = 0; i < CL_SNB - 1; ++i) { glds_quarter(isrc, lane_off, idst); issue_next(); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (CL_SNB - 2)) : "memory");     // mini-chunk 0 has landed
    f32x4 abuf[4];        // A operands of k-step pairs, read two pairs ahead
    abuf[0] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16);
    abuf[1] = *reinterpret_cast<const f32x4*>(ring_ptr + lane * 16 + 1024);

    // ---- the cluster's exchange buffers and counters ----------------------------------------------------------------------
    // (descriptors from kernel arguments and blockIdx only: wave-uniform by construction)
    const size_t xb_bytes = (size_t)2 * CL_XCH_G_BYTES;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.cl_xbuf)) + (size_t)cl * xb_bytes, 0, (int)xb_bytes, 0x00020000);
    unsigned ep = a.cl_epoch_base;              // exchange counter: every wave of the cluster counts the same sequence
    bool dead = false;                           // a spin ran out: publish, never wait again (the host discards the run)

    float sin_[128];   // full input slab of the current pass (every wave holds all 512 rows of the 16 points)
    f32x4 acc[2];      // this wave's two row tiles
#pragma unroll
    for (int i = 0; i < 128; ++i) sin_[i] = 0.f;
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // One hand-off round the cluster.  pub[0], pub[1]: this wave's row tiles 2u, 2u + 1 (published when `publish`).  Tile t takes part in
    // the exchange when t < n_lo or t >= hi_from (ordinary passes: the layer's 4 * nog row tiles and hi_from = 32; the final hand-off:
    // tiles 0..3 and the latent_in skip tiles at the top).  Slot = tile index, so tile t always comes from workgroup t / 8.
    auto exchange = [&](const f32x4 (&pub)[2], bool publish, int n_lo, int hi_from) {
        ++ep;
        const unsigned par = (ep & 1u) * CL_XCH_G_BYTES;
        lds_barrier();                                     // every wave of this CU is done reading the previous exchange from LDS
        if (publish) {
            xch[(2 * u) * 64 + lane] = pub[0];
            xch[(2 * u + 1) * 64 + lane] = pub[1];
        }
        if (!(a.cl_fault && rank == 3)) {                 // (fault injection: a member that never publishes)
            const unsigned ub = par + ((3 * u) * 64 + lane) * 16;
            if (publish) {
                const u32x4_t u0 = {__float_as_uint(pub[0].x), __float_as_uint(pub[0].y), __float_as_uint(pub[0].z), ep};
                const u32x4_t u1 = {__float_as_uint(pub[0].w), __float_as_uint(pub[1].x), __float_as_uint(pub[1].y), ep};
                const u32x4_t u2 = {__float_as_uint(pub[1].z), __float_as_uint(pub[1].w), 0u, ep};
                __builtin_amdgcn_raw_buffer_store_b128(u0, xrs, ub, 0, 16);                 // aux 16 = sc1: write-through
                __builtin_amdgcn_raw_buffer_store_b128(u1, xrs, ub + 64 * 16, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(u2, xrs, ub + 2 * 64 * 16, 0, 16);
            } else {
                // nothing to hand over in this exchange: ONE presence unit, so that the siblings know this workgroup has finished reading
                // the previous exchange (every exchange is a cluster-wide barrier)
                const u32x4_t u0 = {0u, 0u, 0u, ep};
                __builtin_amdgcn_raw_buffer_store_b128(u0, xrs, ub, 0, 16);
            }
        }
        if (wave < 3) {                                    // this wave relays ONE remote workgroup's tiles into LDS
            const int r = (rank + 1 + wave) & 3;
            u32x4_t v[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = (u32x4_t){0u, 0u, 0u, 0u};
            if (!dead) {
                unsigned spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {       // wave slot 4 r + sl of the remote workgroup = its row tiles 8 r + 2 sl, + 1
                        const int tl = 8 * r + 2 * sl;
                        if (tl < n_lo || tl >= hi_from) {
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                v[3 * sl + k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, par + ((3 * (4 * r + sl) + k) * 64 + lane) * 16, 0, 16);
                        } else {                           // not part of this exchange: its presence unit only
                            v[3 * sl] = __builtin_amdgcn_raw_buffer_load_b128(xrs, par + ((3 * (4 * r + sl)) * 64 + lane) * 16, 0, 16);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int tl = 8 * r + 2 * sl;
                        if (tl < n_lo || tl >= hi_from) ok = ok && v[3 * sl][3] == ep && v[3 * sl + 1][3] == ep && v[3 * sl + 2][3] == ep;
                        else ok = ok && v[3 * sl][3] == ep;
                    }
                    if (__all(ok)) break;
                    // bounded in TIME, not in polls: the clock is read only once a poll has failed, and then on every 16th
                    if (spins == 0) t0 = wall_clock64();
                    if ((++spins & 15u) == 0u && wall_clock64() - t0 > (unsigned long long)a.cl_spin_ticks) {
                        dead = true;
                        if (lane == 0) atomicCAS(a.cl_err, 0u, a.cl_epoch_base | 1u);     // non-zero, and says WHICH launch gave up (mlp_split_kernel: no second direct_commit)
                        __threadfence();      // the word is out before anything this wave publishes from here on (stale data under a current tag)
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int tl = 8 * r + 2 * sl;
                if (tl < n_lo || tl >= hi_from) {
                    xch[tl * 64 + lane] = (f32x4){__uint_as_float(v[3 * sl][0]), __uint_as_float(v[3 * sl][1]), __uint_as_float(v[3 * sl][2]),
                                                  __uint_as_float(v[3 * sl + 1][0])};
                    xch[(tl + 1) * 64 + lane] = (f32x4){__uint_as_float(v[3 * sl + 1][1]), __uint_as_float(v[3 * sl + 1][2]), __uint_as_float(v[3 * sl + 2][0]),
                                                        __uint_as_float(v[3 * sl + 2][1])};
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own stores out, relayed tiles in -- and the weight ring drained: stores and LDS-DMA share the counter
        lds_barrier();
    };
    // all live tiles of the exchange -> the input slab
    auto gather = [&](int n_live) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < n_live) {
                const f32x4 v = xch[t * 64 + lane];
                sin_[4 * t + 0] = v.x; sin_[4 * t + 1] = v.y; sin_[4 * t + 2] = v.z; sin_[4 * t + 3] = v.w;
            }
        }
    };

    for (int tile = cl; tile < n_tiles; tile += n_clusters) {
        const int4 td = a.direct.kind ? direct_tile(a.direct, dl, tile, SPLIT_TILE_PTS) : a.tiles[tile];
        const bool valid = pl < td.y;
        const int pidx = td.x + (valid ? pl : 0);
        float4 pt = a.pts[pidx];
        if (!valid) pt = make_float4(0.f, 0.f, 0.f, 0.f);
        lds_barrier();        // previous tile's readers of cb_l are done
        reinterpret_cast<float4*>(cb_l)[tid] = reinterpret_cast<const float4*>(a.code_bias + (size_t)td.z * a.code_bias_stride)[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        {   // layer 0 on the VALU, redundantly in every wave (the slab is needed everywhere); masks of the own tiles only
            const float* w0 = bias_l + a.w0_row * WIDTH + 4 * g;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int row = 16 * t;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb_l + row + 4 * g);
                const f32x4 wx = *reinterpret_cast<const f32x4*>(w0 + row);
                const f32x4 wy = *reinterpret_cast<const f32x4*>(w0 + WIDTH + row);
                const f32x4 wz = *reinterpret_cast<const f32x4*>(w0 + 2 * WIDTH + row);
#pragma unroll
                for (int r = 0; r < 4; ++r) sin_[4 * t + r] = fmaf(wz[r], pt.z, fmaf(wy[r], pt.y, fmaf(wx[r], pt.x, c0[r])));
            }
            // relu masks of the own two tiles: the same fmaf chains once more, addressed by the wave slot (selecting them out of the
            // 128 registers by a run-time tile index costs ~270 registers of live ranges: measured, 67 spills)
            unsigned bits = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 16 * (2 * u + j);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb_l + row + 4 * g);
                const f32x4 wx = *reinterpret_cast<const f32x4*>(w0 + row);
                const f32x4 wy = *reinterpret_cast<const f32x4*>(w0 + WIDTH + row);
                const f32x4 wz = *reinterpret_cast<const f32x4*>(w0 + 2 * WIDTH + row);
#pragma unroll
                for (int r = 0; r < 4; ++r) bits |= (fmaf(wz[r], pt.z, fmaf(wy[r], pt.y, fmaf(wx[r], pt.x, c0[r]))) > 0.f ? 1u : 0u) << (4 * j + r);
            }
            mask_l[0 * 256 + tid] = (unsigned short)bits;
#pragma unroll
            for (int i = 0; i < 128; ++i) sin_[i] = relu1(sin_[i]);
        }
        f32x4 skip[2];          // the latent_in layer's gradient rows of the re-injected [xyz | code] that THIS wave produced (unmasked)
        skip[0] = skip[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float y = 0.f;
        float gfirst = 0.f;

        for (int ps = 0; ps < a.n_pass; ++ps) {
            const PassDesc pd = a.pass[ps];
            const bool mine = og < pd.nog;
            if (pd.kind == 2) {
                // (row 445 = tile 27, row 477 = tile 29 with 32-D codes; both are row 13 of their tile: lane group 3, registers 1..3)
                if (g == 3) {
                    if (a.lat_tile == 29) { sin_[117] = pt.x; sin_[118] = pt.y; sin_[119] = pt.z; }
                    else { sin_[109] = pt.x; sin_[110] = pt.y; sin_[111] = pt.z; }
                }
            } else if (pd.kind == 5) {
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

            // ---- this wave's two row tiles: nchunks x 8 k-step pairs, four MFMAs per pair ---------------------------------------
            if (mine) {
                f32x4 bias2[2];
                if (pd.bias_row != -1) {
                    const float* bp = (pd.bias_row == -2 ? cb_l + WIDTH : bias_l + pd.bias_row * WIDTH) + 32 * u + 4 * g;
                    bias2[0] = *reinterpret_cast<const f32x4*>(bp);
                    bias2[1] = *reinterpret_cast<const f32x4*>(bp + 16);
                } else {
                    bias2[0] = bias2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c < pd.nchunks) {
#pragma unroll
                        for (int p2 = 0; p2 < 8; ++p2) {
                            const int m = p2 & 3;       // pair inside the mini-chunk
                            const int nx_slot = (rd_slot + 1 == CL_SNB) ? 0 : rd_slot + 1;
                            const char* cbp = ring_ptr + rd_slot * CL_MINI + lane * 16;
                            const char* nbp = ring_ptr + nx_slot * CL_MINI + lane * 16;
                            if (m == 2) {
                                // the next mini-chunk is about to be read: it has landed once only the two mini-chunks after it and the
                                // two pieces issued so far in this one are still in flight (mlp_split_kernel.hip)
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (CL_SNB - 3) + 2) : "memory");
                            }
                            if ((p2 & 1) == 0) {
                                __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0), vmcnt / expcnt untouched
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int q = 0; q < 2; ++q) {
                                    const int sp = m + 2 + q;
                                    abuf[(p2 + 2 + q) % 4] = (sp < 4) ? *reinterpret_cast<const f32x4*>(cbp + sp * 1024)
                                                                      : *reinterpret_cast<const f32x4*>(nbp + (sp - 4) * 1024);
                                }
                            }
                            const f32x4 av = abuf[p2 % 4];       // {tile 0, k-step 2 p2 | tile 1, 2 p2 | tile 0, 2 p2 + 1 | tile 1, 2 p2 + 1}
                            const float b0 = sin_[16 * c + 2 * p2], b1 = sin_[16 * c + 2 * p2 + 1];
                            acc[0] = MFMA16(av.x, b0, (c == 0 && p2 == 0) ? bias2[0] : acc[0]);
                            // refill of the slot behind the read pointer: one DMA piece per pair, behind an MFMA
                            if (m == 0) { glds_set_dst(idst); glds_piece_m0<0>(isrc, lane_off, idst); };
                            if (m == 1) glds_piece_m0<1>(isrc, lane_off, idst);
                            if (m == 2) glds_piece_m0<2>(isrc, lane_off, idst);
                            if (m == 3) { glds_piece_m0<3>(isrc, lane_off, idst); issue_next(); }
                            acc[1] = MFMA16(av.y, b0, (c == 0 && p2 == 0) ? bias2[1] : acc[1]);
                            acc[0] = MFMA16(av.z, b1, acc[0]);
                            acc[1] = MFMA16(av.w, b1, acc[1]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (m == 3) rd_slot = nx_slot;
                        }
                    }
                }
            }

            // ---- layer epilogue on the own rows ---------------------------------------------------------------------------------
            f32x4 own[2] = {acc[0], acc[1]};
            if (mine) {
                if (pd.relu) {
                    unsigned bits = 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            bits |= (own[j][r] > 0.f ? 1u : 0u) << (4 * j + r);
                            own[j][r] = relu1(own[j][r]);
                        }
                    mask_l[pd.mask_slot * 256 + tid] = (unsigned short)bits;
                } else if (pd.mask_slot >= 0) {
                    if (pd.kind == 4) {      // latent_in layer: gradients of the re-injected xyz / code rows, kept unmasked by their owner
                        if (2 * u + 1 >= a.lat_tile) { skip[0] = own[0]; skip[1] = own[1]; }
                    }
                    const unsigned bits = mask_l[pd.mask_slot * 256 + tid];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) own[j][r] = ((bits >> (4 * j + r)) & 1u) ? own[j][r] : 0.f;
                }
            }
            if (ps + 1 < a.n_pass) {
                exchange(own, mine, 4 * pd.nog, 32);
                gather(4 * pd.nog);
            } else {
                // final hand-off: the first layer's code rows (tiles 0..3, waves 0 and 1 of workgroup 0) and, in the same round, the latent_in
                // skip rows their owners kept, at their own tile slots: tile lat_tile (xyz at its rows 13..15) and the code tiles above it
                // (64-D codes: 27 | 28..31, 32-D: 29 | 30, 31).  lat_tile is odd: the wave that owns it publishes (lat_tile - 1, lat_tile).
                const bool has_skip = 2 * u + 1 >= a.lat_tile;
                f32x4 pub[2];       // (element-wise selects: a reference to one of two arrays would pin both in scratch memory)
                pub[0] = has_skip ? skip[0] : own[0];
                pub[1] = has_skip ? skip[1] : own[1];
                exchange(pub, mine || has_skip, 4, a.lat_tile - 1);
                gather(4);
            }

            if (ps == a.n_fwd - 1) {
                // final layer (512 -> 1) + tanh, redundantly in every wave; then the backward seed -- for ALL rows, locally: the relu output
                // of the last hidden layer is positive exactly where its mask bit is set
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
                const float d = 1.f - y * y;
#pragma unroll
                for (int t = 0; t < 32; ++t) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + 16 * t);
                    sin_[4 * t + 0] = sin_[4 * t + 0] > 0.f ? d * w4.x : 0.f;
                    sin_[4 * t + 1] = sin_[4 * t + 1] > 0.f ? d * w4.y : 0.f;
                    sin_[4 * t + 2] = sin_[4 * t + 2] > 0.f ? d * w4.z : 0.f;
                    sin_[4 * t + 3] = sin_[4 * t + 3] > 0.f ? d * w4.w : 0.f;
                }
            }
        }

        // d y / d code = first-layer rows (tiles 0..3, held by every wave after the final hand-off) + the latent_in skip rows (slots 4..8 of
        // the last exchange); wave 0 of workgroup 0 writes the row
        // A wave of this launch has given up (the error word is set before it publishes anything else): this tile may have been computed from
        // stale hand-offs.  Its rows are recomputed by the latency-form kernel behind this launch; what must NOT happen is that a wrong sdf
        // replaces the prepass value (the recomputation's guard compares against it) or reaches the guard (a false trip would re-run the
        // object without the prepass and widen the handle's margins).
        const bool launch_ok = __hip_atomic_load(a.cl_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        if (u == 0) {
            const int n_code_tiles = 31 - a.lat_tile;              // 4 (64-D codes) or 2 (32-D)
            const f32x4 sx = xch[a.lat_tile * 64 + lane];           // the xyz tile: rows 13..15 = lane group 3, components 1..3
            float* orow = a.out_grad + (size_t)(pidx + td.w) * GRAD_STRIDE;
            if (valid) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (t < n_code_tiles) sc = xch[(a.lat_tile + 1 + t) * 64 + lane];
                    float4 o4;
                    o4.x = sin_[4 * t + 0] + sc.x;
                    o4.y = sin_[4 * t + 1] + sc.y;
                    o4.z = sin_[4 * t + 2] + sc.z;
                    o4.w = sin_[4 * t + 3] + sc.w;
                    *reinterpret_cast<float4*>(orow + 16 * t + 4 * g) = o4;
                }
            }
            const float s0 = __shfl(sx.y, pl + 48);
            const float s1 = __shfl(sx.z, pl + 48);
            const float s2 = __shfl(sx.w, pl + 48);
            const float sk = (g == 0) ? s0 : (g == 1) ? s1 : s2;
            if (valid) orow[64 + g] = (g < 3) ? (gfirst + sk) : y;
            if (launch_ok && a.sdf_scatter && tile >= (a.direct.kind ? dl.nt0 : *a.scatter_tile_begin)) {
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

size_t mlp_cluster_lds_bytes() { return BIAS_BYTES + CODEBIAS_BYTES + CL_MASK_BYTES + CL_XCH_BYTES + CL_RING_BYTES; }

hipError_t mlp_cluster_prepare_device() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_cluster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_cluster_lds_bytes());
}

// n_clusters: multiple of 8 (one cluster per XCD and group of 32 workgroups); grid = 4 x n_clusters workgroups, all of which must be
// resident at once (one per CU: the LDS footprint admits no second one) -- the caller sizes it to the CU count
hipError_t launch_mlp_cluster(const MlpArgs& args, int n_clusters, hipStream_t stream) {
    hipLaunchKernelGGL(mlp_cluster_kernel, dim3(4 * n_clusters), dim3(256), mlp_cluster_lds_bytes(), stream, args);
    return hipGetLastError();
}

}  // namespace dsp
