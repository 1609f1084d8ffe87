// Internal definitions shared by the HIP kernels and the host side of libdspgn.
// (The public C-ABI is include/dsp_gn.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdexcept>

namespace dsp {

// ---- decoder geometry the MLP kernel is built for (DeepSDF as used by DSP-SLAM) -------------
// hidden width 512, code length 64, one latent_in layer; see DESIGN.md "MLP kernel".
constexpr int WIDTH = 512;          // hidden width == slab rows
constexpr int CODE_LEN = 64;
constexpr int IN_DIM = CODE_LEN + 3;
constexpr int TILE_PTS = 64;        // points per workgroup tile (4 waves x 16)
constexpr int SPLIT_TILE_PTS = 16;  // points per workgroup tile of the latency form (4 waves share them, mlp_split_kernel.hip)
constexpr int WAVE_PTS = 16;
constexpr int KSTEPS_PER_CHUNK = 16;            // 16 MFMA k-steps (64 slab rows) per weight chunk
constexpr int CHUNK_BYTES = KSTEPS_PER_CHUNK * 64 * 16;   // 16 KiB: [kstep][lane] float4
constexpr int MAX_PASSES = 32;
constexpr int GRAD_STRIDE = 68;     // per-point output row: d/dcode[64], d/dxyz[3], sdf

// One "pass" = one dense layer traversal (forward layer or backward layer) of the weight stream.
struct PassDesc {
    int16_t nog;        // number of 64-row output groups
    int16_t nchunks;    // number of 64-row K chunks per output group
    int16_t bias_row;   // row of the LDS bias table to start the accumulators from, -1 = zero, -2 = per-object code bias
    int16_t relu;       // 1: relu + store mask (forward hidden layer)
    int16_t mask_slot;  // forward: slot the relu mask is stored to; backward: slot applied; -1 none
    int16_t kind;       // 1 fwd hidden, 2 fwd latent_in layer, 3 bwd, 4 bwd latent_in, 5 bwd first layer (code rows only)
    int32_t chunk_base; // first chunk of this pass in the packed stream
};

// Tile list of a ONE-object batch without a list: the decoder kernels derive their tiles from the object's counters themselves (the
// k_build_tiles launch in front of them goes: 4.6 us + a kernel boundary, twice per iteration of a detection).  kind 0 = off (tiles /
// n_tiles are read); 1 = forward list over the V in-sphere samples (with the "< 10 samples" rule of loss.py:73-74 and the work
// counters k_build_tiles mode 0 keeps); 3 = jacobian list, n0 surface points then the P band rows (k_build_tiles mode 3).
struct ObjState;
struct DirectTiles {
    int kind;
    int off0, n0, off1;         // first point of segment 0 / its length where the host knows it (surface points) / first point of segment 1
    ObjState* st;               // the object's state: status, V, P
    double* counters;           // dsp_batch::counters()
};
struct DirectList { int n_tiles, nt0, n0, n1; };

struct MlpArgs {
    const float* wstream;       // packed weight stream (fwd passes then bwd passes), chunk-major
    const float* bias_tab;      // [n_bias_rows][512]: hidden biases, last row = final layer weights
    float b_last;               // final layer bias
    int n_bias_rows;            // hidden biases b1.., final-layer weights (row wlast_row), first-layer xyz columns (rows w0_row..+2)
    int wlast_row, w0_row;
    int n_fwd;                  // number of forward passes (hidden layers)
    int n_pass;                 // fwd (+ bwd when BWD)
    int total_chunks;           // chunks consumed per tile
    PassDesc pass[MAX_PASSES];
    // work list (device memory, produced on device)
    const int* n_tiles;         // [1] one past the last tile
    const int* tile_begin;      // optional [1]: first tile (0 when null)
    const int4* tiles;          // [n_tiles] {first point, n points, object (code index), output offset added to the point index}
    const float4* pts;          // object-frame points (xyz, w unused)
    const int* index;           // forward kernel, optional: point i of a tile is pts[index[first + i]] and its sdf goes to
                                // out_sdf[index[first + i]] (front-to-back ray passes evaluate a subset of the sample list)
    const float* codes;         // code of object o at codes + o * code_stride
    int code_stride;            // in floats (multiple of 4)
    const float* code_bias;     // per object: [0..511] = W0[:, :64] code + b0, [512..1023] = W_lat[:, code cols] code + b_lat
    int code_bias_stride;       // in floats (multiple of 4)
    unsigned short* mask_buf;   // MODE 1 writes / MODE 3 reads: [sample][lane group 4][layer 8][output group 8] relu bits (16 per word)
    const float* sdf_in;        // MODE 3: sdf of the samples, as written by the forward launches
    float th;                   // MODE 1: export masks of samples with |sdf| < th
    int seed_slot;              // MODE 3: mask slot of the last hidden layer
    int lat_tile;               // 16-row slab tile of the latent_in layer's re-injected xyz: 27 (rows 445..447, 64-D codes) or 29 (477..479, 32-D)
    float* out_sdf;             // FWD: [n_points]
    float* out_grad;            // BWD: [n_points][GRAD_STRIDE]
    // always-on guard of the low-precision pre-classification (prepass_guard, mlp_common.h): the fp32 kernels compare every sample they
    // re-decode with the prepass value it replaces.  guard = per-object words {lp_delta (float), trips, max error (float bits)} of object 0,
    // guard_stride words apart (inside ObjState); nullptr = off
    unsigned* guard;
    int guard_stride;
    float* sdf_scatter;         // BWD, optional: tiles from *scatter_tile_begin on also store their sdf at sdf_scatter[int(pt.w)] (speculative band rows)
    const int* scatter_tile_begin;
    const int* bwd_only_tile_begin;   // latency form, mixed launch: tiles from *this on run the backward sweep only, from exported masks (mlp_split_kernel<2>); nullptr = none
    const float* wsplit;        // latency form (mlp_split_kernel): the same chunks laid out per wave (see pack_decoder)
    int split_off[4];           //   first chunk of wave w's stream inside wsplit
    int split_len[4];           //   chunks wave w consumes per tile (forward + backward)
    int split_len_fwd[4];       //   ... of which the forward passes (a prefix of the wave's stream)
    unsigned long long* clk;    // optional: block 0 writes {clock64, wall_clock64} at entry and exit (effective shader clock)
    // cluster form (mlp_cluster_kernel.hip): four workgroups per 16-point tile, layer rows split over their 16 waves
    const float* wcluster;      //   the same chunks laid out per wave slot (two row tiles, two k-steps per 16-byte element; pack_decoder)
    int cl_off[16];             //   first 4 KiB mini-chunk of wave slot u inside wcluster
    int cl_len[16];             //   mini-chunks slot u consumes per tile (forward + backward)
    float* cl_xbuf;             //   exchange buffers: [cluster][2 parities][16 wave slots x 3 units][64 lanes] x 16 B (3 floats + counter tag)
    unsigned* cl_err;           //   raised when a bounded spin ran out: the latency-form kernel behind the cluster kernel then takes the list (device-side fallback)
    unsigned cl_spin_ticks;     //   bound of every spin in ticks of the 100 MHz wall clock (CL_SPIN_TICKS_DEFAULT = 2 ms)
    double* cl_tiles_done;      //   optional: + the list's tile count when the cluster kernel takes it (dsp_stats.n_cluster_tiles)
    unsigned cl_epoch_base;     //   counter value this launch starts from (the host advances it by CL_EPOCH_STRIDE per launch)
    int cl_fault;               //   fault injection (tests): workgroup 3 of every cluster never publishes -> its siblings' bounded spins run out
    int cluster_max_tiles;      //   the cluster kernel runs lists of up to this many tiles ...
    int split_min_tiles;        //   ... and the latency form (mlp_split_kernel<true>) lists of at least this many (0 = always)
    DirectTiles direct;         // one-object batches: no tile list (kind != 0)
};
constexpr unsigned CL_SPIN_TICKS_DEFAULT = 200000;   // 2 ms: ~500 healthy hand-offs of ~3.5 us
constexpr int CL_EPOCH_STRIDE = 4096;      // exchanges a launch may count: 15 per tile and cluster
constexpr int CL_XCH_UNITS = 16 * 3;         // tagged 16-byte units per lane in one parity of a cluster's global exchange buffer (mlp_cluster_kernel.hip)


// ---- low-precision prepass (mlp_lp_kernel.hip): f16 / bf16 MFMA forward used ONLY to classify ray samples ------------
// Occupancy is exactly 0 for sdf >= th and exactly 1 for sdf <= -th (loss_utils.py:40-48), so a sample whose low-precision
// sdf is farther than a calibrated margin from the band needs no fp32 decode at all (DESIGN.md "Prepass").
constexpr int LP_TILE_PTS = 128;    // points per workgroup tile (4 waves x 32)
constexpr int LP_TILE_PTS_SMALL = 64;   // ... of the latency form (4 waves x 16: one column block per wave), for lists that fill less than half the chip
constexpr int LP_WAVE_PTS = 32;
constexpr int LP_NBUF = 8;          // LDS ring depth in 16 KiB chunks
constexpr int LP_KSTEPS_PER_CHUNK = 8;   // chunk = 8 k-steps (of 16 slab rows) x 2 row tiles (of 32 output rows) x 1 KiB
constexpr int LP_MAX_PASSES = 12;
constexpr int LP_XYZ_KSTEPS = 2;    // k-steps reserved for the split-precision xyz products (f16 uses one, bf16 two)

// xyz enters layer 0 and the latent_in layer as sums of low-precision parts: x = x1 + x2 (+ x3), w = w1 + w2 (+ w3),
// x.w ~ sum of the listed (x part, w part) products -- 22-24 significant bits, so the prepass error is set by the hidden
// layers' activation rounding only.  Entry [kstep][term] = 4 * xpart + wpart (parts 1-based), 0 = unused; MFMA k index
// of (term t, coordinate c) = 3 t + c.
constexpr unsigned char LP_XYZ_TERMS[2][LP_XYZ_KSTEPS][5] = {
    {{4 * 1 + 1, 4 * 2 + 1, 4 * 1 + 2, 4 * 2 + 2, 0}, {0, 0, 0, 0, 0}},                                   // f16
    {{4 * 1 + 1, 4 * 2 + 1, 4 * 3 + 1, 4 * 1 + 2, 4 * 2 + 2}, {4 * 1 + 3, 4 * 3 + 2, 4 * 2 + 3, 0, 0}},   // bf16
};

struct LpPass {
    int16_t nog;        // 64-row output groups (2 MFMA row tiles each)
    int16_t nchunks;    // chunks per group = ceil(k-steps / 8)
    int16_t bias_row;   // row of the fp32 bias table; -2 = per-object latent_in code bias, -3 = per-object layer-0 code bias
    int16_t kind;       // 0 first layer (xyz k-steps only), 1 hidden, 2 latent_in (slab k-steps, then xyz k-steps)
    int16_t npad;       // latent_in layer: padding k-steps between the slab rows and the xyz k-steps (the xyz B operands sit at
                        // fixed k-steps: 0, 1 of the first layer; the last two of the latent_in layer's groups)
    int16_t last;       // 1: final 512->1 layer + tanh follow in the epilogue
    int32_t chunk_base;
};

struct LpArgs {
    const void* wstream;        // packed 16-bit weight stream, chunk-major (pack_decoder_lp)
    const float* bias_tab;      // the fp32 kernel's table: hidden biases, final-layer weights (row wlast_row)
    float b_last;
    int n_bias_rows, wlast_row;
    int n_pass, total_chunks;
    LpPass pass[LP_MAX_PASSES];
    const int* n_tiles;
    const int4* tiles;          // {first point, n points (<= 128), object, output offset}
    const float4* pts;
    const int* index;           // optional indirection, as MlpArgs::index
    const float* code_bias;     // per object [2][512] fp32 (k_code_bias)
    int code_bias_stride;
    float* out_sdf;
    unsigned long long* clk;
    DirectTiles direct;         // one-object batches: no tile list (kind != 0)
};

// ---- low-precision compute mode (mlp_lpj_kernel.hip): forward + input gradient on 16-bit MFMA operands ------------------------------
constexpr int LPJ_MAX_PASSES = 8;   // eight forward passes (the prepass stream: mlp_lpj_fwd_kernel) or eight backward passes (mlp_lpj_bwd_kernel)
struct LpjArgs {
    const void* wstream;        // the prepass stream (forward kernel) or the transposed stream (backward kernel: pack_decoder_lpj_host), chunk-major
    const float* bias_tab;
    float b_last;
    int n_bias_rows, wlast_row;
    int total_chunks;
    LpPass pass[LPJ_MAX_PASSES];   // backward passes, last hidden layer first: kind 3 hidden, 4 latent_in (also yields the re-injected rows' gradient), 5 first layer (two output groups)
    const int* n_tiles;
    const int4* tiles;          // {first point, n points (<= 128), object, output offset}
    const float4* pts;
    const float* code_bias;     // per object [2][512] fp32 (k_code_bias)
    int code_bias_stride;
    float* out_grad;            // [point + offset][GRAD_STRIDE]: d/dcode[64], d/dxyz[3], sdf -- what mlp_kernel<2> writes
    uint4* mask_buf;            // [tile][8 layers][4 waves][2 column blocks][64 lanes] x 16 B: the relu masks (forward kernel writes, backward kernel reads)
    int lat_tile;               // 27 (64-D codes) / 29 (32-D)
    unsigned long long* clk;
};

// ---- Gauss-Newton batch state ------------------------------------------------------------------
constexpr int DSP_STATUS_GOOD = 0;
constexpr int DSP_STATUS_FEW = 1;    // < 10 in-sphere samples (loss.py:73-74)
constexpr int DSP_STATUS_NAN = 2;    // NaN loss / singular system (optimizer.py:135-136,149-150)
constexpr int DSP_STATUS_SKIP = 3;   // internal: not part of this (partial re-)run; never leaves the library (k_init_state run_mask, k_finalize)
constexpr int MAX_DEPTH_SAMPLES = 64;
constexpr int TRACE_STRIDE = 5344;   // 71*71 H | 71 b | 71 dx | 16 t_oc | 64 code | V m K | vsum lo,hi | ksum lo,hi | pad | 64 depths

struct ObjConst {           // static layout of one object inside the batch arrays
    int pts_off, n_pts;     // surface points (camera frame)
    int ray_off, n_rays, n_fg;
    int depth_off;          // foreground depths
    int samp_off;           // segment of the in-sphere sample list (capacity roundup64(n_rays * D))
    int jsdf_off;           // jacobian-point segment, surface term (capacity roundup64(n_pts))
    int jren_off;           // jacobian-point segment, render term (capacity = sample capacity)
    int pad;
};

struct ObjState {           // per-object optimiser state, lives on the device for the whole run
    float t_oc[16];         // camera -> object Sim(3)/SE(3)   (t_obj_cam)
    float t_co[16];
    float code[CODE_LEN];
    float depths[MAX_DEPTH_SAMPLES];
    float scale, dmin, dmax, loss;
    int status, V, m, K;
    int n_alive;
    int P;                  // samples selected for the current front-to-back pass
    unsigned vsum, ksum;    // order-independent checksums of the in-sphere set and of the kept (jacobian) sample set
    // prepass: this object's margin (from the magnitude of its CURRENT code, lp_delta_of), and what the guard saw
    float lp_delta;
    unsigned guard_trips;   // waves that re-decoded a sample whose prepass value was off by >= lp_delta / 2
    unsigned guard_err;     // largest |sdf_lp - sdf_fp32| over the re-decoded samples (float bits)
    int pad0;
};

#ifdef __HIPCC__
// (wave-uniform: kernel arguments and scalar loads only; no side effects -- every workgroup of every kernel that may take the list calls it)
__device__ __forceinline__ DirectList direct_list(const DirectTiles& d, int tile_pts) {
    DirectList l{0, 0, 0, 0};
    const int status = d.st->status;
    if (d.kind == 1) {
        const int V = d.st->V;
        l.n0 = (status == DSP_STATUS_GOOD && V >= 10) ? V : 0;          // "< 10 in-sphere samples": the object fails (direct_commit records it)
    } else {
        const bool good = status == DSP_STATUS_GOOD;
        l.n0 = good ? d.n0 : 0;
        l.n1 = good ? d.st->P : 0;
    }
    l.nt0 = (l.n0 + tile_pts - 1) / tile_pts;
    l.n_tiles = l.nt0 + (l.n1 + tile_pts - 1) / tile_pts;
    return l;
}
// ONE thread of the ONE kernel that takes the list: what k_build_tiles records besides the list
__device__ __forceinline__ void direct_commit(const DirectTiles& d, const DirectList& l) {
    if (d.kind == 1) {
        if (d.st->status == DSP_STATUS_GOOD && d.st->V < 10) d.st->status = DSP_STATUS_FEW;
        d.counters[4] += (double)l.n0;
        d.counters[2] += (double)l.n0;
    } else {
        d.counters[1] += (double)l.n0;
        d.counters[3] += (double)l.n1;
    }
}
__device__ __forceinline__ int4 direct_tile(const DirectTiles& d, const DirectList& l, int tile, int tile_pts) {
    if (tile < l.nt0) return make_int4(d.off0 + tile * tile_pts, min(tile_pts, l.n0 - tile * tile_pts), 0, 0);
    const int t = tile - l.nt0;
    return make_int4(d.off1 + t * tile_pts, min(tile_pts, l.n1 - t * tile_pts), 0, 0);
}
#endif
static_assert(sizeof(ObjState) % 16 == 0, "ObjState is addressed as float4-aligned rows");

// Prepass margin as a function of the code's largest entry: calibrate_prepass (dsp_gn.hip) measures the largest |sdf_lp - sdf_fp32|
// with codes drawn at each of these magnitudes; delta is interpolated between them (linear extrapolation above, capped at 0.5).
constexpr int LP_NMAG = 5;
struct LpDeltaTab {
    float mag[LP_NMAG];
    float delta[LP_NMAG];
};

struct GnParamsDev {
    float k1, k2, k3, k4, b1, b2, lr, s_damp, cut_off;
    int n_depth, pose_only;
    int code_len;       // of the decoder (32 or 64); the state always carries CODE_LEN entries
    LpDeltaTab lp;      // prepass margin table of the dtype in use (all entries equal when the caller fixed delta)
};

// kernels_mlp / kernels_gn launchers
size_t mlp_lds_bytes(int mode);
void launch_code_bias(const float* codew, const float* b0, const float* blat, const float* codes, int code_stride, float* out, int n_obj, hipStream_t s);
hipError_t mlp_prepare_device();
hipError_t launch_mlp(int mode, const MlpArgs& args, int n_blocks, hipStream_t stream);   // mode: see mlp_kernel
hipError_t mlp_split_prepare_device();
hipError_t launch_mlp_split(int mode, const MlpArgs& args, int n_blocks, hipStream_t stream);   // 16-point tiles; mode 0 forward, 1 forward + mask export, 2 forward + backward (+ backward-only tiles)
hipError_t mlp_cluster_prepare_device();
hipError_t launch_mlp_cluster(const MlpArgs& args, int n_clusters, hipStream_t stream);   // 16-point tiles, four workgroups each; grid = 4 x n_clusters (n_clusters a multiple of 8), all resident
hipError_t mlp_lp_prepare_device();
hipError_t launch_mlp_lp(bool bf16, const LpArgs& args, int n_blocks, hipStream_t stream, int tile_pts = LP_TILE_PTS);   // 128- or 64-point tiles, forward only
hipError_t mlp_lpj_prepare_device();
hipError_t launch_mlp_lpj(int which, bool bf16, const LpjArgs& args, int n_blocks, hipStream_t stream);   // 128-point tiles; which: 0 forward + mask export, 1 backward from the masks
// run_mask (optional, B bytes): objects with a zero byte are left out of the run (status DSP_STATUS_SKIP, state and result row untouched);
// summary: the run's counter words, zeroed here (summary_words of them)
void launch_init_state(ObjState* st, const float* t, const float* codes, const float* scale, const float* depths /*optional B x 64*/, int B, int D, int pose_only,
                       const LpDeltaTab& lp, const unsigned char* run_mask, unsigned* summary, int summary_words, hipStream_t s);
void launch_sample_count(const ObjConst* oc, ObjState* st, const float* rays, unsigned long long* m, int* c, int D, int maxR, int B, hipStream_t s);
void launch_scan_rays(const ObjConst* oc, ObjState* st, const int* cnt, int* off, int which, int B, hipStream_t s);
void launch_sample_write(const ObjConst* oc, const ObjState* st, const float* rays, const unsigned long long* m, const int* off, float4* spts,
                         float* ssdf, unsigned char* alive, int D, int maxR, int B, hipStream_t s);
void launch_surface(const ObjConst* oc, const ObjState* st, const float* pts, float4* jpts, float2* jaux, int maxM, int B, hipStream_t s);
void launch_build_tiles(const ObjConst* oc, ObjState* st, int B, int mode, int4* tiles, int* n_tiles, double* counters, int add_v, int tile_pts,
                        int cnt_slot, hipStream_t s, int apply_few = 0);   // mode 3 = mode 1 with the band samples (P) in place of the kept render rows (K)   // cnt_slot: counter the point count of a mode 0 / 2 list is added to
// th = cut_off; the widened band of object b is |sdf_lp| < th + st[b].lp_delta.  guard_salt: samples OUTSIDE the band whose id hash
// (xor salt) selects them (1/8 of the ring just beyond the band, 1/512 farther out) are listed too, so that the fp32 kernel re-decodes them
// and prepass_guard compares (0 = no guard samples)
void launch_tail_tiles(const int4* tiles, int* n_tiles, int4* tiles16, int* n_tiles16, int n_cu, hipStream_t s);   // see k_tail_tiles
void launch_band_select(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float* ssdf, float th,
                        unsigned guard_salt, int* pcnt, int* poff, int* plist, int maxR, int B, hipStream_t s);
// wave-per-ray forms (latency path; one wave per ray, 16 rays per workgroup, whole chip) of front = sample_count + scan + sample_write +
// surface, band = count + scan + write, render tail = scan + sum_m + render_write (behind k_render_scan): list segments from running counters
// (ObjState::V / ::P) instead of scans; kept rows stay in ray-major order.  See gn_kernels.hip.
void launch_front_wave(const ObjConst* oc, ObjState* st, const float* rays, const float* pts, unsigned long long* raymask, int* raycnt, int* rayoff,
                       float4* spts, float* ssdf, unsigned char* alive, float4* jpts, float2* jaux, int D, int maxR, int maxM, int B, hipStream_t s);
void launch_band_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float* ssdf, float th, unsigned guard_salt,
                      int* plist, const float4* spts, float4* jpts, int* srow, int maxR, int B, hipStream_t s);   // jpts / srow: speculative band rows (or null)
void launch_render_tail_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float4* spts, const float* sdeds,
                             const float* ray_res, const int* kcnt, const int* mcnt, float4* jpts, float2* jaux, const int* srow, int* jrow, int maxR, int B,
                             hipStream_t s);
void launch_prepass_audit(const ObjConst* oc, const ObjState* st, const float* ssdf, const float* saudit, float th, unsigned* out,
                          int B, hipStream_t s);
void launch_render_scan(const ObjConst* oc, ObjState* st, const unsigned long long* m, const int* off, const float* ssdf, const float* depth,
                        float* sdeds, float* ray_res, int* kcnt, int* mcnt, int D, float th, int maxR, int B, hipStream_t s);
void launch_render_write(const ObjConst* oc, const ObjState* st, const int* raycnt, const int* rayoff, const int* koff, const float4* spts,
                         const float* sdeds, const float* ray_res, float4* jpts, float2* jaux, int maxR, int B, hipStream_t s);
// one front-to-back forward pass: fixed depth-index range [j0, j1) (hint == nullptr), or per-ray adaptive ranges where
// j0 = margin added to the hint in pass 0 and j1 = step of the middle passes
struct PassSpec {
    int j0, j1, n_depth, pass, last;
    unsigned char* hint;   // per ray: depth index of the first solid sample in the previous GN iteration
    unsigned char* plo;    // per ray: end of the range decoded so far in this iteration
};
void launch_pass_select(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const unsigned char* alive,
                        int* pcnt, const PassSpec& ps, int maxR, int B, hipStream_t s);
void launch_pass_write(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff, const unsigned char* alive,
                       const int* poff, int* plist, const PassSpec& ps, int maxR, int B, hipStream_t s);
void launch_pass_update(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff, unsigned char* alive,
                        const float* ssdf, float th, int use_lp_delta, const PassSpec& ps, int maxR, int B, hipStream_t s);   // use_lp_delta: a ray stops at sdf <= -(th + st[b].lp_delta)
void launch_sum_m(const ObjConst* oc, ObjState* st, const int* mcnt, int B, hipStream_t s);
void launch_gram(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux, const float* jgrad, const int* jrow,
                 const unsigned char* alive, float* partials, int n_slices, float b_sdf, float b_render, int robust, int n_terms, int B, hipStream_t s);
void launch_jrows(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux, const float* jgrad, const int* jrow, int term, float* rows,
                  int cap, hipStream_t s);   // jrow (optional): render row i takes its gradient from jgrad row jrow[i]
void launch_solve(const ObjConst* oc, ObjState* st, const float* partials, double* gsum, int n_slices, const GnParamsDev& prm, int iter,
                  float* trace, const float* codew, const float* b0, const float* blat, float* cbias, const float* depths_next, int B,
                  hipStream_t s);    // cbias: next iteration's code bias; depths_next: optional B x 64 override of the next iteration's depth samples
void launch_inlier_filter(const ObjConst* oc, ObjState* st, const float* jgrad, unsigned char* alive, int maxM, int B, hipStream_t s);
constexpr int DSP_RESULT_WIDTH_DEV = 82;   // == DSP_RESULT_WIDTH (dsp_gn.h): t_cam_obj 16 | code 64 | loss | status
void launch_finalize(ObjState* st, const float* scale, int B, int pose_only, float* packed, unsigned* guard_out /*optional B x 3*/, hipStream_t s);

hipError_t launch_debug_lie(int kind, const float* x_dev, float* out_dev, int n_depth, hipStream_t s);   // testing: exp_sim3 / exp_se3 / rotation prior as k_solve evaluates them

// ---- mesh extraction (mesh_kernels.hip) ---------------------------------------------------------
constexpr int MC_MAX_TRI = 5;
struct McTables {                        // marching-cubes case table, generated on the host (mc_build_tables)
    unsigned char n_tri[256];            // triangles of a cell whose corner-inside bits are the index
    unsigned char tri[256][16];          // 3 cube-edge ids per triangle, 0xff padded
    unsigned char edge_off[12][4];       // cube edge -> (offset of the owning grid point along axes 0, 1, 2; axis)
};
void mc_build_tables(McTables& t);
int mc_num_blocks(int n_pts);
hipError_t launch_grid_points(float4* pts, int n, float voxel_size, int regular, hipStream_t s);
hipError_t launch_mc_count(const float* vol, int n0, int n1, int n2, float level, const McTables* tab, int2* block_sums, long long* totals,
                           hipStream_t s);
hipError_t launch_mc_emit(const float* vol, int n0, int n1, int n2, float level, const McTables* tab, const int2* block_off, float spacing,
                          float origin, float* verts, int* vidmap, int* faces, hipStream_t s);

}  // namespace dsp
