/* dsp_gn.h -- C ABI of libdspgn: MI355X (gfx950) DeepSDF shape-code + pose Gauss-Newton optimiser.
 *
 * Drop-in boundary for ONE hot path of DSP-SLAM (reference: JingwenWang95/DSP-SLAM).  The reference has
 * no FFI of its own for this path: C++ reaches it through an embedded CPython interpreter
 * (src/LocalMapping.cc:38-40, src/LocalMapping_util.cc:109-110,179-196,391-426) calling the Python
 * functions listed below.  Each entry point here cites the reference interface it replaces; the Python
 * mirror package (dsp_slam_amd/reconstruct, dsp_slam_amd/deep_sdf) binds these through ctypes and keeps
 * the reference's Python names, so the C++ caller is unchanged.
 *
 * Conventions: plain C types; all pointers are HOST pointers unless a name ends in _dev; row-major
 * float32; every function returns 0 on success or a negative DSP_E_* code (dsp_last_error() gives
 * text).  A handle owns one device, one HIP stream and its device memory; calls on one handle (and on
 * batches created from it) are serialised INSIDE the library (a per-handle mutex), so they may come from
 * any thread; different handles are independent.  Nothing here depends on Python or PyTorch.
 */
#ifndef DSP_GN_H
#define DSP_GN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_OK 0
#define DSP_E_ARG (-1)      /* bad argument / unsupported decoder geometry */
#define DSP_E_HIP (-2)      /* HIP runtime error */
#define DSP_E_NOMEM (-3)
#define DSP_E_STATE (-4)

#define DSP_CODE_LEN 64
#define DSP_GRAD_DIM 67     /* d sdf / d [code(64), x, y, z] */

typedef struct dsp_handle dsp_handle;
typedef struct dsp_batch dsp_batch;

/* Folded decoder weights, layer k: weights[k] is (out_dims[k] x in_dims[k]) row-major, biases[k] (out_dims[k]).
 * Replaces deep_sdf/workspace.py:202-223 (config_decoder) + Decoder.__init__ (deep_sdf/deep_sdf_decoder.py:10-73):
 * weight-norm is folded by the caller, W = g * v / ||v||_row.  Supported geometry (checked by dsp_create, DSP_E_ARG otherwise): code_len
 * 64 or 32; ONE hidden width, a multiple of 16 and at most 512 (narrower decoders run embedded in the 512-row slabs); 2..8 hidden layers;
 * exactly one latent_in layer >= 2 whose input is [h | code | xyz] (the layer in front of it emits width - code_len - 3 rows); final layer
 * width -> 1 + tanh.  The optimiser state always carries 64 code entries (the unused ones of a 32-D decoder stay zero).  The low-precision
 * prepass additionally needs an even number of hidden layers (DeepSDF's 8); other decoders run with the prepass off. */
typedef struct dsp_decoder_desc {
    int32_t n_layers;            /* number of Linear layers (9 for DSP-SLAM) */
    int32_t code_len;            /* 64 or 32 */
    int32_t latent_in;           /* index of the layer whose input is [h | code | xyz] (4), or -1 */
    const int32_t* out_dims;     /* [n_layers] */
    const int32_t* in_dims;      /* [n_layers] */
    const float* const* weights; /* [n_layers] */
    const float* const* biases;  /* [n_layers] */
} dsp_decoder_desc;

/* Hyper-parameters read by Optimizer.__init__ (reconstruct/optimizer.py:27-43). */
typedef struct dsp_gn_params {
    float k1, k2, k3, k4;        /* render / sdf / code-prior / rotation-prior weights */
    float b1, b2;                /* Huber thresholds: render, sdf */
    float lr;                    /* learning_rate */
    float s_damp;                /* scale_damping */
    int32_t num_iterations;      /* joint_optim.num_iterations */
    int32_t num_depth_samples;   /* optimizer.num_depth_samples (<= 64) */
    float cut_off;               /* optimizer.cut_off_threshold */
    int32_t pose_only_iterations;/* pose_only_optim.num_iterations */
} dsp_gn_params;

/* Per-object status of a batch run (ragged failures do not poison the batch). */
#define DSP_OBJ_GOOD 0
#define DSP_OBJ_FEW_SAMPLES 1   /* < 10 in-sphere ray samples: compute_render_loss returned None (loss.py:73-74) */
#define DSP_OBJ_NAN 2           /* NaN loss, e.g. K == 0 (optimizer.py:135-136,149-150) */

/* Counters and device timings of the last run of a batch. */
typedef struct dsp_stats {
    double n_fwd_points;         /* forward-only decoder points actually evaluated (<= sum of V: samples behind a solid sample are skipped) */
    double n_jac_points;         /* points decoded forward + backward: sum of M (surface), plus K (render rows) without mask reuse */
    double ms_total;             /* HIP-event time of the whole run on the handle's stream */
    double ms_mlp_fwd;           /* summed time of the forward-only decoder kernel launches (0 with kernel timing off: dsp_batch_set_kernel_timing) */
    double ms_mlp_jac;           /* summed time of the forward+gradient decoder kernel launches (likewise) */
    int32_t n_mlp_fwd_launches;
    int32_t n_mlp_jac_launches;
    double n_insphere_points;    /* sum over iterations and objects of V, the in-sphere sample count the reference decodes */
    double n_render_rows;        /* render rows that ran the backward sweep only (mask reuse on): sum of K, else 0 */
    /* prepass (ABI version 2): with it on, n_fwd_points counts only the samples that still needed the fp32 kernel */
    double n_prepass_points;     /* samples decoded by the low-precision prepass kernel */
    double ms_mlp_prepass;       /* summed time of the prepass kernel launches */
    int32_t n_mlp_prepass_launches;
    int32_t prepass_mode;        /* DSP_PREPASS_* the run used */
    float prepass_delta;         /* margin the run used: samples with |sdf_lp| < cut_off + delta went to the fp32 kernel */
    float prepass_max_err;       /* audit only: max |sdf_lp - sdf_fp32| over the audited samples */
    double prepass_misclassified;/* audit only: samples classified against their fp32 value (must be 0) */
    double prepass_audited;      /* audit only: samples compared */
    /* always-on prepass guard (ABI version 3): every sample the fp32 kernel re-decodes -- the widened band plus a stratified sample of
     * the classified ones -- is compared with the prepass value it replaces */
    double prepass_guard_trips;  /* waves that saw |sdf_lp - sdf_fp32| >= half the object's margin (0 in a healthy run) */
    double prepass_guard_objects;/* objects with at least one trip */
    float prepass_guard_max_err; /* largest |sdf_lp - sdf_fp32| over the compared samples */
    int32_t prepass_guard_rerun; /* 1: the guard tripped; the objects it tripped on were run again with the prepass off (their results come from that run) */
    double n_cluster_tiles;      /* (ABI version 4) 16-point jacobian tiles that ran in the cluster form: four workgroups per tile (dsp_batch_set_cluster_tiles) */
    int32_t cluster_fallback;    /* (ABI version 5) 1: a cluster-form launch of this run lost a hand-off within its time bound (2 ms: a busy shared GPU kept one of the
                                  * four workgroups off its CU); the latency-form kernel recomputed that launch and took the run's remaining lists on the device --
                                  * same results, a few ms later; the handle keeps the cluster form off for its next 64 runs */
    int32_t cluster_cooldown;    /* (ABI version 6) runs for which the handle still keeps the cluster form off after a lost hand-off (0 = it is in use again) */
} dsp_stats;

/* ---- lifetime --------------------------------------------------------------------------------- */
/* Thread safety: calls on ONE handle (and on batches created from it) are serialised inside the library (one call at a time per handle);
 * different handles are independent.  Callers that release the GIL around these calls (ctypes, pybind11) may therefore call from any thread. */
int dsp_create(const dsp_decoder_desc* decoder, int device, dsp_handle** out);
/* Destroys the handle AND every resident batch still alive on it.  A dsp_batch* is an opaque, generation-tagged TOKEN, not an address: once its
 * batch is gone -- destroyed, or taken by dsp_destroy of its handle -- every call that is handed the token returns DSP_E_ARG (dsp_batch_destroy:
 * does nothing), also after the memory has been reused for another batch.  So the two destroy calls are safe in either order and from any
 * thread (finalisers at interpreter exit run in no particular order); dsp_destroy waits for calls in flight on the handle's batches, and
 * dsp_batch_destroy for calls in flight on that batch (a call that starts after it sees a stale token). */
void dsp_destroy(dsp_handle* h);
/* A handle keeps device blocks (up to 1 GiB, size-classed) and pinned host staging of dropped one-shot batches for its next call.  dsp_trim
 * hands all of it back to the runtime: for a process that shares the GPU with another allocator (torch, RCCL, a second handle).  Destroying
 * a LARGE resident batch (more than 512 MiB left parked behind it) trims the cache to the footprint of one KITTI-size detection (64 MiB) on
 * its own; detection-sized batches created and destroyed per frame keep their blocks. */
int dsp_trim(dsp_handle* h);
/* Priority of the handle's HIP stream among the queues of the device: 1 = highest, 0 = default, -1 = lowest the device offers.  Where the GPU
 * is shared -- DSP-SLAM runs its detectors on it from the Tracking thread (src/Tracking_util.cc:31-57) while this path runs in the
 * LocalMapping thread (src/LocalMapping_util.cc:165-203) -- the integrator decides who yields: -1 lets the detectors' kernels go first,
 * 1 shortens a detection's ~110 dependent kernels beside them.  The stream is re-created (after a synchronisation); batches of the handle
 * pick the new one up at their next run.  Results do not depend on it. */
int dsp_set_stream_priority(dsp_handle* h, int priority);
const char* dsp_last_error(const dsp_handle* h);   /* h may be NULL: error of the last failed dsp_create */
int dsp_abi_version(void);
/* Which compiler produced this library (hipcc --version at build time, clang version, HIP header version) -- the decoder kernels rely on
 * properties of the generated code that the build checks by disassembly (dsp_slam_amd/build.py: check_isa); and the HIP runtime / driver
 * versions of the box it is running on (hipRuntimeGetVersion / hipDriverGetVersion encodings). */
const char* dsp_build_info(void);
int dsp_runtime_versions(int* hip_runtime, int* hip_driver);
int dsp_device_count(void);   /* number of gfx950 devices dsp_create accepts (ordinals 0 .. n-1); 0 without a HIP device */

/* ---- decoder ---------------------------------------------------------------------------------- */
/* decode_sdf(decoder, lat_vec, x)  -- reconstruct/loss_utils.py:51-79.  pts (n,3) object frame -> sdf (n). */
int dsp_decode_sdf(dsp_handle* h, const float* code, const float* pts, int64_t n, float* sdf_out);
/* The same decoder evaluated by the low-precision PREPASS kernel (v_mfma_f32_16x16x32_f16 / _bf16, fp32 accumulation; xyz, code
 * and biases enter at fp32 accuracy, hidden activations are rounded to 16 bits per layer).  Never used for results: the optimiser
 * uses it only to classify ray samples whose occupancy is exactly 0 or 1 (sdf outside the cut-off band by more than a calibrated
 * margin, reconstruct/loss_utils.py:40-48); exposed for calibration and tests. */
#define DSP_PREPASS_OFF 0
#define DSP_PREPASS_F16 1
#define DSP_PREPASS_BF16 2
#define DSP_PREPASS_SMALL_TILES 0x100   /* or-ed into dtype: run the kernel's 64-point-tile form (one 16-point column block per wave) */
int dsp_decode_sdf_prepass(dsp_handle* h, int dtype, const float* code, const float* pts, int64_t n, float* sdf_out);
/* The same point set decoded for n_codes shape codes in ONE launch: sdf_out[c * n + i].  Batched form of the
 * MeshExtractor grid decode (reconstruct/optimizer.py:217-218) / the per-object loop of extract_map_objects.py:46-63. */
int dsp_decode_sdf_multi(dsp_handle* h, const float* codes, int64_t n_codes, const float* pts, int64_t n, float* sdf_out);
/* get_batch_sdf_jacobian through the 16-bit kernels of the LOW-PRECISION COMPUTE MODE (dsp_batch_set_compute below): f16 / bf16 matrix operands,
 * fp32 accumulation, forward and backward.  NOT the parity path -- exposed so that its accuracy can be measured point by point. */
#define DSP_COMPUTE_F32 0
#define DSP_COMPUTE_F16 1
#define DSP_COMPUTE_BF16 2
int dsp_sdf_jacobian_lp(dsp_handle* h, int dtype, const float* code, const float* pts, int64_t n, float* sdf_out, float* grad_out);
/* get_batch_sdf_jacobian(decoder, lat_vec, x, 1) -- reconstruct/loss_utils.py:82-103.
 * sdf_out (n), grad_out (n, 67) = d sdf / d [code, xyz]. */
int dsp_sdf_jacobian(dsp_handle* h, const float* code, const float* pts, int64_t n, float* sdf_out, float* grad_out);

/* ---- residual terms --------------------------------------------------------------------------- */
/* compute_sdf_loss(decoder, pts_surface_cam, t_obj_cam, latent_vector) -- reconstruct/loss.py:22-43.
 * Outputs: jac_pose (n,7), jac_code (n,64), res (n). */
int dsp_compute_sdf_loss(dsp_handle* h, const float* pts_cam, int64_t n, const float* t_obj_cam, const float* code,
                         float* jac_pose, float* jac_code, float* res);
/* compute_render_loss(decoder, ray_directions, depth_obs, t_obj_cam, sampled_ray_depth, latent_vector, th)
 * -- reconstruct/loss.py:46-152.  *k_out = number of rows K, or -1 when the reference returns None
 * (< 10 in-sphere samples).  Outputs need capacity n_rays * n_depths rows; row order = the reference's
 * (ray-major, depth-minor).  v_out / m_out (optional) receive the ragged set sizes V and m.  m (samples with |sdf| < th) is
 * INFORMATIONAL: with early ray termination or the prepass on, samples behind a ray's first solid sample are not decoded (their
 * transmittance is exactly 0, so they cannot reach K, the rows or the results), and they are not counted in m either -- it can be smaller
 * than the reference's m.  V and K are the reference's. */
int dsp_compute_render_loss(dsp_handle* h, const float* rays, int64_t n_rays, const float* depth_obs,
                            const float* t_obj_cam, const float* sampled_depth, int32_t n_depths, const float* code,
                            float th, int64_t* k_out, float* jac_pose, float* jac_code, float* res, int64_t* v_out,
                            int64_t* m_out);

/* ---- optimiser -------------------------------------------------------------------------------- */
/* Optimizer.reconstruct_object for a ragged batch of independent objects -- reconstruct/optimizer.py:88-203.
 * Object i uses pts[pts_off[i]..pts_off[i+1]) (camera frame, (M_i,3)), rays[ray_off[i]..ray_off[i+1]) ((R_i,3);
 * the first n_fg[i] rows are foreground and pair with depth[depth_off[i] + r]), t_cam_obj[i] (4x4 Sim(3)
 * object->camera initial estimate) and codes_in[i] (64; NULL => zero start, optimizer.py:96-99).
 * Outputs per object: t_cam_obj (16), code (64), loss (k1*L_render + k2*L_sdf at the last linearisation
 * point, :155), status (DSP_OBJ_*).  For a failed object t_cam_obj/code hold the last state. */
int dsp_reconstruct_batch(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off,
                          const float* pts, const int64_t* ray_off, const float* rays, const int64_t* depth_off,
                          const float* depth, const float* t_cam_obj_in, const float* codes_in,
                          float* t_cam_obj_out, float* codes_out, float* loss_out, int32_t* status_out);

/* Optimizer.estimate_pose_cam_obj for a ragged batch -- reconstruct/optimizer.py:45-86.
 * t_co_se3_in[i] (4x4 SE(3)), scale[i], pts as above, codes[i] (64).  Output t_co_se3_out[i] (4x4 SE(3)).
 * The inputs are not modified (the reference scales the caller's matrix in place, :53-54). */
int dsp_estimate_pose_batch(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off,
                            const float* pts, const float* t_co_se3_in, const float* scale, const float* codes,
                            float* t_co_se3_out);

/* ---- mesh extraction (reference MeshExtractor.extract_mesh_from_code, reconstruct/optimizer.py:206-223 ->
 *      create_voxel_grid / decode_sdf / convert_sdf_voxels_to_mesh, reconstruct/utils.py:97-140) -----------------------------------
 * dsp_extract_mesh decodes the vol_dim^3 grid over [-1,1]^3 and runs marching cubes (level 0) on the device; the SDF volume
 * never leaves HBM.  The sample points are the ones the reference's create_voxel_grid really produces: under torch >= 1.6 its
 * `overall_index.long() / vol_dim` is true division, so the grid is sheared by up to one voxel (y index + z / N, x index + y / N + z / N^2);
 * DSP_MESH_REGULAR_GRID samples the regular lattice instead.  Vertices are placed on the regular lattice either way, as in the reference.  It returns the mesh size; dsp_mesh_fetch then copies the last mesh extracted on this handle:
 * vertices (n_vertices x 3 float32, object frame: index * voxel_size - 1, voxel_size = 2 / (vol_dim - 1)) and faces
 * (n_faces x 3 int32 vertex ids, normals by the right-hand rule along +grad sdf, i.e. outward).  One vertex per sign-changing grid
 * edge (shared between faces, like scikit-image's output); order: vertices by (grid point, axis), faces by (cell, case-table order).
 * An empty mesh (no sign change) is n_vertices = n_faces = 0 and not an error here (the Python mirror raises scikit-image's
 * ValueError for it).  The case table is a generated classic table, not Lewiner's: see oracle/mc_oracle.py. */
#define DSP_MESH_REGULAR_GRID 1   /* flags: sample the regular lattice instead of the reference's sheared grid (see below) */
int dsp_extract_mesh(dsp_handle* h, const float* code, int32_t vol_dim, int32_t flags, int64_t* n_vertices, int64_t* n_faces);
/* The marching-cubes step alone on a host volume (n0 x n1 x n2, axis 0 slowest): vertices = index * spacing + origin
 * (replaces convert_sdf_voxels_to_mesh, utils.py:119-140, with spacing = 2 / (n - 1), origin = -1, level = 0). */
int dsp_marching_cubes(dsp_handle* h, const float* volume, int32_t n0, int32_t n1, int32_t n2, float level, float spacing, float origin,
                       int64_t* n_vertices, int64_t* n_faces);
/* Copies the last mesh extracted on this handle.  n_vertices / n_faces are the counts the caller sized its buffers for (as returned by
 * its dsp_extract_mesh / dsp_marching_cubes call); if the handle's last mesh has other counts -- another thread extracted in between --
 * nothing is copied and DSP_E_STATE is returned. */
int dsp_mesh_fetch(dsp_handle* h, float* vertices, int64_t n_vertices, int32_t* faces, int64_t n_faces);

/* ---- device-resident batches (bench / steady-state serving: inputs stay in HBM between runs) --- */
int dsp_batch_create(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off,
                     const float* pts, const int64_t* ray_off, const float* rays, const int64_t* depth_off,
                     const float* depth, const float* t_cam_obj_in, const float* codes_in, dsp_batch** out);
int dsp_batch_run(dsp_batch* b);        /* resets the state to the uploaded initial estimate, runs, synchronises */
int dsp_batch_results(dsp_batch* b, float* t_cam_obj_out, float* codes_out, float* loss_out, int32_t* status_out);
int dsp_batch_stats(dsp_batch* b, dsp_stats* out);
/* ---- settings (all optional; results are identical, bit for bit, for every value) ------------------------------------------------------ */
/* Number of front-to-back depth ranges the forward decoder is run in per iteration (exact early ray termination: a ray
 * stops being sampled behind its first solid sample, where the transmittance is exactly 0).  0 = automatic (ten uniform
 * ranges for large batches; otherwise 2-3 per-ray ranges steered by where each ray stopped in the previous iteration); 1 = decode every in-sphere sample like the reference does.
 * With the prepass on these are the ranges of the low-precision kernel; the fp32 kernel then runs once per iteration over the samples the
 * prepass could not classify. */
int dsp_batch_set_ray_passes(dsp_batch* b, int n_passes);
/* Exact low-precision pre-classification of the forward ray samples.  The render term only sees clamp(sdf, -cut_off, cut_off)
 * (reconstruct/loss_utils.py:40-48): occupancy is exactly 0 for sdf >= cut_off and exactly 1 for sdf <= -cut_off.  With the
 * prepass on, every candidate sample is first decoded by an f16 (or bf16) MFMA kernel at 16x the fp32 matrix rate; samples with
 * |sdf_lp| >= cut_off + delta are classified by that value alone, rays stop behind their first certainly-solid sample, and only the
 * samples inside the widened band are decoded by the fp32 kernel (one launch per iteration).  delta must exceed the largest
 * |sdf_lp - sdf_fp32| of the decoder.  The default is calibrated PER DECODER at dsp_create: 6x the largest difference over 32 768 seeded
 * unit-ball points x 2 codes per code magnitude (dsp_prepass_calibration_table), not below 5e-4 (f16) / 3e-3 (bf16); the margin of every
 * object follows the largest entry of its CURRENT code.
 * mode: -1 automatic (f16 when the decoder geometry is supported), DSP_PREPASS_OFF / _F16 / _BF16; delta < 0 = default. */
int dsp_batch_set_prepass(dsp_batch* b, int mode, float delta);
/* The guard is ON by default and costs ~0.3 % more fp32 points: with the prepass on, the fp32 kernel also re-decodes a sample of the
 * samples the prepass classified -- 1/8 of the ring th + delta <= |sdf_lp| < th + 2 delta, where an error between delta and 2 delta would
 * misclassify, 1/512 of everything farther out; a different sample every launch -- and compares EVERY sample it decodes (the whole
 * widened band included) with the prepass value it replaces.  A difference of half the object's margin or more trips the guard:
 * dsp_batch_run then runs the OBJECTS IT TRIPPED ON again with the prepass off and returns those results for them (the healthy objects
 * keep theirs: dsp_stats.prepass_guard_rerun = 1, prepass_guard_objects = how many were re-run); the stand-alone dsp_compute_render_loss
 * re-evaluates the term the same way.  Where the run used the calibrated margins, the handle's margins are raised to 4 x the error seen
 * (at most 0.125; dsp_prepass_reset_guard undoes it); a margin forced through dsp_batch_set_prepass leaves them alone.
 * on = 0 turns the guard off (tests: what an unguarded run would have returned). */
int dsp_batch_set_prepass_guard(dsp_batch* b, int on);
/* HIP events around every decoder launch, i.e. the dsp_stats.ms_mlp_* fields: -1 = automatic (on for batches of more than 16 objects --
 * the bench's roofline needs them; off for latency-sized batches, where an event record between two kernels is a queue packet of its own),
 * 0 = off, 1 = on.  Launch COUNTS and ms_total are filled either way. */
int dsp_batch_set_kernel_timing(dsp_batch* b, int mode);
/* OPT-IN, NON-PARITY fast path (BASELINE.json north_star: "fp32/bf16 GEMMs"; SURVEY.md 8(d): "optional non-parity fast path, reported
 * separately").  DSP_COMPUTE_F32 (default): everything that reaches a result is decoded in fp32 -- the mode every parity statement is about.
 * DSP_COMPUTE_F16 / _BF16: the decoder runs on 16-bit matrix operands with fp32 accumulation everywhere -- the ray samples' sdf come from the
 * 16-bit forward kernel alone (no fp32 re-decode of the band, no guard), the jacobian rows from 16-bit forward + backward kernels
 * (mlp_lpj_kernel.hip); thresholds, scans, the Gram matrices and the fp64 solve are unchanged.  The precision class of the reference's own
 * published runs (PyTorch 1.10 on Ampere multiplied in TF32: 10-bit mantissas, as f16).  How far H, b and the results move:
 * profiles/r06_lp_compute.md, tests/test_gpu_lp_compute.py.  DSP_E_ARG for a decoder geometry other than DeepSDF's (eight hidden layers, the
 * latent_in layer fourth) and for pose-only batches.  A DETECTION-SIZED batch (<= 16 objects whose surface points + band samples fit one round of
 * 16-point tiles over the CUs: SLAM's own per-detection calls) keeps the fp32 latency path with the mode set -- it is faster there (2.89 against
 * 3.40 ms) and exact; one cfg2-size object: 14.0 -> 6.0 ms in the mode (profiles/r06_latency_ab.md). */
int dsp_batch_set_compute(dsp_batch* b, int mode);
/* The iteration count of the following runs (instead of dsp_gn_params.num_iterations / pose_only_iterations given at creation). */
int dsp_batch_set_iterations(dsp_batch* b, int32_t n);
/* The calibration dsp_create made for this decoder at a ZERO code: largest |sdf_lp - sdf_fp32| it measured and the margin it derived
 * (DSP_E_STATE when the decoder's geometry has no prepass kernel); the full table follows. */
int dsp_prepass_calibration(dsp_handle* h, int dtype, float* max_err, float* delta);
/* The whole calibration: the prepass error grows with the hidden activations, hence with the code, so dsp_create measures it with codes
 * drawn uniformly in +-mag on every entry for mag in {0, 0.15, 0.5, 1, 2} (5 entries each: mags, largest error, margin = max(floor,
 * 6 x largest error up to that magnitude), capped at 0.5).  Every object's margin follows the largest entry of its CURRENT code,
 * piecewise linearly through this table, re-evaluated after every Gauss-Newton step.  guard_err: largest error a guard trip on this
 * handle has reported (the margins returned are already raised to 4 x that).  Any pointer may be NULL. */
int dsp_prepass_calibration_table(dsp_handle* h, int dtype, float* mags, float* max_err, float* delta, float* guard_err);
/* Forget what earlier guard trips on this handle have left behind (dsp_prepass_calibration_table: guard_err): the margins return to the
 * decoder's calibration. */
int dsp_prepass_reset_guard(dsp_handle* h);

/* ---- testing: ONE door for the forms the library chooses between by itself -----------------------------------------------------------
 * The launch sequence has several bit-identical forms per stage, chosen from the batch's size.  Tests pin a form to
 * compare it with the one it replaces; an integrator has no reason to.  value: -1 automatic, 0 off, 1 on where applicable, unless noted. */
#define DSP_DBG_MASK_REUSE 1        /* render rows run the backward sweep only, from relu masks the forward launches exported (throughput form: a launch of their own) */
#define DSP_DBG_SPLIT_ROWS 2        /* latency form of the decoder launches: 16-point tiles, a layer's rows split over the four waves */
#define DSP_DBG_TAIL_SPLIT 3        /* the last, at most half-empty round of a 64-point forward launch as 16-point tiles in a launch of its own */
#define DSP_DBG_WAVE_BOOKKEEPING 4  /* per-ray bookkeeping as one wave per ray with running counters (1) instead of count / scan / write launches (0) */
#define DSP_DBG_SPECULATIVE_BAND 5  /* prepass on: unclassified samples go straight into the jacobian launch (no forward launch of their own) */
#define DSP_DBG_MIXED_REUSE 6       /* mask reuse inside the latency form: backward-only tiles in the same launch as the surface points' tiles */
#define DSP_DBG_CLUSTER_TILES 7     /* cluster form of the jacobian launch: four workgroups per 16-point tile (lists of <= 128 tiles); 1 also bypasses the handle's cool-down */
#define DSP_DBG_DIRECT_TILES 8      /* one-object batches: the decoder kernels derive their tile lists themselves */
#define DSP_DBG_PREPASS_TILE 9      /* value = 128 or 64 points per prepass tile, -1 / 0 automatic */
#define DSP_DBG_PREPASS_AUDIT 10    /* value != 0: every run also decodes all in-sphere samples in fp32 and fills dsp_stats.prepass_max_err / _misclassified / _audited */
#define DSP_DBG_LP_SMALL_BATCHES 12 /* 1: the low-precision compute mode also on detection-sized batches (automatic: they keep the fp32 latency path, which is faster there) */
#define DSP_DBG_CLUSTER_FAULT 11    /* value != 0: the following runs' cluster launches lose one workgroup's hand-off (the device-side fallback takes over); 0 also ends the cool-down */
int dsp_batch_set_debug(dsp_batch* b, int key, int value);
/* Forensics: front-to-back ranges with explicit depth-index boundaries: bounds[0] = 0 <= ... <= bounds[n_passes] = num_depth_samples. */
int dsp_batch_debug_ray_pass_bounds(dsp_batch* b, const int32_t* bounds, int n_passes);
/* Forensics: start the following runs from the given camera->object matrices (n_objects x 16, used as they are -- no
 * inversion, so a recorded state of the reference can be injected bit for bit) and / or codes (n_objects x 64); NULL t_obj_cam returns
 * to the uploaded object->camera estimates.  depths (optional, n_objects x 64, needs t_obj_cam): the FIRST iteration samples the rays at
 * exactly these num_depth_samples depths instead of deriving them from the pose (optimizer.py:120-125) -- the reference derives them in
 * fp32 LAPACK / powf arithmetic that can differ from this library's by 1-2 ulp, which is enough to move samples across the render term's
 * thresholds. */
int dsp_batch_debug_start_state(dsp_batch* b, const float* t_obj_cam, const float* codes, const float* depths);
/* Forensics: iteration e < n_iterations of the following runs samples the rays at depths[(e * n_objects + i) * 64 ..] instead of
 * the depths derived from the pose (n_iterations = 0 turns the schedule off). */
int dsp_batch_debug_depth_schedule(dsp_batch* b, const float* depths, int32_t n_iterations);
/* Forensics: the per-sample arrays the LAST iteration of the last run left behind for object obj, expanded to
 * (n_rays x num_depth_samples) grids: raymask (n_rays; bit j = sample j lies inside the unit sphere), ssdf (the sdf the occupancy
 * scan read: fp32 inside the band, the prepass value or the placeholder 1.0 elsewhere; NaN = not in the sphere), sdeds (de_ds of a kept
 * sample, 0 = not kept, NaN = not in the sphere).  cap = floats available in ssdf / sdeds (>= n_rays * num_depth_samples). */
int dsp_batch_debug_samples(dsp_batch* b, int32_t obj, uint64_t* raymask, float* ssdf, float* sdeds, int64_t cap);
/* Testing: the Lie-group maps and the rotation prior evaluated ON THE DEVICE by the very functions the solve kernel calls (one thread, same
 * fp32 / fp64 arithmetic), so that every branch can be compared with vectors recorded from the reference:
 *   kind 0: x[7]  -> out[16] = exp_sim3(x)   -- reconstruct/loss_utils.py:188-233 (theta <= 1e-8 / s == 0 branch :211-218, the
 *                                               `c = 0. if s <= eps` quirk :223)
 *   kind 1: x[6]  -> out[16] = exp_se3(x)    -- reconstruct/loss_utils.py:129-163
 *   kind 2: x[16] = t_obj_cam -> out[0..6] = J_rot, out[7] = res_rot (compute_rotation_loss_sim3, reconstruct/loss.py:155-178, zero branch
 *           :172-173), out[8] = scale = det(R_co)^(1/3), out[9], out[10] = the depth range t_z -+ scale (reconstruct/optimizer.py:120-125),
 *           out[11] = status (DSP_OBJ_NAN for a singular matrix: the other entries are then zero)
 *   kind 3: x[0..15] = t_obj_cam, x[16..22] = dx -> out[16] = exp_sim3(dx) @ t_obj_cam  (the update of reconstruct/optimizer.py:187-188)
 * n_depth = num_depth_samples (2..64; only kind 2 reads it). */
int dsp_debug_lie(dsp_handle* h, int kind, const float* x, int32_t n_depth, float* out16);
/* Testing: the (n + 1)-th fresh device allocation the handle's pool makes from now on fails as if HBM were exhausted (n < 0: off).  The call
 * that hits it returns DSP_E_NOMEM with nothing of its half-built batch left allocated; the handle stays usable. */
int dsp_debug_fail_alloc(dsp_handle* h, int n);
/* Record per-iteration traces during the following runs (testing; costs ~21 KB per object-iteration). */
int dsp_batch_enable_trace(dsp_batch* b, int on);
/* Per-iteration trace of the last run (testing): for iteration e < num_iterations and object i,
 * H (71x71), b (71), dx (71), V, m, K, the state the iteration started from, and set_sums[2*i + {0,1}] = order-
 * independent checksums of the in-sphere sample set and of the kept (jacobian) sample set: sum over members of
 * hash(ray << 6 | depth_index), hash(x) = (x * 2654435761) ^ (x >> 7), mod 2^32.  Any pointer may be NULL. */
int dsp_batch_trace(dsp_batch* b, int32_t iteration, float* H, float* bvec, float* dx, int64_t* V, int64_t* m,
                    int64_t* K, float* t_obj_cam, float* code, uint32_t* set_sums, float* depths /* 64 per object */);
void dsp_batch_destroy(dsp_batch* b);

/* ---- multi-GPU: objects are independent, so GPUs take disjoint blocks of the object list (one handle per GPU, one host
 *      thread per handle) and the ONLY exchange is one gather of the per-object results.  The Python mirror does this across processes with
 *      torch.distributed (dsp_slam_amd/distributed.py); these two calls are the same for a single C / C++ process that owns several GPUs. */
#define DSP_RESULT_WIDTH 82   /* t_cam_obj 16 | code 64 | loss | status (as float) */
/* Pack the outputs of dsp_reconstruct_batch / dsp_batch_results into n x DSP_RESULT_WIDTH rows (host only). */
void dsp_pack_results(int32_t n, const float* t_cam_obj, const float* codes, const float* loss, const int32_t* status, float* packed);
/* results[i]: n_objects[i] x DSP_RESULT_WIDTH host floats produced on handles[i] (distinct GPUs).  The blocks are uploaded, gathered
 * device-to-device to handles[0]'s GPU by ONE RCCL ncclGather (xGMI; uneven blocks padded to the largest), and returned in `out`
 * concatenated in handle order (sum(n_objects) x DSP_RESULT_WIDTH).  librccl is loaded on first use (dlopen), not linked. */
int dsp_gather_results(dsp_handle* const* handles, int32_t n_handles, const float* const* results, const int32_t* n_objects, float* out);
/* The same gather straight from device-resident batches (one per GPU, each run before): every batch keeps its results as packed rows in
 * HBM, so the data goes device -> ncclGather -> host ONCE (no host bounce in front of the collective).  out: sum of the batches' object
 * counts x DSP_RESULT_WIDTH, in batch order.  Both gathers hold the mutex of EVERY handle involved (taken in a fixed order) for the whole
 * call: a dsp_batch_run issued on one of the handles by its own host thread waits, it cannot interleave with the collective. */
int dsp_gather_batch_results(dsp_batch* const* batches, int32_t n_batches, float* out);
/* Copy a batch's packed result rows (n_objects x DSP_RESULT_WIDTH) to a DEVICE buffer on the batch's GPU -- e.g. a PyTorch-ROCm tensor's
 * data_ptr() that a torch.distributed collective then sends (dsp_slam_amd/distributed.py: gather_results_device). */
int dsp_batch_results_packed_dev(dsp_batch* b, float* dst_dev);

#ifdef __cplusplus
}
#endif
#endif /* DSP_GN_H */
