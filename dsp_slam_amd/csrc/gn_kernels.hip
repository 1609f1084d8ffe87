// Everything around the decoder in one Gauss-Newton iteration, batched over independent objects:
// ray sampling + in-sphere compaction, occupancy/transmittance scan per ray, J-row assembly,
// 72x72 Gram (normal-equation) reduction, damped solve + Lie-group update.
//
// Reference functions restated on the device (file:line relative to the reference tree):
//   compute_sdf_loss / compute_render_loss / compute_rotation_loss_sim3   reconstruct/loss.py:22-178
//   get_points_to_pose_jacobian_sim3, exp_se3, exp_sim3, huber_norm_weights reconstruct/loss_utils.py:107-265
//   Optimizer.reconstruct_object / estimate_pose_cam_obj                   reconstruct/optimizer.py:45-203
// These kernels are HBM/latency-bound integer+float bookkeeping (< 0.5 % of the FLOPs); they exist so
// that the whole iteration stays on the device with no host round trip.
#include "dsp_internal.h"

#include <algorithm>
#include <cstdio>

namespace dsp {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// (p[..., None, :] * R).sum(-1) + t with fp32 products and adds in the reference's order, no FMA
// contraction, so that in-sphere decisions agree with the oracle bit for bit (loss.py:31-32,62-63).
__device__ __forceinline__ float3 xform(const float* T, float x, float y, float z) {
    float3 o;
    o.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, T[0]), __fmul_rn(y, T[1])), __fmul_rn(z, T[2])), T[3]);
    o.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, T[4]), __fmul_rn(y, T[5])), __fmul_rn(z, T[6])), T[7]);
    o.z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, T[8]), __fmul_rn(y, T[9])), __fmul_rn(z, T[10])), T[11]);
    return o;
}

__device__ bool inv4(const double* a, double* out) {   // Gauss-Jordan with partial pivoting
    // Fully unrolled with compile-time indices only (row swaps are conditional register swaps): a run-time row index would
    // put the 4x8 tableau into scratch memory, i.e. ~150 dependent HBM-latency accesses in a single-thread section
    // (measured: 100 us of k_solve's 137 us).
    double m[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[i][j] = a[4 * i + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(m[c][c]);
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double v = fabs(m[r][c]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0.0) ok = false;
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const bool sw = (p == r);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double x = m[c][j], y = m[r][j];
                m[c][j] = sw ? y : x;
                m[r][j] = sw ? x : y;
            }
        }
        const double inv = 1.0 / m[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[c][j] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r != c) {
                const double f = m[r][c];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * i + j] = m[i][4 + j];
    return ok;
}

__device__ double det3(const double* r, int ld) {
    return r[0] * (r[ld + 1] * r[2 * ld + 2] - r[ld + 2] * r[2 * ld + 1]) -
           r[1] * (r[ld] * r[2 * ld + 2] - r[ld + 2] * r[2 * ld]) +
           r[2] * (r[ld] * r[2 * ld + 1] - r[ld + 1] * r[2 * ld]);
}

// Per-iteration derived quantities of one object (optimizer.py:120-126): T_co = inv(T_oc), scale =
// det(R_co)^(1/3), depth range t_z -+ scale, torch.linspace(d_min, d_max, D) in float32.
// In registers (k_solve runs it in every lane of one wave and stores from lane 0 / one depth per lane: nothing is read back from memory):
struct IterDerived { float t_co[16], scale, dmin, dmax, step; bool ok; };
__device__ __forceinline__ IterDerived derive_iter_core(const float* t_oc, int n_depth) {
    IterDerived r;
    double toc[16], tco[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) toc[i] = (double)t_oc[i];
    r.ok = inv4(toc, tco);
#pragma unroll
    for (int i = 0; i < 16; ++i) r.t_co[i] = (float)tco[i];
    const double det = det3(tco, 4);
    r.scale = (float)cbrt(det);
    const float tz = r.t_co[11];
    r.dmin = tz - 1.0f * r.scale;
    r.dmax = tz + 1.0f * r.scale;
    r.step = (r.dmax - r.dmin) / (float)(n_depth - 1);
    return r;
}
// ATen linspace: first half from start, second half from end
__device__ __forceinline__ float linspace_at(const IterDerived& r, int i, int n_depth) {
    return (i < n_depth / 2) ? __fadd_rn(r.dmin, __fmul_rn(r.step, (float)i)) : __fsub_rn(r.dmax, __fmul_rn(r.step, (float)(n_depth - 1 - i)));
}
__device__ void derive_iter_state(ObjState& s, int n_depth) {
    float toc[16];
    for (int i = 0; i < 16; ++i) toc[i] = s.t_oc[i];
    const IterDerived r = derive_iter_core(toc, n_depth);
    if (!r.ok) { s.status = DSP_STATUS_NAN; return; }
    for (int i = 0; i < 16; ++i) s.t_co[i] = r.t_co[i];
    s.scale = r.scale;
    s.dmin = r.dmin; s.dmax = r.dmax;
    for (int i = 0; i < n_depth; ++i) s.depths[i] = linspace_at(r, i, n_depth);
}

// Prepass margin of an object from the largest entry of its current code (LpDeltaTab: measured per decoder at dsp_create with codes
// drawn at the listed magnitudes).  Piecewise linear, extrapolated above the last magnitude, never below the first entry, capped at
// 0.5 (a band that wide sends every sample to the fp32 kernel); a non-finite code gets the cap.
__device__ __forceinline__ float lp_delta_of(float zmax, const LpDeltaTab& t) {
    if (!(zmax < 1e30f)) return 0.5f;
    float d = t.delta[0];
#pragma unroll
    for (int i = 0; i + 1 < LP_NMAG; ++i) {
        const float m0 = t.mag[i], m1 = t.mag[i + 1];
        if (zmax > m0 && (zmax <= m1 || i + 2 == LP_NMAG)) d = t.delta[i] + (t.delta[i + 1] - t.delta[i]) * ((zmax - m0) / (m1 - m0));
    }
    return fminf(fmaxf(d, t.delta[0]), 0.5f);
}

__global__ void k_init_state(ObjState* st, const float* t_cam_obj, const float* codes, const float* scale_in, const float* depths,
                             int n_obj, int n_depth, int pose_only, LpDeltaTab lp, const unsigned char* run_mask, unsigned* summary, int summary_words) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // the run's counters (work counters, audit words, ...; read back once at the end of the run) start from zero: this is the first kernel
    for (int i = b; i < summary_words; i += gridDim.x * blockDim.x) summary[i] = 0u;
    if (b >= n_obj) return;
    ObjState& s = st[b];
    if (run_mask && !run_mask[b]) { s.status = DSP_STATUS_SKIP; return; }     // partial re-run (batch_run): this object keeps its results
    {
        unsigned* w = reinterpret_cast<unsigned*>(&s);
        for (int i = 0; i < (int)(sizeof(ObjState) / 4); ++i) w[i] = 0u;
    }
    double tco[16], toc[16];
    for (int i = 0; i < 16; ++i) tco[i] = (double)t_cam_obj[16 * b + i];
    if (pose_only & 1) {   // optimizer.py:52-55: R *= scale before inverting
        const double sc = (double)scale_in[b];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) tco[4 * r + c] = (double)(float)(tco[4 * r + c] * sc);
    }
    s.status = DSP_STATUS_GOOD;
    if (pose_only & 2) {          // the input IS the camera->object matrix (dsp_batch_set_start_state): taken as it is
        for (int i = 0; i < 16; ++i) toc[i] = tco[i];
    } else if (!inv4(tco, toc)) { s.status = DSP_STATUS_NAN; for (int i = 0; i < 16; ++i) toc[i] = (i % 5 == 0); }
    pose_only &= 1;
    for (int i = 0; i < 16; ++i) s.t_oc[i] = (float)toc[i];
    for (int i = 0; i < CODE_LEN; ++i) s.code[i] = codes ? codes[CODE_LEN * b + i] : 0.f;
    s.loss = 0.f; s.V = 0; s.m = 0; s.K = 0; s.P = 0; s.n_alive = -1; s.vsum = 0; s.ksum = 0;
    float zmax = 0.f;
    for (int i = 0; i < CODE_LEN; ++i) { const float a = fabsf(s.code[i]); zmax = (a > zmax || a != a) ? a : zmax; }
    s.lp_delta = lp_delta_of(zmax, lp);
    s.guard_trips = 0; s.guard_err = 0;
    if (!pose_only) {
        derive_iter_state(s, n_depth);
        if (depths) {     // forensics: the first iteration samples exactly these depths (dsp_batch_set_start_state)
            for (int i = 0; i < n_depth; ++i) s.depths[i] = depths[MAX_DEPTH_SAMPLES * b + i];
            s.dmin = s.depths[0];
            s.dmax = s.depths[n_depth - 1];
        }
    }
}

// membership checksum of a sample id (ray << 6 | depth index): wrap-around sum of a per-id hash, so two runs agree on it
// iff (up to 2^-32) they selected the same SET of samples -- equal counts alone do not show that (tests compare it with
// the oracle's).
__device__ __forceinline__ unsigned id_hash(unsigned id) { return (id * 2654435761u) ^ (id >> 7); }

// ------------------------------------------------------------------------------------------------
// ray sampling: in-sphere mask + count per ray  (loss.py:60-70)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample_count_ray(const ObjConst& c, ObjState* st, int b, const float* rays, unsigned long long* raymask,
                                                 int* raycnt, int n_depth, int r) {
    const ObjState& s = st[b];
    unsigned long long mask = 0ull;
    if (s.status == DSP_STATUS_GOOD) {
        const float* d3 = rays + 3 * (size_t)(c.ray_off + r);
        const float dx = d3[0], dy = d3[1], dz = d3[2];
        for (int j = 0; j < n_depth; ++j) {
            const float d = s.depths[j];
            const float3 p = xform(s.t_oc, __fmul_rn(dx, d), __fmul_rn(dy, d), __fmul_rn(dz, d));
            const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(p.x, p.x), __fmul_rn(p.y, p.y)), __fmul_rn(p.z, p.z));
            if (__fsqrt_rn(n2) < 1.0f) mask |= 1ull << j;
        }
    }
    raymask[c.ray_off + r] = mask;
    raycnt[c.ray_off + r] = __popcll(mask);
    unsigned h = 0;
    for (unsigned long long m = mask; m; m &= m - 1) h += id_hash(((unsigned)r << 6) | (unsigned)(__ffsll((long long)m) - 1));
    if (h) atomicAdd(&st[b].vsum, h);
}

__global__ void k_sample_count(const ObjConst* oc, ObjState* st, const float* rays, unsigned long long* raymask,
                               int* raycnt, int n_depth) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    sample_count_ray(c, st, b, rays, raymask, raycnt, n_depth, r);
}

// exclusive scan of per-ray counts inside each object (one workgroup per object)
// (NT = threads of the workgroup, a multiple of 64; part = NT ints of LDS.)  Thread t owns a contiguous chunk of rays; chunk sums
// are scanned inside each wave with shuffles and across the <= 16 waves through LDS: three barriers.
template <int NT>
__device__ __forceinline__ void scan_rays_block(const ObjConst& c, ObjState* st, int b, const int* cnt, int* off, int which, int* part) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = NT / 64;
    const int per = (c.n_rays + NT - 1) / NT;
    const int lo = min(tid * per, c.n_rays), hi = min(lo + per, c.n_rays);
    int sum = 0;
    for (int r = lo; r < hi; ++r) sum += cnt[c.ray_off + r];
    int incl = sum;                                 // inclusive scan of the chunk sums inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    __syncthreads();                                // part may still be read by an earlier phase of the caller
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int w = lane < NWV ? part[lane] : 0;
#pragma unroll
        for (int d = 1; d < NWV; d <<= 1) {
            const int v = __shfl_up(w, d);
            if (lane >= d) w += v;
        }
        if (lane < NWV) part[NWV + lane] = w;       // inclusive scan of the wave totals
    }
    __syncthreads();
    int run = incl - sum + (wave ? part[NWV + wave - 1] : 0);
    for (int r = lo; r < hi; ++r) { off[c.ray_off + r] = run; run += cnt[c.ray_off + r]; }
    if (tid == NT - 1) {
        const int total = part[2 * NWV - 1];
        ObjState& s = st[b];
        if (which == 0) {
            s.V = total;
            if (s.status == DSP_STATUS_GOOD && total < 10) s.status = DSP_STATUS_FEW;   // loss.py:73-74
        } else if (which == 1) {
            s.K = total;
        } else {
            s.P = total;
        }
    }
}

__global__ __launch_bounds__(256) void k_scan_rays(const ObjConst* oc, ObjState* st, const int* cnt, int* off, int which) {
    __shared__ int part[256];
    const int b = blockIdx.x;
    const ObjConst c = oc[b];
    scan_rays_block<256>(c, st, b, cnt, off, which, part);
}

// One WAVE per ray, one lane per depth index: a ray's in-sphere samples are contiguous in the compact list, so the wave's stores coalesce
// (one thread per ray wrote 16 bytes at a stride of the ray's length: 149 us per iteration on the bench batch; this form ~40: the step's
// `other` 10.8 -> 9.8 ms, profiles/r06_kernel_stats.md).  Per sample the same arithmetic as before: p_o = T_oc (dir * d) with the products rounded first.
__global__ __launch_bounds__(256) void k_sample_write(const ObjConst* oc, const ObjState* st, const float* rays, const unsigned long long* raymask,
                                                      const int* rayoff, float4* spts, float* ssdf, unsigned char* alive, int n_depth) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (r >= c.n_rays) return;
    const ObjState& s = st[b];
    if (s.status != DSP_STATUS_GOOD) return;
    const int j = threadIdx.x & 63;
    const unsigned long long mask = raymask[c.ray_off + r];
    if (j == 0) alive[c.ray_off + r] = mask ? 1 : 0;
    if (!((mask >> j) & 1ull)) return;
    const float* d3 = rays + 3 * (size_t)(c.ray_off + r);
    const float d = s.depths[j];
    const float3 p = xform(s.t_oc, __fmul_rn(d3[0], d), __fmul_rn(d3[1], d), __fmul_rn(d3[2], d));
    const size_t o = (size_t)c.samp_off + rayoff[c.ray_off + r] + __popcll(mask & ((1ull << j) - 1ull));
    spts[o] = make_float4(p.x, p.y, p.z, __int_as_float((r << 6) | j));
    ssdf[o] = 1.0f;     // "not evaluated": free space (o = 0).  Only samples BEHIND a solid one stay unevaluated, where the
                        // transmittance is exactly 0, so the value cannot reach d_u, de_do, H or b (see k_pass_update).
}

// ------------------------------------------------------------------------------------------------
// front-to-back ray passes (exact early ray termination)
// ------------------------------------------------------------------------------------------------
// occupancy is exactly 1 for sdf <= -th (0.5 + th/(2 th) in fp32), so T_l = prod(1 - o_i) is exactly 0 behind the first
// such sample: every term those samples could contribute to the rendered depth (o_l * T_{l-1}), to any suffix sum of T and
// hence to any kept row is 0 -- their decoder values are never needed.  The forward decoder therefore runs in passes over
// depth-index ranges [j0, j1), front to back, and a ray drops out after the pass in which it first reports a solid sample.
__device__ __forceinline__ unsigned long long range_mask(int j0, int j1) {
    const unsigned long long hi = (j1 >= 64) ? ~0ull : ((1ull << j1) - 1ull);
    return hi & ~((1ull << j0) - 1ull);
}

// Range of ray gr in the current pass.  Fixed mode (hint == nullptr): [j0, j1) for every ray.  Adaptive mode: pass 0 covers
// [0, hint + margin) where hint is the depth index at which this ray terminated in the previous GN iteration (the pose moves
// little between iterations), pass 1 the next `step` indices, the last pass whatever is left -- always contiguous and front
// to back, so exactness does not depend on the hints, only the amount of skipped work does.
__device__ __forceinline__ void pass_range(int gr, int j0, int j1, int n_depth, int pass, int last, const unsigned char* hint,
                                           const unsigned char* plo, int& lo, int& hi) {
    if (!hint) { lo = j0; hi = j1; return; }
    lo = (pass == 0) ? 0 : (int)plo[gr];
    hi = last ? n_depth : (pass == 0 ? min(n_depth, (int)hint[gr] + j0) : min(n_depth, lo + j1));
    hi = max(hi, lo);
}

__global__ void k_pass_select(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff,
                              const unsigned char* alive, int* pcnt, int j0, int j1, int n_depth, int pass, int last,
                              const unsigned char* hint, const unsigned char* plo) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    const int gr = c.ray_off + r;
    const bool on = st[b].status == DSP_STATUS_GOOD && alive[gr];
    int lo, hi;
    pass_range(gr, j0, j1, n_depth, pass, last, hint, plo, lo, hi);
    pcnt[gr] = on ? __popcll(raymask[gr] & range_mask(lo, hi)) : 0;
}

__global__ void k_pass_write(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff,
                             const unsigned char* alive, const int* poff, int* plist, int j0, int j1, int n_depth, int pass, int last,
                             const unsigned char* hint, const unsigned char* plo) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    const int gr = c.ray_off + r;
    if (st[b].status != DSP_STATUS_GOOD || !alive[gr]) return;
    int lo, hi;
    pass_range(gr, j0, j1, n_depth, pass, last, hint, plo, lo, hi);
    const unsigned long long mask = raymask[gr];
    unsigned long long sel = mask & range_mask(lo, hi);
    int* dst = plist + c.samp_off + poff[gr];
    const int base = c.samp_off + rayoff[gr];
    while (sel) {
        const int j = __ffsll((long long)sel) - 1;
        sel &= sel - 1;
        *dst++ = base + __popcll(mask & ((1ull << j) - 1ull));   // position of sample (r, j) in the compact sample list
    }
}

__global__ void k_pass_update(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff,
                              unsigned char* alive, const float* ssdf, float th, int use_lp_delta, int j0, int j1, int n_depth, int pass, int last,
                              unsigned char* hint, unsigned char* plo) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    const int gr = c.ray_off + r;
    if (st[b].status != DSP_STATUS_GOOD || !alive[gr]) return;
    if (use_lp_delta) th += st[b].lp_delta;      // prepass values: only a CERTAINLY solid sample stops the ray
    int lo, hi;
    pass_range(gr, j0, j1, n_depth, pass, last, hint, plo, lo, hi);
    const unsigned long long mask = raymask[gr];
    unsigned long long sel = mask & range_mask(lo, hi);
    const int base = c.samp_off + rayoff[gr];
    int first_solid = -1;
    while (sel) {
        const int j = __ffsll((long long)sel) - 1;
        sel &= sel - 1;
        if (ssdf[base + __popcll(mask & ((1ull << j) - 1ull))] <= -th) { first_solid = j; break; }
    }
    if (first_solid >= 0) alive[gr] = 0;
    if (hint) {
        plo[gr] = (unsigned char)hi;
        if (first_solid >= 0) hint[gr] = (unsigned char)first_solid;
        else if (last) hint[gr] = (unsigned char)n_depth;      // never terminated: decode the whole ray in pass 0 next time
    }
}

// ------------------------------------------------------------------------------------------------
// prepass: which samples still need the fp32 decoder
// ------------------------------------------------------------------------------------------------
// After the low-precision front-to-back passes ssdf holds sdf_lp for every sample they decoded (+1 elsewhere).  With
// |sdf_lp - sdf_fp32| < delta (calibrated margin), sdf_lp >= th + delta means occupancy exactly 0 and sdf_lp <= -(th + delta)
// means occupancy exactly 1 (loss_utils.py:40-48) -- and behind the first such sample of a ray the transmittance is exactly
// 0, so nothing there can reach the result.  What is left for the fp32 kernel: the samples IN FRONT of a ray's first
// certainly-solid sample whose |sdf_lp| < th + delta.  Their exact values then replace the low-precision ones in ssdf; the
// classified samples keep sdf_lp (any value beyond +-th gives the same occupancy bit for bit).
// Guard samples: a sample of the samples the prepass CLASSIFIED -- beyond the widened band, up to and including the ray's first
// certainly-solid sample -- joins the list, so that the fp32 kernel re-decodes it and prepass_guard (mlp_common.h) compares the two
// values.  Its exact value then replaces the prepass value: the same occupancy, bit for bit, whenever the classification was right.
// Stratified by where an error would matter: 1/8 of the RING th + delta <= |sdf_lp| < th + 2 delta (an error between delta and 2 delta
// misclassifies exactly these), 1/512 of everything farther out (there only a gross failure -- overflow, a broken weight stream --
// can misclassify, and a gross failure hits many samples).  Together with the band itself, where EVERY sample is compared, that is
// ~0.3 % of the in-sphere samples on the bench workload.  The id hash is xor-ed with a per-launch salt; salt == 0: no guard samples.
__device__ __forceinline__ bool guard_pick(unsigned id, unsigned salt, bool ring) {
    return salt != 0u && ((id_hash(id) ^ salt) & (ring ? 7u : 511u)) == 0u;
}

__device__ __forceinline__ void band_count_ray(const ObjConst& c, const ObjState& s, const unsigned long long* raymask, const int* rayoff,
                                               const float* ssdf, float th, unsigned salt, int* pcnt, int r) {
    const int gr = c.ray_off + r;
    int n = 0;
    if (s.status == DSP_STATUS_GOOD) {
        const float thd = th + s.lp_delta;
        unsigned long long mask = raymask[gr];
        const float* sd = ssdf + c.samp_off + rayoff[gr];
        for (int i = 0; mask; ++i) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float v = sd[i];
            const bool pick = v != 1.0f && guard_pick(((unsigned)r << 6) | (unsigned)j, salt, fabsf(v) < thd + s.lp_delta);
            n += (!(fabsf(v) >= thd) || pick) ? 1 : 0;      // `!(>=)`, not `<`: a NaN prepass value belongs to the band (the fp32 kernel decides)
            if (v <= -thd) break;
        }
    }
    pcnt[gr] = n;
}

__device__ __forceinline__ void band_write_ray(const ObjConst& c, const ObjState& s, const unsigned long long* raymask, const int* rayoff,
                                               const float* ssdf, float th, unsigned salt, const int* poff, int* plist, int r) {
    const int gr = c.ray_off + r;
    if (s.status != DSP_STATUS_GOOD) return;
    const float thd = th + s.lp_delta;
    unsigned long long mask = raymask[gr];
    const int base = c.samp_off + rayoff[gr];
    int* dst = plist + c.samp_off + poff[gr];
    for (int i = 0; mask; ++i) {
        const int j = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float v = ssdf[base + i];
        const bool pick = v != 1.0f && guard_pick(((unsigned)r << 6) | (unsigned)j, salt, fabsf(v) < thd + s.lp_delta);
        if (!(fabsf(v) >= thd) || pick) *dst++ = base + i;
        if (v <= -thd) break;
    }
}

__global__ void k_band_count(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff,
                             const float* ssdf, float th, unsigned salt, int* pcnt) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    band_count_ray(c, st[b], raymask, rayoff, ssdf, th, salt, pcnt, r);
}

__global__ void k_band_write(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff,
                             const float* ssdf, float th, unsigned salt, const int* poff, int* plist) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.n_rays) return;
    band_write_ray(c, st[b], raymask, rayoff, ssdf, th, salt, poff, plist, r);
}

// audit (tests / calibration): saudit = fp32 sdf of EVERY in-sphere sample, ssdf = prepass values (+1 where not decoded).
// out[0] = max |sdf_lp - sdf_fp32| (float bits), out[1] = samples the prepass classified against the fp32 value
// (sdf_lp >= thd but sdf_fp32 < th, or sdf_lp <= -thd but sdf_fp32 > -th), out[2] = samples compared.
__global__ void k_prepass_audit(const ObjConst* oc, const ObjState* st, const float* ssdf, const float* saudit, float th,
                                unsigned* out) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const ObjState& s = st[b];
    if (s.status != DSP_STATUS_GOOD) return;
    const float thd = th + s.lp_delta;
    float worst = 0.f;
    unsigned bad = 0, n = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.V; i += gridDim.x * blockDim.x) {
        const float lp = ssdf[c.samp_off + i], ex = saudit[c.samp_off + i];
        if (lp == 1.0f) continue;                       // not decoded by the prepass
        ++n;
        worst = fmaxf(worst, fabsf(lp - ex));
        if ((lp >= thd && ex < th) || (lp <= -thd && ex > -th)) ++bad;
    }
    if (n) {
        atomicMax(out + 0, __float_as_uint(worst));
        if (bad) atomicAdd(out + 1, bad);
        atomicAdd(out + 2, n);
    }
}

// surface points -> object frame (loss.py:31-32); also the pose-only inlier bookkeeping (optimizer.py:76-78)
__device__ __forceinline__ void surface_point(const ObjConst& c, const ObjState& s, const float* pts, float4* jpts, float2* jaux, int i) {
    if (s.status != DSP_STATUS_GOOD) return;
    const float* p = pts + 3 * (size_t)(c.pts_off + i);
    const float3 o = xform(s.t_oc, p[0], p[1], p[2]);
    jpts[c.jsdf_off + i] = make_float4(o.x, o.y, o.z, __int_as_float(i));
    jaux[c.jsdf_off + i] = make_float2(1.f, 0.f);
}

__global__ void k_surface(const ObjConst* oc, const ObjState* st, const float* pts, float4* jpts, float2* jaux) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_pts) return;
    surface_point(c, st[b], pts, jpts, jaux, i);
}

// ------------------------------------------------------------------------------------------------
// tile lists for the decoder kernels (single workgroup; counts live on the device)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_tiles(const ObjConst* oc, ObjState* st, int n_obj, int mode, int4* tiles,
                                                     int* n_tiles, double* counters, int add_v, int tile_pts, int cnt_slot, int apply_few) {
    // mode 0: forward tiles over the V in-sphere samples; mode 1: jacobian tiles over M surface + K render points;
    // mode 2: forward tiles over the P samples selected for the current front-to-back pass (indexed through plist)
    // One thread per object (rounds of 256): tile counts, a block-wide exclusive scan for the list offsets, then the four waves
    // fill the objects' tile entries side by side -- same lists, in the same order, as walking the objects one by one.
    __shared__ int s_n[256], s_off[256], s_base[256], s_part[8];
    __shared__ double s_red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double cnt = 0.0, vtot = 0.0, rows = 0.0;       // per-thread partial sums of integers: exact in any order
    int base = 0, n_surface_tiles = 0;
    // mode 1 lists every object's surface tiles first and the render tiles after them, so that the two jacobian launches
    // (forward+backward / backward-only) each take one contiguous range
    const bool jac = mode == 1 || mode == 3;   // mode 3: the band samples (P) stand in for the kept render rows (K): speculative band rows
    for (int phase = 0; phase < (jac ? 2 : 1); ++phase) {
        for (int b0 = 0; b0 < n_obj; b0 += 256) {
            const int b = b0 + tid;
            int n = 0, off = 0;
            if (b < n_obj) {
                const ObjConst c = oc[b];
                const ObjState& s = st[b];
                int status = s.status;
                // wave-per-ray bookkeeping (k_front_wave) counts V with a running counter and leaves the "< 10 in-sphere samples" rule
                // (loss.py:73-74; k_scan_rays applies it in the other forms) to the first tile list built from that count
                if ((apply_few & 1) && !jac && status == DSP_STATUS_GOOD && s.V < 10) { status = DSP_STATUS_FEW; st[b].status = DSP_STATUS_FEW; }
                const bool good = status == DSP_STATUS_GOOD;
                if (!jac) {
                    n = good ? (mode == 0 ? s.V : s.P) : 0;
                    if ((apply_few & 2) && mode == 2) st[b].P = 0;      // a front-to-back pass's count (k_scan_rays): consumed; k_band_wave counts from zero
                    off = c.samp_off;
                    cnt += n;
                    if (good) vtot += s.V;
                } else {
                    n = good ? (phase == 0 ? c.n_pts : (mode == 3 ? s.P : s.K)) : 0;
                    off = phase == 0 ? c.jsdf_off : c.jren_off;
                    if (phase == 0) cnt += n; else rows += n;
                }
            }
            const int nt = (n + tile_pts - 1) / tile_pts;
            int incl = nt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            __syncthreads();                        // the previous round's fill has finished reading s_*
            if (lane == 63) s_part[wave] = incl;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int pw = s_part[w];
                if (w < wave) before += pw;
                total += pw;
            }
            s_n[tid] = n;
            s_off[tid] = off;
            s_base[tid] = base + before + incl - nt;
            __syncthreads();
            const int in_round = min(256, n_obj - b0);
            for (int j = wave; j < in_round; j += 4) {
                const int nj = s_n[j], oj = s_off[j], bj = s_base[j];
                const int ntj = (nj + tile_pts - 1) / tile_pts;
                for (int i = lane; i < ntj; i += 64) tiles[bj + i] = make_int4(oj + i * tile_pts, min(tile_pts, nj - i * tile_pts), b0 + j, 0);
            }
            base += total;
        }
        if (phase == 0) n_surface_tiles = base;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        cnt += __shfl_xor(cnt, d);
        vtot += __shfl_xor(vtot, d);
        rows += __shfl_xor(rows, d);
    }
    if (lane == 0) { s_red[0][wave] = cnt; s_red[1][wave] = vtot; s_red[2][wave] = rows; }
    __syncthreads();
    if (tid == 0) {
        n_tiles[0] = base;
        if (jac) n_tiles[1] = n_surface_tiles;
        counters[jac ? 1 : cnt_slot] += s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        if (add_v) counters[2] += s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (jac) counters[3] += s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
    }
}

// Tail split of a forward tile list (latency-sized batches): a launch of T 64-point tiles on n_cu CUs takes ceil(T / n_cu) rounds of
// one tile time each, and the last round is often nearly empty (one cfg2 object: 312 band tiles = one full round + 56).  When that
// remainder is at most half a round, its tiles are re-listed as 16-point tiles for the latency-form kernel (mlp_split_kernel<false>,
// ~0.3 of a 64-point tile's time, bit-identical results), which runs them in a launch of its own: the 64-point launch loses its
// last round.  tiles16 needs 4 * (n_cu / 2) entries.
__global__ __launch_bounds__(256) void k_tail_tiles(const int4* tiles, int* n_tiles, int4* tiles16, int* n_tiles16, int n_cu) {
    const int T = n_tiles[0];
    int tail = 0;
    if (T > n_cu && T <= 8 * n_cu) {
        const int r = T % n_cu;
        if (r > 0 && r <= n_cu / 2) tail = r;
    }
    __syncthreads();                 // every thread has read T before thread 0 rewrites it
    for (int i = threadIdx.x; i < tail; i += 256) {
        const int4 td = tiles[T - tail + i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // a quarter beyond the tile's points is an empty tile: it keeps the tile's FIRST point as its (unused) address -- lanes without
            // a point still read list entry td.x, and entries behind the tile's last point are not valid indices
            const int n = max(0, min(SPLIT_TILE_PTS, td.y - SPLIT_TILE_PTS * q));
            tiles16[4 * i + q] = make_int4(n > 0 ? td.x + SPLIT_TILE_PTS * q : td.x, n, td.z, td.w);
        }
    }
    if (threadIdx.x == 0) { n_tiles[0] = T - tail; n_tiles16[0] = 4 * tail; }
}

// ------------------------------------------------------------------------------------------------
// per-ray occupancy / transmittance scan  (loss.py:84-141)
// ------------------------------------------------------------------------------------------------
// One thread per ray, the 50-sample row kept in registers.  Pass 1 (count): occupancy o_j, T_l =
// prod_{i<=l}(1-o_i), rendered depth d_u, suffix sums for de_do, keeps samples with |sdf| < th and
// de_do > 1e-2; stores de_ds per compact sample (0 = dropped), d_u per ray and the kept count.
// One WAVE per ray, lane = depth index: everything per-sample (occupancy, the two divisions, the keep tests, the compaction) runs in
// parallel across the lanes; only the three chains whose rounding order the reference fixes -- cumprod of (1 - o), the rendered-depth
// sum and the suffix sums of T, all front to back resp. back to front in depth order -- are walked with v_readlane, one lane at a
// time.  ~600 wave instructions per ray instead of ~7000 thread instructions, and every ray on its own SIMD.
__device__ __forceinline__ float lane_value(float x, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j)); }

__global__ __launch_bounds__(256) void k_render_scan(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff,
                              const float* ssdf, const float* depth_fg, float* sdeds, float* ray_res, int* kcnt, int* mcnt,
                              int n_depth, float th) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // wave-uniform
    if (r >= c.n_rays) return;
    const ObjState& s = st[b];
    const int gr = c.ray_off + r;
    if (s.status != DSP_STATUS_GOOD) { if (lane == 0) { kcnt[gr] = 0; mcnt[gr] = 0; } return; }
    const unsigned long long mask = raymask[gr];
    const int base = c.samp_off + rayoff[gr];
    const bool in = lane < n_depth && ((mask >> lane) & 1ull);
    const int k = __popcll(mask & ((1ull << lane) - 1ull));
    const float sd = in ? ssdf[base + k] : 0.f;
    const float dj = s.depths[lane < n_depth ? lane : 0];
    float oj = 0.f;
    bool wg = false;                     // with_grad: -th < sdf < th  (loss.py:88)
    if (in) {
        const float cl = fminf(fmaxf(sd, -th), th);
        oj = __fsub_rn(0.5f, __fdiv_rn(cl, __fmul_rn(2.f, th)));   // sdf_to_occupancy (loss_utils.py:40-48)
        wg = sd > -th && sd < th;
    }
    const float f = __fsub_rn(1.f, oj);  // 1 exactly where there is no sample
    // T_l = prod_{i<=l} (1 - o_i), sequential in l (torch.cumprod)
    float Tj = 0.f, acc = 1.f;
    for (int j = 0; j < n_depth; ++j) {
        acc = __fmul_rn(acc, lane_value(f, j));
        if (lane == j) Tj = acc;
    }
    const float d_bg = __fmul_rn(1.1f, s.depths[n_depth - 1]);
    // d_u = sum_l d_l * o_l * T_{l-1} + d_bg * T_{D-1}, summed front to back   (loss.py:100-114)
    const float tprev = __shfl_up(Tj, 1);
    const float term = __fmul_rn(dj, __fmul_rn(oj, lane == 0 ? 1.f : tprev));
    float du = 0.f;
    for (int j = 0; j < n_depth; ++j) du = __fadd_rn(du, lane_value(term, j));
    du = __fadd_rn(du, __fmul_rn(d_bg, lane_value(Tj, n_depth - 1)));
    const float obs = (r < c.n_fg) ? depth_fg[c.depth_off + r] : d_bg;   // optimizer.py:126
    float res = __fsub_rn(obs, du);
    // loss.py:139-140 as written there -- `res_d[res_d > 0.30] = 0.30; res_d[res_d < -0.30] = -0.30` -- NOT fminf / fmaxf: a NaN residual (a NaN
    // observed depth) compares false twice and stays NaN, the render loss turns NaN and the object fails (optimizer.py:149-150); fminf / fmaxf
    // would turn it into a valid 0.30 (found by tests/test_gpu_errors.py).  Identical bits for every finite or infinite residual.
    res = res > 0.30f ? 0.30f : res;
    res = res < -0.30f ? -0.30f : res;
    // de_do_k = sum_{l>=k} T_l / (1 - o_k), the suffix sums accumulated back to front; keep > 1e-2;
    // de_ds = de_do * delta_d * (-1/(2 th))  (loss.py:118-130)
    float Sj = 0.f, sacc = 0.f;
    for (int j = n_depth - 1; j >= 0; --j) {
        sacc = __fadd_rn(sacc, lane_value(Tj, j));
        if (lane == j) Sj = sacc;
    }
    const float delta_d = __fdiv_rn(__fsub_rn(s.depths[n_depth - 1], s.depths[0]), (float)(n_depth - 1));
    const float do_ds = __fdiv_rn(-1.f, __fmul_rn(2.f, th));
    float deds = 0.f;
    bool keep = false;
    if (wg) {
        const float dedo = __fdiv_rn(Sj, __fsub_rn(1.f, oj));
        if (dedo > 1e-2f) { deds = __fmul_rn(__fmul_rn(dedo, delta_d), do_ds); keep = true; }
    }
    if (in) sdeds[base + k] = deds;      // never exactly 0 for a kept sample (dedo > 0.01, delta_d > 0)
    const unsigned long long keptm = __ballot(keep), wgm = __ballot(wg);
    unsigned khash = keep ? id_hash(((unsigned)r << 6) | (unsigned)lane) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) khash += __shfl_xor(khash, d);
    if (lane == 0) {
        ray_res[gr] = res;
        kcnt[gr] = __popcll(keptm);
        mcnt[gr] = __popcll(wgm);
        if (khash) atomicAdd(&st[b].ksum, khash);
    }
}

// Row compaction: the kept samples of a ray (those with de_ds != 0: typically 2-4 of its <= 50) become jacobian rows, in depth order behind
// the rows of the rays in front (koff).  One WAVE per ray, lane = position in the ray's compact sample run: one coalesced load finds the kept
// ones, a ballot places them.
__global__ __launch_bounds__(256) void k_render_write(const ObjConst* oc, const ObjState* st, const int* raycnt, const int* rayoff, const int* koff,
                               const float4* spts, const float* sdeds, const float* ray_res, float4* jpts, float2* jaux) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (r >= c.n_rays) return;
    if (st[b].status != DSP_STATUS_GOOD) return;
    const int k = threadIdx.x & 63;
    const int gr = c.ray_off + r;
    const int n = raycnt[gr];
    const int base = c.samp_off + rayoff[gr];
    const float dv = k < n ? sdeds[base + k] : 0.f;
    const bool kept = k < n && dv != 0.f;
    const unsigned long long keptm = __ballot(kept);
    if (!kept) return;
    const int dst = c.jren_off + koff[gr] + __popcll(keptm & ((1ull << k) - 1ull));
    float4 p = spts[base + k];
    p.w = __int_as_float(base + k);   // compact sample index: where the forward launch left this sample's sdf and relu masks
    jpts[dst] = p;
    jaux[dst] = make_float2(dv, ray_res[gr]);
}

// ------------------------------------------------------------------------------------------------
// wave-per-ray forms of the same three stages (latency path): sampling + compaction (+ surface points), band selection, row compaction, one
// launch each.  A ray is a wave (lane = depth index, as in k_render_scan), 16 rays a workgroup, and the rays of an object spread over as many
// CUs as they fill (a per-object fused form -- one workgroup per object -- cost a detection 27 + 41 + 22 us per iteration on ONE CU:
// profiles/r06_removed_experiments.md).  Two of the throughput form's three scans over the rays are not needed at all:
//   * the in-sphere sample list and the band / speculative-row list are INTERNAL orders -- every consumer goes through rayoff / plist /
//     srow / jrow, each point's decoder result is independent of the tile it shares (test_decode_is_tile_independent), and the Gram
//     kernel walks the kept rows in koff order -- so their segments are handed out by one atomicAdd per workgroup (ObjState::V / ::P);
//   * the kept-row order IS the Gram summation order, so k_render_tail_wave keeps it: every workgroup sums the counts of the rays in
//     front of its own (<= 2500 integers) instead of waiting for a scan launch.
// Same per-sample arithmetic, same sets, same H / b / dx bits as the throughput form (test_wave_bookkeeping_is_exact).
// ------------------------------------------------------------------------------------------------
constexpr int WAVE_RAYS = 16;            // rays (= waves) per workgroup of k_front_wave / k_band_wave
constexpr int WAVE_THREADS = 64 * WAVE_RAYS;

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// per-workgroup segment of `tot` list slots from an object's running counter: s_cnt[w] = wave w's count; returns this wave's first slot
__device__ __forceinline__ int wave_segment(int* counter, int* s_cnt, int* s_base, int wave, int cnt, int lane) {
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < WAVE_RAYS; ++w) tot += s_cnt[w];
        *s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    return off;
}

// k_sample_count + k_sample_write + k_surface (the scan is replaced by the running counter ObjState::V, zero at the start of an
// iteration; the "< 10 samples" rule of loss.py:73-74 is applied by the tile builder that follows, k_build_tiles apply_few)
__global__ __launch_bounds__(WAVE_THREADS) void k_front_wave(const ObjConst* oc, ObjState* st, const float* __restrict__ rays, const float* pts,
                                                             unsigned long long* raymask, int* raycnt, int* rayoff, float4* spts, float* ssdf,
                                                             unsigned char* alive, float4* jpts, float2* jaux, int n_depth, int n_ray_blocks) {
    __shared__ int s_cnt[WAVE_RAYS], s_base;
    __shared__ unsigned s_hash[WAVE_RAYS];
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= n_ray_blocks) {           // the surface points, 1024 per workgroup
        const int i = ((int)blockIdx.x - n_ray_blocks) * WAVE_THREADS + (int)threadIdx.x;
        if (i < c.n_pts) surface_point(c, st[b], pts, jpts, jaux, i);
        return;
    }
    if ((int)blockIdx.x * WAVE_RAYS >= c.n_rays) return;       // workgroup-uniform
    const int r = blockIdx.x * WAVE_RAYS + wave;
    const bool live = r < c.n_rays && st[b].status == DSP_STATUS_GOOD;
    bool in = false;
    float3 p = make_float3(0.f, 0.f, 0.f);
    if (live && lane < n_depth) {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = st[b].t_oc[i];
        const float dj = st[b].depths[lane];
        const float* d3 = rays + 3 * (size_t)(c.ray_off + r);
        p = xform(T, __fmul_rn(d3[0], dj), __fmul_rn(d3[1], dj), __fmul_rn(d3[2], dj));
        const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(p.x, p.x), __fmul_rn(p.y, p.y)), __fmul_rn(p.z, p.z));
        in = __fsqrt_rn(n2) < 1.0f;
    }
    const unsigned long long mask = __ballot(in);
    const int cnt = __popcll(mask);
    unsigned h = in ? id_hash(((unsigned)r << 6) | (unsigned)lane) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d);
    if (lane == 0) s_hash[wave] = h;
    const int off = wave_segment(&st[b].V, s_cnt, &s_base, wave, cnt, lane);
    if (threadIdx.x == 0) {
        unsigned hs = 0;
#pragma unroll
        for (int w = 0; w < WAVE_RAYS; ++w) hs += s_hash[w];
        if (hs) atomicAdd(&st[b].vsum, hs);
    }
    if (r >= c.n_rays) return;
    const int gr = c.ray_off + r;
    if (lane == 0) { raymask[gr] = mask; raycnt[gr] = cnt; rayoff[gr] = off; alive[gr] = mask ? 1 : 0; }
    if (in) {
        const int k = __popcll(mask & lanes_below(lane));
        spts[c.samp_off + off + k] = make_float4(p.x, p.y, p.z, __int_as_float((r << 6) | lane));
        ssdf[c.samp_off + off + k] = 1.0f;       // "not evaluated": free space (see k_sample_write)
    }
}

// k_band_count + k_band_write (+ the speculative band rows): the selection of band_count_ray / band_write_ray with lane = depth
// index; list slots from the running counter ObjState::P (zero at the start of an iteration)
__global__ __launch_bounds__(WAVE_THREADS) void k_band_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff,
                                                            const float* ssdf, float th, unsigned salt, int* plist, const float4* spts, float4* jpts,
                                                            int* srow) {
    __shared__ int s_cnt[WAVE_RAYS], s_base;
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    if ((int)blockIdx.x * WAVE_RAYS >= c.n_rays) return;       // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * WAVE_RAYS + wave;
    const bool live = r < c.n_rays && st[b].status == DSP_STATUS_GOOD;
    bool sel = false;
    int idx = 0;
    if (live) {
        const float delta = st[b].lp_delta, thd = th + delta;
        const int gr = c.ray_off + r;
        const unsigned long long rmask = raymask[gr];
        const bool in = (rmask >> lane) & 1ull;
        idx = c.samp_off + rayoff[gr] + __popcll(rmask & lanes_below(lane));
        const float v = in ? ssdf[idx] : 1.0f;
        const bool solid = in && v <= -thd;
        const bool band = in && !(fabsf(v) >= thd);                 // `!(>=)`: a NaN prepass value belongs to the band
        const bool decoded = in && v != 1.0f;
        const bool ring = fabsf(v) < thd + delta;
        const unsigned long long solidm = __ballot(solid);
        const int first = solidm ? __ffsll((long long)solidm) - 1 : 64;     // samples behind the first certainly-solid one are skipped
        // guard samples (band_count_ray): classified samples up to AND INCLUDING the first certainly-solid one
        const bool pick = decoded && guard_pick(((unsigned)r << 6) | (unsigned)lane, salt, ring);
        sel = (band || pick) && (salt ? lane <= first : lane < first);
    }
    const unsigned long long selm = __ballot(sel);
    const int off = wave_segment(&st[b].P, s_cnt, &s_base, wave, __popcll(selm), lane);
    if (sel) {
        const int pos = off + __popcll(selm & lanes_below(lane));
        plist[c.samp_off + pos] = idx;
        if (jpts) {       // speculative band rows: the sample goes straight into the jacobian launch (forward + backward), row jren_off + pos
            float4 p = spts[idx];
            p.w = __int_as_float(idx);
            jpts[c.jren_off + pos] = p;
            srow[idx] = c.jren_off + pos;
        }
    }
}

// k_scan_rays(1) + k_sum_m + k_render_write behind k_render_scan: 64 rays per workgroup (16 waves x 4); the rows keep ray-major,
// depth-minor order (the reference's row order = the Gram summation order)
constexpr int TAIL_RAYS = 64;

__global__ __launch_bounds__(WAVE_THREADS) void k_render_tail_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff,
                                                                   const float4* spts, const float* sdeds, const float* ray_res, const int* kcnt,
                                                                   const int* mcnt, float4* jpts, float2* jaux, const int* srow, int* jrow) {
    __shared__ int part[3][WAVE_RAYS];
    __shared__ int s_koff[TAIL_RAYS];
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int r0 = blockIdx.x * TAIL_RAYS;
    if (blockIdx.x != 0 && r0 >= c.n_rays) return;             // workgroup-uniform; workgroup 0 always runs (it owns K and m)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // kept rows in front of this workgroup's first ray; workgroup 0: the object's totals K and m
    int pre = 0, ktot = 0, mtot = 0;
    for (int i = tid; i < c.n_rays; i += WAVE_THREADS) {
        const int kc = kcnt[c.ray_off + i];
        if (i < r0) pre += kc;
        if (blockIdx.x == 0) { ktot += kc; mtot += mcnt[c.ray_off + i]; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { pre += __shfl_xor(pre, d); ktot += __shfl_xor(ktot, d); mtot += __shfl_xor(mtot, d); }
    if (lane == 0) { part[0][wave] = pre; part[1][wave] = ktot; part[2][wave] = mtot; }
    if (wave == 0) {       // exclusive scan of this workgroup's 64 per-ray counts
        const int kc = (r0 + lane < c.n_rays) ? kcnt[c.ray_off + r0 + lane] : 0;
        int incl = kc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        s_koff[lane] = incl - kc;
    }
    __syncthreads();
    pre = 0; ktot = 0; mtot = 0;
#pragma unroll
    for (int w = 0; w < WAVE_RAYS; ++w) { pre += part[0][w]; ktot += part[1][w]; mtot += part[2][w]; }
    const int status = st[b].status;
    // (an object that is not part of a partial re-run keeps its state untouched -- K and m included: launch_init_state, run_mask)
    if (blockIdx.x == 0 && tid == 0 && status != DSP_STATUS_SKIP) { st[b].K = ktot; st[b].m = mtot; }
    if (status != DSP_STATUS_GOOD) return;
    for (int q = 0; q < TAIL_RAYS / WAVE_RAYS; ++q) {
        const int rl = wave * (TAIL_RAYS / WAVE_RAYS) + q, r = r0 + rl;
        if (r >= c.n_rays) break;                               // wave-uniform
        const int gr = c.ray_off + r;
        const unsigned long long rmask = raymask[gr];
        const bool in = (rmask >> lane) & 1ull;
        const int sidx = c.samp_off + rayoff[gr] + __popcll(rmask & lanes_below(lane));
        const float dv = in ? sdeds[sidx] : 0.f;
        const bool kept = in && dv != 0.f;                      // de_ds is never exactly 0 for a kept sample (k_render_scan)
        const unsigned long long keptm = __ballot(kept);
        if (kept) {
            const int dst = c.jren_off + pre + s_koff[rl] + __popcll(keptm & lanes_below(lane));
            float4 p = spts[sidx];
            p.w = __int_as_float(sidx);       // compact sample index: where the forward launch left this sample's sdf and relu masks
            jpts[dst] = p;
            jaux[dst] = make_float2(dv, ray_res[gr]);
            if (jrow) jrow[dst] = srow[sidx]; // speculative band rows: this row's gradient already sits in jgrad row srow[sample]
        }
    }
}

__global__ __launch_bounds__(256) void k_sum_m(const ObjConst* oc, ObjState* st, const int* mcnt) {
    __shared__ int part[256];
    const int b = blockIdx.x;
    const ObjConst c = oc[b];
    int sum = 0;
    for (int r = threadIdx.x; r < c.n_rays; r += 256) sum += mcnt[c.ray_off + r];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) st[b].m = part[0];
}

// ------------------------------------------------------------------------------------------------
// J rows and the 72x72 Gram reduction  (loss.py:36-41,143-150; loss_utils.py:166-185,236-265; optimizer.py:159-167)
// ------------------------------------------------------------------------------------------------
// row = [ J_pose(7) | J_code(64) | r~ ]:  J^T J is H, -J^T r~ is b and r~^2 sums to the loss, so one
// 72x72 Gram matrix per (object, term) carries everything the solve needs.
__device__ __forceinline__ void build_row(const float* g68, float4 p, float2 aux, int term, float huber_b, int robust,
                                          float* row /*72, stride 1*/) {
    const float sc = aux.x;                       // de_ds (render) or 1 (sdf)
    const float r = (term == 0) ? g68[67] : aux.y;   // sdf term: residual is the sdf itself (loss.py:34,43)
    const float d0 = __fmul_rn(sc, g68[64]), d1 = __fmul_rn(sc, g68[65]), d2 = __fmul_rn(sc, g68[66]);
    row[0] = d0; row[1] = d1; row[2] = d2;
    // [I | -[p]x | p]  (loss_utils.py:166-185)
    row[3] = __fadd_rn(__fmul_rn(-p.z, d1), __fmul_rn(p.y, d2));
    row[4] = __fadd_rn(__fmul_rn(p.z, d0), __fmul_rn(-p.x, d2));
    row[5] = __fadd_rn(__fmul_rn(-p.y, d0), __fmul_rn(p.x, d1));
    row[6] = __fadd_rn(__fadd_rn(__fmul_rn(p.x, d0), __fmul_rn(p.y, d1)), __fmul_rn(p.z, d2));
    for (int i = 0; i < 64; ++i) row[7 + i] = __fmul_rn(sc, g68[i]);
    float rr = r;
    if (robust) {   // huber_norm_weights: w = sqrt(rho)/|r|, only the residual is reweighted (loss_utils.py:236-265)
        const float a = fabsf(r);
        const float rho = (a <= huber_b) ? __fmul_rn(a, a) : __fsub_rn(__fmul_rn(__fmul_rn(2.f, huber_b), a), __fmul_rn(huber_b, huber_b));
        const float w = (a == 0.f) ? 0.f : __fdiv_rn(__fsqrt_rn(rho), a);
        rr = __fmul_rn(w, r);
    }
    row[71] = rr;
}

constexpr int GRAM_PTS = 32;    // granularity of a slice's share of the rows (part of the summation grouping: do not change)
constexpr int GRAM_SUB = 4;     // row groups of GRAM_PTS staged per barrier pair: the loads of all of them are in flight together
constexpr int JLD = 73;         // LDS row stride (odd: conflict-free column access)

__global__ __launch_bounds__(256) void k_gram(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux,
                                              const float* jgrad, const int* jrow, const unsigned char* alive, float* partials, int n_slices,
                                              float b_sdf, float b_render, int robust) {
    __shared__ float J[GRAM_SUB * GRAM_PTS * JLD];
    const int slice = blockIdx.x, b = blockIdx.y, term = blockIdx.z;
    const ObjConst c = oc[b];
    const ObjState& s = st[b];
    const int n = (s.status != DSP_STATUS_GOOD) ? 0 : (term == 0 ? c.n_pts : s.K);
    const int off = (term == 0) ? c.jsdf_off : c.jren_off;
    const int per = ((n + n_slices - 1) / n_slices + GRAM_PTS - 1) / GRAM_PTS * GRAM_PTS;
    const int lo = min(slice * per, n), hi = min(lo + per, n);
    const int tid = threadIdx.x;
    // thread owns a 4 x 6 block of the 72 x 72 Gram matrix (18 x 12 = 216 active threads)
    const int bi = (tid / 12) * 4, bj = (tid % 12) * 6;
    const bool active = tid < 216;
    float acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = 0.f;
    const float hb = (term == 0) ? b_sdf : b_render;
    for (int p0 = lo; p0 < hi; p0 += GRAM_SUB * GRAM_PTS) {
        const int np = min(GRAM_SUB * GRAM_PTS, hi - p0);
        __syncthreads();
        {   // stage up to 128 rows: 8 threads per row (the 64 code columns in eighths; the first of them also the 7 pose columns and the
            // residual) -- build_row's arithmetic, spread over the workgroup.  A detection-sized slice (<= 128 rows) is ONE round of
            // loads, all in flight together, instead of one dependent global round trip per 32 rows (14 -> ~7 us per iteration).
            const int r0 = tid >> 3, part = tid & 7;
            float4 ga[GRAM_SUB], gb[GRAM_SUB], gx[GRAM_SUB], pp[GRAM_SUB];
            float2 aux[GRAM_SUB];
            bool live[GRAM_SUB];
#pragma unroll
            for (int q = 0; q < GRAM_SUB; ++q) {
                const int r = r0 + GRAM_PTS * q;
                live[q] = r < np && (alive == nullptr || term != 0 || alive[off + p0 + r]);
                ga[q] = gb[q] = gx[q] = pp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                aux[q] = make_float2(0.f, 0.f);
                if (live[q]) {
                    const int idx = off + p0 + r;
                    const int gi = (term == 1 && jrow) ? jrow[idx] : idx;     // speculative band rows: the gradient stays where the launch wrote it
                    const float* g68 = jgrad + (size_t)gi * GRAD_STRIDE;
                    aux[q] = jaux[idx];
                    ga[q] = *reinterpret_cast<const float4*>(g68 + 8 * part);
                    gb[q] = *reinterpret_cast<const float4*>(g68 + 8 * part + 4);
                    if (part == 0) {
                        gx[q] = *reinterpret_cast<const float4*>(g68 + 64);      // d sdf / d xyz, sdf
                        pp[q] = jpts[idx];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < GRAM_SUB; ++q) {
                const int r = r0 + GRAM_PTS * q;
                if (r >= (np + GRAM_PTS - 1) / GRAM_PTS * GRAM_PTS) continue;      // row groups beyond this stage's rows are not read below
                float* row = J + r * JLD;
                float* rc = row + 7 + 8 * part;
                if (live[q]) {
                    const float sc = aux[q].x;                                    // de_ds (render) or 1 (sdf)
                    rc[0] = __fmul_rn(sc, ga[q].x); rc[1] = __fmul_rn(sc, ga[q].y); rc[2] = __fmul_rn(sc, ga[q].z); rc[3] = __fmul_rn(sc, ga[q].w);
                    rc[4] = __fmul_rn(sc, gb[q].x); rc[5] = __fmul_rn(sc, gb[q].y); rc[6] = __fmul_rn(sc, gb[q].z); rc[7] = __fmul_rn(sc, gb[q].w);
                    if (part == 0) {
                        const float4 p = pp[q];
                        const float res = (term == 0) ? gx[q].w : aux[q].y;   // sdf term: residual is the sdf itself (loss.py:34,43)
                        const float d0 = __fmul_rn(sc, gx[q].x), d1 = __fmul_rn(sc, gx[q].y), d2 = __fmul_rn(sc, gx[q].z);
                        row[0] = d0; row[1] = d1; row[2] = d2;
                        // [I | -[p]x | p]  (loss_utils.py:166-185)
                        row[3] = __fadd_rn(__fmul_rn(-p.z, d1), __fmul_rn(p.y, d2));
                        row[4] = __fadd_rn(__fmul_rn(p.z, d0), __fmul_rn(-p.x, d2));
                        row[5] = __fadd_rn(__fmul_rn(-p.y, d0), __fmul_rn(p.x, d1));
                        row[6] = __fadd_rn(__fadd_rn(__fmul_rn(p.x, d0), __fmul_rn(p.y, d1)), __fmul_rn(p.z, d2));
                        float rr = res;
                        if (robust) {   // huber_norm_weights: w = sqrt(rho)/|r|, only the residual is reweighted (loss_utils.py:236-265)
                            const float a = fabsf(res);
                            const float rho = (a <= hb) ? __fmul_rn(a, a) : __fsub_rn(__fmul_rn(__fmul_rn(2.f, hb), a), __fmul_rn(hb, hb));
                            const float w = (a == 0.f) ? 0.f : __fdiv_rn(__fsqrt_rn(rho), a);
                            rr = __fmul_rn(w, res);
                        }
                        row[71] = rr;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) rc[i] = 0.f;
                    if (part == 0) {
#pragma unroll
                        for (int i = 0; i < 7; ++i) row[i] = 0.f;
                        row[71] = 0.f;
                    }
                }
            }
        }
        __syncthreads();
        if (active) {
            // rows in order; a stage's tail up to the next multiple of GRAM_PTS is zero rows (as every stage of 32 had them before)
            const int nr = (np + GRAM_PTS - 1) / GRAM_PTS * GRAM_PTS;
            for (int p = 0; p < nr; ++p) {
                const float* row = J + p * JLD;
                float a[4], bb[6];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = row[bi + i];
#pragma unroll
                for (int j = 0; j < 6; ++j) bb[j] = row[bj + j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
            }
        }
    }
    if (active) {
        float* out = partials + (((size_t)b * 2 + term) * n_slices + slice) * (72 * 72);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) out[(bi + i) * 72 + bj + j] = acc[i][j];
    }
}

// J rows to memory, for the stand-alone compute_sdf_loss / compute_render_loss entry points
__global__ void k_jrows(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux, const float* jgrad,
                        const int* jrow, int term, float* rows /*[n][72]*/) {
    const ObjConst c = oc[0];
    const int n = (term == 0) ? c.n_pts : st[0].K;
    const int off = (term == 0) ? c.jsdf_off : c.jren_off;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float row[72];
    const int gi = (term == 1 && jrow) ? jrow[off + i] : off + i;
    build_row(jgrad + (size_t)gi * GRAD_STRIDE, jpts[off + i], jaux[off + i], term, 0.f, 0, row);
    for (int k = 0; k < 72; ++k) rows[(size_t)i * 72 + k] = row[k];
}

// ------------------------------------------------------------------------------------------------
// solve + update  (optimizer.py:153-192, 68-78; loss.py:155-178; loss_utils.py:129-233)
// ------------------------------------------------------------------------------------------------
__device__ void exp_so3_parts(const float* w, float& theta, float wh[9], float wh2[9]) {
    wh[0] = 0.f;   wh[1] = -w[2]; wh[2] = w[1];
    wh[3] = w[2];  wh[4] = 0.f;   wh[5] = -w[0];
    wh[6] = -w[1]; wh[7] = w[0];  wh[8] = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float a = 0.f;
            for (int k = 0; k < 3; ++k) a += wh[3 * i + k] * wh[3 * k + j];
            wh2[3 * i + j] = a;
        }
    theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
}

// float32 exp / sin / cos as the reference's CPU torch (Sleef u10) returns them: the correctly rounded value for 99 % / 95 % / 95 % of arguments
// (measured over 2e5 arguments each, tools/make_golden_lie.py) -- here the float64 function rounded once.  It matters because exp_sim3 forms
// c = (e^s - 1) / s, which amplifies the last bit of e^s by 1 / s: with s = 1e-3 one ulp of e^s moves the translation update by 6e-5 relative.
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float sin_cr(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cos_cr(float x) { return (float)cos((double)x); }

__device__ void exp_sim3_dev(const float* x, float* out /*16*/) {   // loss_utils.py:188-233, quirks kept
    const float* v = x; const float* w = x + 3; const float s = x[6];
    float theta, wh[9], wh2[9];
    exp_so3_parts(w, theta, wh, wh2);
    const float t2 = theta * theta;
    const float sn = sin_cr(theta), cs = cos_cr(theta);
    const float es = exp_cr(s);
    const float s2 = s * s;
    float ew[9], j[9];
    const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta <= 1e-8f) {
        const float c = (s == 0.f) ? 1.f : (es - 1.f) / s;
        for (int i = 0; i < 9; ++i) { ew[i] = eye[i]; j[i] = c * eye[i]; }
    } else {
        const float a = es * sn, bq = es * cs;
        const float c = (s <= 1e-8f) ? 0.f : (es - 1.f) / s;     // :223 -- drops the c*I term for s <= 1e-8
        const float k1 = (a * s + (1.f - bq) * theta) / (s2 + t2);
        const float k2 = c - ((bq - 1.f) * s + a * theta) / (s2 + t2);
        for (int i = 0; i < 9; ++i) {
            ew[i] = eye[i] + wh[i] * sn / theta + wh2[i] * (1.f - cs) / t2;
            j[i] = c * eye[i] + k1 * wh[i] / theta + k2 * wh2[i] / t2;
        }
    }
    for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[4 * r + c] = es * ew[3 * r + c];
        out[4 * r + 3] = j[3 * r] * v[0] + j[3 * r + 1] * v[1] + j[3 * r + 2] * v[2];
    }
}

__device__ void exp_se3_dev(const float* x, float* out /*16*/) {    // loss_utils.py:129-163
    const float* v = x; const float* w = x + 3;
    float theta, wh[9], wh2[9];
    exp_so3_parts(w, theta, wh, wh2);
    float ew[9], j[9];
    const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta <= 1e-8f) {
        for (int i = 0; i < 9; ++i) { ew[i] = eye[i]; j[i] = eye[i]; }
    } else {
        const float sn = sin_cr(theta), cs = cos_cr(theta);
        const float t2 = theta * theta, t3 = t2 * theta;
        const float k1 = (1.f - cs) / t2, k2 = (theta - sn) / t3;
        for (int i = 0; i < 9; ++i) {
            ew[i] = eye[i] + wh[i] * sn / theta + wh2[i] * (1.f - cs) / t2;
            j[i] = eye[i] + k1 * wh[i] + k2 * wh2[i];
        }
    }
    for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[4 * r + c] = ew[3 * r + c];
        out[4 * r + 3] = j[3 * r] * v[0] + j[3 * r + 1] * v[1] + j[3 * r + 2] * v[2];
    }
}

// compute_rotation_loss_sim3 (loss.py:155-178): J (7) and residual from T_oc
__device__ void rotation_prior(const float* t_co, float scale, float* jrot, float& res);
__device__ void rotation_prior(const ObjState& s, float* jrot, float& res) {
    float t_co[12];
    for (int i = 0; i < 12; ++i) t_co[i] = s.t_co[i];
    rotation_prior(t_co, s.scale, jrot, res);
}
// (on values: k_solve loads them with everything else it reads, before it waits for any of it)
__device__ void rotation_prior(const float* t_co, float scale, float* jrot, float& res) {
    double rco[9];
    const double sc = (double)scale;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rco[3 * r + c] = (double)(float)((double)t_co[4 * r + c] / sc);
    // r_oc = inv(r_co); for a (near-)rotation this is the adjugate / det
    const double det = det3(rco, 3);
    double roc[9];
    roc[0] = (rco[4] * rco[8] - rco[5] * rco[7]) / det; roc[1] = (rco[2] * rco[7] - rco[1] * rco[8]) / det; roc[2] = (rco[1] * rco[5] - rco[2] * rco[4]) / det;
    roc[3] = (rco[5] * rco[6] - rco[3] * rco[8]) / det; roc[4] = (rco[0] * rco[8] - rco[2] * rco[6]) / det; roc[5] = (rco[2] * rco[3] - rco[0] * rco[5]) / det;
    roc[6] = (rco[3] * rco[7] - rco[4] * rco[6]) / det; roc[7] = (rco[1] * rco[6] - rco[0] * rco[7]) / det; roc[8] = (rco[0] * rco[4] - rco[1] * rco[3]) / det;
    // ry = r_co e_y (column 1); res = 1 - ry . n_g, n_g = (0,-1,0)
    const float ry1 = (float)rco[4];
    res = 1.f - (-ry1);
    for (int i = 0; i < 7; ++i) jrot[i] = 0.f;
    if (res < 1e-7f) { res = 0.f; return; }
    // J[3:6] = (r_oc n_g) x e_y ; r_oc n_g = -column 1 of r_oc
    const float a0 = (float)(-roc[1]), a2 = (float)(-roc[7]);
    // a x e_y = (a1*0 - a2*1, a2*0 - a0*0, a0*1 - a1*0) = (-a2, 0, a0)
    jrot[3] = -a2; jrot[4] = 0.f; jrot[5] = a0;
}

constexpr int NSOLVE = 71;

// per-slice Gram partials -> one fp64 Gram matrix per (object, term); fixed summation order
__global__ __launch_bounds__(256) void k_gram_reduce(const ObjState* st, const float* partials, int n_slices, double* gsum) {
    const int b = blockIdx.y, term = blockIdx.z;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 72 * 72 || st[b].status != DSP_STATUS_GOOD) return;
    const float* p = partials + (((size_t)b * 2 + term) * n_slices) * (72 * 72) + e;
    double a = 0.0;
#pragma unroll 8
    for (int sl = 0; sl < n_slices; ++sl) a += (double)p[(size_t)sl * 72 * 72];
    gsum[((size_t)b * 2 + term) * (72 * 72) + e] = a;
}

constexpr int SOLVE_THREADS = 1024;   // 16 waves: assembly, trace and the code bias use all of them; the elimination nine
constexpr int NS1 = NSOLVE + 1;       // rows of the augmented system: the unknowns + the right-hand side as row n

// 1 / d to full double precision without the IEEE division sequence (it sits on the factorisation's critical path, once per pivot):
// v_rcp_f64 (>= 25 bits) + two Newton steps
__device__ __forceinline__ double fast_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// The 71 x 71 (pose-only: 6 x 6) normal equations in fp64, pivot-free: rows in lanes, eight columns per wave, one barrier per panel of eight
// pivots.  (Earlier forms -- one barrier per pivot, packed LDL^T, Gauss-Jordan on 16 waves -- and their measurements: profiles/r06_removed_experiments.md.)
template <bool V> struct BoolC { static constexpr bool value = V; };

__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(const ObjConst* oc, ObjState* st, const double* gsum, GnParamsDev prm, int iter,
                                                         const float* codew, const float* cb0, const float* cblat, float* cbias,
                                               float* trace /*nullable*/, const float* depths_next /*nullable: forensics*/, int n_obj) {
    __shared__ double A[NS1][NS1 + 1];          // [H | b] in rows 0..n-1 (b = column n); b is also kept as ROW n (rows 64 .. 71 are one register of the elimination)
    const int b = blockIdx.x, tid = threadIdx.x;
    const ObjConst c = oc[b];
    ObjState& s = st[b];
    // Everything this launch reads from global memory goes out HERE, before the first of it is waited for -- the status word included
    // (k_solve is a chain of latencies: status -> Gram loads -> state for the prior -> ... was three round trips of ~1 us each).
    const int status = s.status, K = s.K;
    float pr_tco[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) pr_tco[i] = s.t_co[i];
    const float pr_scale = s.scale;
    // the pose and the code this iteration started from, parked in LDS for the update at the end (visible after the assembly's barrier)
    __shared__ float s_toc0[16], s_code0[CODE_LEN];
    const float park = tid < 16 ? s.t_oc[tid] : (tid >= 64 && tid < 64 + CODE_LEN) ? s.code[tid - 64] : 0.f;
    // The 72 x 72 Gram matrices of the two terms, reduced over the Gram kernel's slices by k_gram_reduce.  (Summing the per-slice partials HERE
    // -- one launch and one kernel boundary less per iteration -- was measured and lost: 786 KB of partials through ONE CU's load path take longer
    // than the 41-workgroup reduce kernel and the boundary together, profiles/r06_removed_experiments.md.)
    const double* G0p = gsum + ((size_t)b * 2 + 0) * (72 * 72);
    const double* G1p = gsum + ((size_t)b * 2 + 1) * (72 * 72);
    auto gram = [=](int term, int idx) -> double { return (term ? G1p : G0p)[idx]; };
    const int M = c.n_pts;
    const int pd = prm.pose_only ? 6 : 7;
    const int n = prm.pose_only ? 6 : NSOLVE;
    // thread (tr, tc) fills column tc of rows tr, tr+12, ... of the joint system: its Gram loads
    const int tr0 = tid / (NSOLVE + 1), j = tid % (NSOLVE + 1);
    constexpr int RG = 12, RPT = (NSOLVE + RG - 1) / RG;
    double g0[RPT], g1[RPT];
    float zq[RPT];
    if (!prm.pose_only) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tr0 + RG * q;
            const bool live = tr0 < RG && i < n;
            const int col = (j < n) ? j : 71;            // the augmented column is b = -J^T r~ (Gram column 71)
            const int gi = live ? i * 72 + col : 0;
            g0[q] = gram(0, gi);
            g1[q] = gram(1, gi);
            zq[q] = s.code[min(max(i - pd, 0), CODE_LEN - 1)];             // the right-hand side's k3 z term (optimizer.py:172); unconditional: a
                                                                           // load in a branch is waited for at the branch's end
        }
    }
    const double g_loss0 = prm.pose_only ? 0.0 : gram(0, 71 * 72 + 71), g_loss1 = prm.pose_only ? 0.0 : gram(1, 71 * 72 + 71);
    if (status != DSP_STATUS_GOOD) return;
    if (tid < 16) s_toc0[tid] = park;
    if (tid >= 64 && tid < 64 + CODE_LEN) s_code0[tid - 64] = park;
    if (!prm.pose_only) {
        // losses (optimizer.py:134-155): mean of robust residual^2; an empty set gives NaN in the reference
        if (M == 0 || K == 0) { if (tid == 0) s.status = DSP_STATUS_NAN; return; }
        const float sdf_loss = (float)g_loss0 / (float)M;
        const float ren_loss = (float)g_loss1 / (float)K;
        if (isnan(sdf_loss) || isnan(ren_loss)) { if (tid == 0) s.status = DSP_STATUS_NAN; return; }
        if (tid == 0) s.loss = prm.k1 * ren_loss + prm.k2 * sdf_loss;
        float jrot[7], res_rot;
        rotation_prior(pr_tco, pr_scale, jrot, res_rot);
        // only entries 3 and 5 of the prior's jacobian are non-zero; selects instead of a run-time index keep it out of scratch
        const float jrot3 = jrot[3], jrot5 = jrot[5];
        auto jr = [=](int i) { return i == 3 ? jrot3 : (i == 5 ? jrot5 : 0.f); };
        const double w_s = (double)prm.k2 / (double)M, w_r = (double)prm.k1 / (double)K;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tr0 + RG * q;
            if (!(tr0 < RG && i < n)) continue;
            double v;
            if (j < n) {
                v = w_s * g0[q] + w_r * g1[q];                                                 // :161-168
                if (i >= pd && i == j) v += (double)prm.k3;                                    // :170
                // code entries beyond the decoder's code length (32-D codes in the 64-wide state) have a zero jacobian: their rows are
                // k3 on the diagonal and 0 on the right -- pinned to the identity so that they stay decoupled (dx = 0) even with k3 = 0
                if (i >= pd + prm.code_len && i == j) v = 1.0;
                if (i < pd && j < pd) v += (double)prm.k4 * (double)jr(i) * (double)jr(j);       // :176,178
                if (i < pd && i == j) v += 1.0;                                                // :183
                if (i == pd - 1 && j == pd - 1) v += (double)prm.s_damp;                       // :184
            } else {
                v = -(w_s * g0[q] + w_r * g1[q]);                                              // b = -J^T r~
                if (i >= pd) v -= (double)prm.k3 * (double)zq[q];                              // :172
                if (i >= pd + prm.code_len) v = 0.0;
                if (i < pd) v += (double)prm.k4 * (double)jr(i) * (double)res_rot;             // :177,179 (sign as written)
            }
            A[i][j] = v;
        }
    } else {
        // pose-only (optimizer.py:68-72): H = J6^T J6 / M + 1e-2 I, b = -J6^T r / M with the raw residual
        const int Ma = (s.n_alive >= 0) ? s.n_alive : M;
        if (Ma == 0) { if (tid == 0) s.status = DSP_STATUS_NAN; return; }
        for (int e = tid; e < n * (n + 1); e += SOLVE_THREADS) {
            const int i = e / (n + 1), j = e % (n + 1);
            double v;
            if (j < n) { v = gram(0, i * 72 + j) / (double)Ma; if (i == j) v += 1e-2; }
            else v = -gram(0, i * 72 + 71) / (double)Ma;
            A[i][j] = v;
        }
    }
    __syncthreads();
    if (trace) {   // [iter][obj][71*71 H | 71 b | 71 dx | 16 t_oc | 64 code | V m K]
        float* tr = trace + ((size_t)iter * n_obj + b) * TRACE_STRIDE;
        for (int e = tid; e < n * n; e += SOLVE_THREADS) tr[(e / n) * NSOLVE + (e % n)] = (float)A[e / n][e % n];
        for (int e = tid; e < n; e += SOLVE_THREADS) tr[NSOLVE * NSOLVE + e] = (float)A[e][n];
        if (tid < 16) tr[NSOLVE * NSOLVE + 2 * NSOLVE + tid] = s.t_oc[tid];
        if (tid < 64) tr[NSOLVE * NSOLVE + 2 * NSOLVE + 16 + tid] = s.code[tid];
        if (tid < 64) tr[5280 + tid] = s.depths[tid];
        if (tid == 0) {
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 80] = (float)s.V;
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 81] = (float)s.m;
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 82] = (float)s.K;
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 83] = (float)(s.vsum & 0xffffu);
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 84] = (float)(s.vsum >> 16);
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 85] = (float)(s.ksum & 0xffffu);
            tr[NSOLVE * NSOLVE + 2 * NSOLVE + 86] = (float)(s.ksum >> 16);
        }
        __syncthreads();
    }
    {
        // 2. H dx = b by elimination in fp64.  H = sum w J^T J + positive diagonal is symmetric (bit for bit: the Gram kernel's fmaf chains
        //    commute) positive definite, so no pivoting is needed (the reference inverts H with fp32 LU, optimizer.py:186).  The right-hand side
        //    rides along as column n (and row n) of the augmented matrix; rows above the pivot are eliminated too, so there is no back substitution:
        //    after step n - 1 column n holds d_i dx_i.
        __shared__ double rdv[NS1];
        __shared__ int s_sing;
        if (tid < n) A[n][tid] = A[tid][n];           // b as row n
        if (tid == 0) s_sing = 0;                     // raised by the panel waves
        {
        // Rows in lanes: wave w < 9 owns
        // columns 8w .. 8w+7 and lane l is row l: v0[jj] = A[l][8w+jj]; rows 64 .. 71 (seven code unknowns and the right-hand side) sit
        // in one more register, vx = A[64 + (l & 7)][8w + (l >> 3)].  A step costs a wave THREE LDS reads (its rows' entries of column
        // k for both register sets, and the per-lane column entry of vx); the pivot and the eight column entries c_jk are rows of the
        // same column, i.e. other lanes' values of the register just read: v_readlane into scalar operands of the FMAs.  Waves whose
        // columns are all finished skip the step.  Values computed for columns <= k are never read again.
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // w in a scalar register: lane selects below are scalar
        const bool worker = w < 9;
        double* Af = &A[0][0];
        constexpr int LDA = NS1 + 1;
        const int o0 = lane * LDA, oxr = (64 + (lane & 7)) * LDA, oxc = (8 * w + (lane >> 3)) * LDA;
        __syncthreads();
        double v0[8], vx = 0.0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v0[jj] = worker ? Af[o0 + 8 * w + jj] : 0.0;
        if (worker) vx = Af[oxr + 8 * w + (lane >> 3)];
        bool sing = false;
        auto readlane_f64 = [](double x, int l) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
        };
        {
        // Panel schedule.  Every a_ij sees v -= (c_ik rd_k) c_jk for k ascending (tests/test_solve_schedule.py emulates the schedule lane for lane),
        // with ONE barrier per EIGHT pivots.  Wave kb owns
        // columns 8kb .. 8kb+7 whole (rows in lanes), so it can run the eight steps of its panel on its own registers: the pivot and the
        // column entries c_jk are lanes of the register that IS column k (v_readlane), and only the eight extra rows (vx) need the LDS --
        // the wave's own write, read back in order, off the pivot chain.  It publishes each column as it becomes final, with the
        // pivot's reciprocal.  After the barrier the waves to its right apply the eight steps in one burst (LDS reads issued up front).
        // Measured with shader-clock stamps (profiles/r05_latency_kernel_stats.md; the stamp patch: profiles/r06_removed_experiments.patch): panel 2600 cycles, barrier 160,
        // burst of the next panel's wave 2150 -> 18 us for the 71 pivots against 25 with one barrier per pivot (860 cycles each).  Both
        // phases are issue-bound, ~8 cycles per instruction for a lone wave of fp64 FMAs and v_readlane pairs (~40 instructions per
        // step each): raising the next panel wave's priority, bursts of 2 or 8 steps, changed nothing.
        const bool active = worker && 8 * w <= n;                         // pose-only (n = 6): wave 0 alone
#pragma unroll 1
        for (int kb = 0; 8 * kb < n; ++kb) {
            if (active && w == kb) {
                auto panel = [&](auto last_c) {
                    constexpr bool LAST = decltype(last_c)::value;        // wave 8: rows / pivots 64 .. 70 live in vx
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int k = 8 * kb + t;
                        if (k < n) {                                      // uniform
                            Af[o0 + k] = v0[t];                           // column k is final: publish it (rows 0 .. 63, then rows 64 .. 71)
                            if ((lane >> 3) == t) Af[oxr + k] = vx;
                            const bool open = t < 7 || LAST;              // a column of mine is still open
                            double cix = 0.0, cjx = 0.0;
                            // rows 64 .. 71 of column k and rows 8w .. 8w+7 come back from the LDS: OTHER LANES' writes of this wave.  The
                            // LDS serves a wave's instructions in order, but to the compiler lanes are unrelated threads (without the
                            // fences it hoisted the non-writing lanes' read above the write): wave-scope release / acquire, no instruction
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if (open) { cix = Af[oxr + k]; cjx = Af[oxc + k]; }
                            __builtin_amdgcn_sched_barrier(0);
                            // the pivot chain: nothing in it waits for the LDS (except in wave 8, whose column entries are rows of vx)
                            const double d = LAST ? readlane_f64(vx, 9 * t) : readlane_f64(v0[t], k);
                            const double rdk = fast_recip(d);
                            sing = sing || !(d > 0.0);                    // also NaN
                            if (open) {
                                const double l0 = lane == k ? 0.0 : v0[t] * rdk;
#pragma unroll
                                for (int jj = t + 1; jj < 8; ++jj)
                                    v0[jj] = fma(-l0, LAST ? readlane_f64(cix, jj) : readlane_f64(v0[t], 8 * kb + jj), v0[jj]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (lane == 0) rdv[k] = rdk;
                            if (open) {
                                const double lx = (64 + (lane & 7)) == k ? 0.0 : cix * rdk;
                                vx = fma(-lx, cjx, vx);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                if (kb == 8) panel(BoolC<true>{}); else panel(BoolC<false>{});
                if (sing && lane == 0) s_sing = 1;
            }
            __syncthreads();                                              // panel kb and its reciprocals are published
            if (active && w > kb) {
                // every column of the panel exists here (8 kb + 7 < 8 w <= n) and every pivot row is one of rows 0 .. 63: straight-line
                // code, in two half-bursts of four steps whose 16 LDS reads are all in flight before the first use
                const int cbase = (w == 8) ? 0 : 8 * w;
                constexpr int BS = 4;       // (8: 125 registers, no faster; 2: slower)
#pragma unroll
                for (int h = 0; h < 8 / BS; ++h) {
                    double ci0[BS], cix[BS], cjx[BS], rk[BS];
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        const int k = 8 * kb + BS * h + t;
                        ci0[t] = Af[o0 + k]; cix[t] = Af[oxr + k]; cjx[t] = Af[oxc + k]; rk[t] = rdv[k];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < BS; ++t) {
                        const int k = 8 * kb + BS * h + t;
                        const double csrc = (w == 8) ? cix[t] : ci0[t];
                        const double l0 = lane == k ? 0.0 : ci0[t] * rk[t], lx = cix[t] * rk[t];
                        // the eight column entries into eight scalar pairs first, then the eight FMAs (one pair reused eight times
                        // serialises v_readlane -> FMA through the scalar write: 2550 -> 2150 cycles per burst)
                        double cj[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) cj[jj] = readlane_f64(csrc, cbase + jj);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) v0[jj] = fma(-l0, cj[jj], v0[jj]);
                        __builtin_amdgcn_sched_barrier(0);
                        vx = fma(-lx, cjx[t], vx);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        }
        __syncthreads();
        // dx_i = A[i][n] / d_i: column n is register n & 7 of wave n >> 3 (n = 71: wave 8, v0[7] and the vx lanes of column 7; n = 6: wave 0, v0[6])
        if (w == (n >> 3)) {
            const double bn = n == NSOLVE ? v0[NSOLVE & 7] : v0[6];
            if (lane < n) Af[o0 + n] = bn * rdv[lane];
            if ((lane >> 3) == (n & 7) && 64 + (lane & 7) < n) Af[oxr + n] = vx * rdv[64 + (lane & 7)];
        }
        }
        __syncthreads();
        if (s_sing) { if (tid == 0) s.status = DSP_STATUS_NAN; return; }      // uniform
        __syncthreads();
    }
    if (trace) {
        float* tr = trace + ((size_t)iter * n_obj + b) * TRACE_STRIDE;
        for (int e = tid; e < n; e += SOLVE_THREADS) tr[NSOLVE * NSOLVE + NSOLVE + e] = (float)A[e][n];
    }
    // 3. update (optimizer.py:187-192 / 73-74).  Wave 0 carries the pose (one lane: exp map, 4x4 product, the next iteration's derived
    //    state -- a few us of serial fp64), wave 1 the code, and waves 2.. the next iteration's code bias once the code is in LDS: the
    //    serial pose work no longer sits in front of the bias loop.
    __shared__ float zc[CODE_LEN];
    if (!prm.pose_only && tid >= 64 && tid < 64 + CODE_LEN) {
        const int i = tid - 64;
        const float zv = s_code0[i] + prm.lr * (float)A[pd + i][n];
        s.code[i] = zv;
        zc[i] = zv;
        // the prepass margin follows the code (this wave holds all CODE_LEN = 64 entries)
        float zmax = fabsf(zv);
        bool bad = zv != zv;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { zmax = fmaxf(zmax, __shfl_xor(zmax, d)); bad = bad || __shfl_xor((int)bad, d); }
        if (i == 0) s.lp_delta = lp_delta_of(bad ? __int_as_float(0x7f800000) : zmax, prm.lp);
    }
    __syncthreads();
    if (tid < 64) {
        // the whole of wave 0 runs the serial pose arithmetic (a wave costs what a lane costs), so that lane 0 stores the matrices and
        // lane i the i-th depth sample; the old pose and code were parked in LDS by the assembly: no global round trip on this path
        float dx[7], dT[16], nt[16];
        for (int i = 0; i < pd; ++i) dx[i] = (prm.pose_only ? 1.f : prm.lr) * (float)A[i][n];
        if (prm.pose_only) exp_se3_dev(dx, dT); else exp_sim3_dev(dx, dT);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += dT[4 * r + k] * s_toc0[4 * k + cc];
                nt[4 * r + cc] = acc;
            }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s.t_oc[i] = nt[i];
            s.vsum = 0; s.ksum = 0;
            s.V = 0; s.P = 0;      // the wave-per-ray bookkeeping counts into these (k_front_wave, k_band_wave); the scans of the other forms overwrite them
        }
        if (!prm.pose_only) {
            const IterDerived r = derive_iter_core(nt, prm.n_depth);
            if (!r.ok) {
                if (tid == 0) s.status = DSP_STATUS_NAN;
            } else {
                if (tid == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s.t_co[i] = r.t_co[i];
                    s.scale = r.scale;
                }
                // forensics (dsp_batch_set_depth_schedule): the next iteration samples exactly these depths
                const float* dn = depths_next ? depths_next + MAX_DEPTH_SAMPLES * b : nullptr;
                if (tid < prm.n_depth) s.depths[tid] = dn ? dn[tid] : linspace_at(r, tid, prm.n_depth);
                if (tid == 0) {
                    s.dmin = dn ? dn[0] : r.dmin;
                    s.dmax = dn ? dn[prm.n_depth - 1] : r.dmax;
                }
            }
        }
    }
    // 4. the next iteration's per-object code bias (k_code_bias: same k-ordered fmaf chains), by the waves that are not busy with the pose
    if (!prm.pose_only && cbias && tid >= 128) {
        for (int e = tid - 128; e < 2 * WIDTH; e += SOLVE_THREADS - 128) {
            const int which = e / WIDTH, o = e % WIDTH;
            float acc = which == 0 ? cb0[o] : cblat[o];
            const float* w = codew + (size_t)e * CODE_LEN;
            for (int cidx = 0; cidx < CODE_LEN; ++cidx) acc = fmaf(w[cidx], zc[cidx], acc);
            cbias[(size_t)b * 2 * WIDTH + e] = acc;
        }
    }
}

// pose-only inlier filter at e == 4 (optimizer.py:76-78): keep |r| <= 0.05 for the following iterations
__global__ void k_inlier_filter(const ObjConst* oc, ObjState* st, const float* jgrad, unsigned char* alive) {
    const int b = blockIdx.y;
    const ObjConst c = oc[b];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_pts) return;
    const int idx = c.jsdf_off + i;
    const bool keep = alive[idx] && fabsf(jgrad[(size_t)idx * GRAD_STRIDE + 67]) <= 0.05f;
    alive[idx] = keep ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_count_alive(const ObjConst* oc, ObjState* st, const unsigned char* alive) {
    __shared__ int part[256];
    const int b = blockIdx.x;
    const ObjConst c = oc[b];
    int sum = 0;
    for (int i = threadIdx.x; i < c.n_pts; i += 256) sum += alive[c.jsdf_off + i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) st[b].n_alive = part[0];
}

// final result: T_co = inv(T_oc) (optimizer.py:200; 81-84 for pose-only, which also divides the scale out), as ONE DSP_RESULT_WIDTH row per
// object (t_cam_obj 16 | code 64 | loss | status): what dsp_batch_results unpacks and what the multi-GPU gather sends, device-resident.
// guard_out (optional): the always-on prepass guard's per-object words {lp_delta, trips, max error bits}, packed for the run's ONE read-back.
__global__ void k_finalize(ObjState* st, const float* scale_in, int n_obj, int pose_only, float* out_packed, unsigned* guard_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_obj) return;
    const ObjState& s = st[b];
    if (s.status == DSP_STATUS_SKIP) return;      // partial re-run: the row of the earlier run stands
    double toc[16], tco[16];
    for (int i = 0; i < 16; ++i) toc[i] = (double)s.t_oc[i];
    if (!inv4(toc, tco)) for (int i = 0; i < 16; ++i) tco[i] = nan("");
    float* row = out_packed + (size_t)DSP_RESULT_WIDTH_DEV * b;
    for (int i = 0; i < 16; ++i) {
        float v = (float)tco[i];
        if (pose_only && (i % 4) < 3 && i < 12) v = v / scale_in[b];
        row[i] = v;
    }
    for (int i = 0; i < CODE_LEN; ++i) row[16 + i] = s.code[i];
    row[80] = s.loss;
    row[81] = (float)s.status;
    if (guard_out) {
        guard_out[3 * b + 0] = __float_as_uint(s.lp_delta);
        guard_out[3 * b + 1] = s.guard_trips;
        guard_out[3 * b + 2] = s.guard_err;
    }
}

// per-object code contribution to layer 0 and to the latent_in layer (one workgroup per object):
// out[b][0][o] = b0[o] + W0[o, :64] . code_b,  out[b][1][o] = b_lat[o] + W_lat[o, code cols] . code_b  (k-ordered fmaf chain)
__global__ __launch_bounds__(256) void k_code_bias(const float* codew, const float* b0, const float* blat, const float* codes,
                                                   int code_stride, float* out) {
    __shared__ float z[CODE_LEN];
    const int b = blockIdx.x;
    if (threadIdx.x < CODE_LEN) z[threadIdx.x] = codes[(size_t)b * code_stride + threadIdx.x];
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * WIDTH; e += 256) {
        const int which = e / WIDTH, o = e % WIDTH;
        float acc = which == 0 ? b0[o] : blat[o];
        const float* w = codew + (size_t)e * CODE_LEN;
        for (int c = 0; c < CODE_LEN; ++c) acc = fmaf(w[c], z[c], acc);
        out[(size_t)b * 2 * WIDTH + e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (used by dsp_gn.hip)
// ------------------------------------------------------------------------------------------------
#define GRID2(n, B) dim3((unsigned)std::max(1, ((n) + 255) / 256), (unsigned)(B))

void launch_code_bias(const float* codew, const float* b0, const float* blat, const float* codes, int code_stride, float* out, int n_obj, hipStream_t s) {
    hipLaunchKernelGGL(k_code_bias, dim3(n_obj), dim3(256), 0, s, codew, b0, blat, codes, code_stride, out);
}
void launch_init_state(ObjState* st, const float* t, const float* codes, const float* scale, const float* depths, int B, int D, int pose_only,
                       const LpDeltaTab& lp, const unsigned char* run_mask, unsigned* summary, int summary_words, hipStream_t s) {
    hipLaunchKernelGGL(k_init_state, dim3((B + 63) / 64), dim3(64), 0, s, st, t, codes, scale, depths, B, D, pose_only, lp, run_mask, summary, summary_words);
}
void launch_sample_count(const ObjConst* oc, ObjState* st, const float* rays, unsigned long long* m, int* c, int D, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_sample_count, GRID2(maxR, B), dim3(256), 0, s, oc, st, rays, m, c, D);
}
void launch_scan_rays(const ObjConst* oc, ObjState* st, const int* cnt, int* off, int which, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_rays, dim3(B), dim3(256), 0, s, oc, st, cnt, off, which);
}
void launch_sample_write(const ObjConst* oc, const ObjState* st, const float* rays, const unsigned long long* m, const int* off, float4* spts,
                         float* ssdf, unsigned char* alive, int D, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_sample_write, dim3((unsigned)std::max(1, (maxR + 3) / 4), (unsigned)B), dim3(256), 0, s, oc, st, rays, m, off, spts, ssdf, alive, D);
}
void launch_pass_select(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const unsigned char* alive,
                        int* pcnt, const PassSpec& ps, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_pass_select, GRID2(maxR, B), dim3(256), 0, s, oc, st, raymask, rayoff, alive, pcnt, ps.j0, ps.j1, ps.n_depth, ps.pass,
                       ps.last, ps.hint, ps.plo);
}
void launch_pass_write(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff, const unsigned char* alive,
                       const int* poff, int* plist, const PassSpec& ps, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_pass_write, GRID2(maxR, B), dim3(256), 0, s, oc, st, raymask, rayoff, alive, poff, plist, ps.j0, ps.j1, ps.n_depth,
                       ps.pass, ps.last, ps.hint, ps.plo);
}
void launch_pass_update(const ObjConst* oc, const ObjState* st, const unsigned long long* raymask, const int* rayoff, unsigned char* alive,
                        const float* ssdf, float th, int use_lp_delta, const PassSpec& ps, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_pass_update, GRID2(maxR, B), dim3(256), 0, s, oc, st, raymask, rayoff, alive, ssdf, th, use_lp_delta, ps.j0, ps.j1, ps.n_depth,
                       ps.pass, ps.last, ps.hint, ps.plo);
}
void launch_band_select(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float* ssdf, float th,
                        unsigned guard_salt, int* pcnt, int* poff, int* plist, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_band_count, GRID2(maxR, B), dim3(256), 0, s, oc, st, raymask, rayoff, ssdf, th, guard_salt, pcnt);
    hipLaunchKernelGGL(k_scan_rays, dim3(B), dim3(256), 0, s, oc, st, pcnt, poff, 2);
    hipLaunchKernelGGL(k_band_write, GRID2(maxR, B), dim3(256), 0, s, oc, st, raymask, rayoff, ssdf, th, guard_salt, poff, plist);
}
void launch_prepass_audit(const ObjConst* oc, const ObjState* st, const float* ssdf, const float* saudit, float th, unsigned* out,
                          int B, hipStream_t s) {
    hipLaunchKernelGGL(k_prepass_audit, dim3(64, B), dim3(256), 0, s, oc, st, ssdf, saudit, th, out);
}
void launch_front_wave(const ObjConst* oc, ObjState* st, const float* rays, const float* pts, unsigned long long* raymask, int* raycnt, int* rayoff,
                       float4* spts, float* ssdf, unsigned char* alive, float4* jpts, float2* jaux, int D, int maxR, int maxM, int B, hipStream_t s) {
    const int nrb = std::max(1, (maxR + WAVE_RAYS - 1) / WAVE_RAYS), nsb = (maxM + WAVE_THREADS - 1) / WAVE_THREADS;
    hipLaunchKernelGGL(k_front_wave, dim3((unsigned)(nrb + nsb), (unsigned)B), dim3(WAVE_THREADS), 0, s, oc, st, rays, pts, raymask, raycnt, rayoff, spts, ssdf, alive,
                       jpts, jaux, D, nrb);
}
void launch_band_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float* ssdf, float th, unsigned guard_salt,
                      int* plist, const float4* spts, float4* jpts, int* srow, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_band_wave, dim3((unsigned)std::max(1, (maxR + WAVE_RAYS - 1) / WAVE_RAYS), (unsigned)B), dim3(WAVE_THREADS), 0, s, oc, st, raymask, rayoff,
                       ssdf, th, guard_salt, plist, spts, jpts, srow);
}
void launch_render_tail_wave(const ObjConst* oc, ObjState* st, const unsigned long long* raymask, const int* rayoff, const float4* spts, const float* sdeds,
                             const float* ray_res, const int* kcnt, const int* mcnt, float4* jpts, float2* jaux, const int* srow, int* jrow, int maxR, int B,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_render_tail_wave, dim3((unsigned)std::max(1, (maxR + TAIL_RAYS - 1) / TAIL_RAYS), (unsigned)B), dim3(WAVE_THREADS), 0, s, oc, st, raymask,
                       rayoff, spts, sdeds, ray_res, kcnt, mcnt, jpts, jaux, srow, jrow);
}
void launch_surface(const ObjConst* oc, const ObjState* st, const float* pts, float4* jpts, float2* jaux, int maxM, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_surface, GRID2(maxM, B), dim3(256), 0, s, oc, st, pts, jpts, jaux);
}
void launch_build_tiles(const ObjConst* oc, ObjState* st, int B, int mode, int4* tiles, int* n_tiles, double* counters, int add_v, int tile_pts,
                        int cnt_slot, hipStream_t s, int apply_few) {
    hipLaunchKernelGGL(k_build_tiles, dim3(1), dim3(256), 0, s, oc, st, B, mode, tiles, n_tiles, counters, add_v, tile_pts, cnt_slot, apply_few);
}
void launch_tail_tiles(const int4* tiles, int* n_tiles, int4* tiles16, int* n_tiles16, int n_cu, hipStream_t s) {
    hipLaunchKernelGGL(k_tail_tiles, dim3(1), dim3(256), 0, s, tiles, n_tiles, tiles16, n_tiles16, n_cu);
}
void launch_render_scan(const ObjConst* oc, ObjState* st, const unsigned long long* m, const int* off, const float* ssdf, const float* depth,
                        float* sdeds, float* ray_res, int* kcnt, int* mcnt, int D, float th, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_render_scan, dim3((unsigned)std::max(1, (maxR + 3) / 4), (unsigned)B), dim3(256), 0, s, oc, st, m, off, ssdf, depth, sdeds, ray_res,
                       kcnt, mcnt, D, th);     // one wave per ray
}
void launch_render_write(const ObjConst* oc, const ObjState* st, const int* raycnt, const int* rayoff, const int* koff, const float4* spts,
                         const float* sdeds, const float* ray_res, float4* jpts, float2* jaux, int maxR, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_render_write, dim3((unsigned)std::max(1, (maxR + 3) / 4), (unsigned)B), dim3(256), 0, s, oc, st, raycnt, rayoff, koff, spts, sdeds, ray_res,
                       jpts, jaux);     // one wave per ray
}
void launch_sum_m(const ObjConst* oc, ObjState* st, const int* mcnt, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_m, dim3(B), dim3(256), 0, s, oc, st, mcnt);
}
void launch_gram(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux, const float* jgrad, const int* jrow,
                 const unsigned char* alive, float* partials, int n_slices, float b_sdf, float b_render, int robust, int n_terms, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_gram, dim3(n_slices, B, n_terms), dim3(256), 0, s, oc, st, jpts, jaux, jgrad, jrow, alive, partials, n_slices, b_sdf, b_render, robust);
}
void launch_jrows(const ObjConst* oc, const ObjState* st, const float4* jpts, const float2* jaux, const float* jgrad, const int* jrow, int term, float* rows,
                  int cap, hipStream_t s) {
    hipLaunchKernelGGL(k_jrows, dim3((cap + 255) / 256), dim3(256), 0, s, oc, st, jpts, jaux, jgrad, jrow, term, rows);
}
void launch_solve(const ObjConst* oc, ObjState* st, const float* partials, double* gsum, int n_slices, const GnParamsDev& prm, int iter,
                  float* trace, const float* codew, const float* b0, const float* blat, float* cbias, const float* depths_next, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_gram_reduce, dim3((72 * 72 + 255) / 256, B, prm.pose_only ? 1 : 2), dim3(256), 0, s, st, partials, n_slices, gsum);
    hipLaunchKernelGGL(k_solve, dim3(B), dim3(SOLVE_THREADS), 0, s, oc, st, gsum, prm, iter, codew, b0, blat, cbias, trace, depths_next, B);
}
void launch_inlier_filter(const ObjConst* oc, ObjState* st, const float* jgrad, unsigned char* alive, int maxM, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_inlier_filter, GRID2(maxM, B), dim3(256), 0, s, oc, st, jgrad, alive);
    hipLaunchKernelGGL(k_count_alive, dim3(B), dim3(256), 0, s, oc, st, alive);
}
void launch_finalize(ObjState* st, const float* scale, int B, int pose_only, float* packed, unsigned* guard_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3((B + 63) / 64), dim3(64), 0, s, st, scale, B, pose_only, packed, guard_out);
}

// Testing (dsp_debug_lie): the Lie-group maps and the rotation prior exactly as k_solve evaluates them -- ONE thread, the same device
// functions, the same fp32 / fp64 arithmetic -- on caller-supplied arguments, so that each branch (theta <= 1e-8, s == 0, the
// `c = 0 if s <= eps` quirk of loss_utils.py:223, the res < 1e-7 zero branch of loss.py:172-173) can be compared with the reference's
// recorded vectors directly instead of through a chained trajectory.
//   kind 0: x[7]  -> out[16] = exp_sim3(x)                          (loss_utils.py:188-233)
//   kind 1: x[6]  -> out[16] = exp_se3(x)                           (loss_utils.py:129-163)
//   kind 2: x[16] = t_obj_cam -> out[0..6] = J_rot, out[7] = res_rot, out[8] = scale, out[9] = d_min, out[10] = d_max
//           (loss.py:155-178 on the state derive_iter_state builds: T_co, det^(1/3), depth range, as k_init_state / k_solve do)
//   kind 3: x[16] = T_oc, x[16..23] = dx[7] -> out[16] = exp_sim3(dx) @ T_oc, the state update of optimizer.py:187-188 in k_solve's order
__global__ void k_debug_lie(int kind, const float* x, float* out, int n_depth) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (kind == 0) { exp_sim3_dev(x, out); return; }
    if (kind == 1) { exp_se3_dev(x, out); return; }
    if (kind == 2) {
        ObjState s;
        for (int i = 0; i < 16; ++i) s.t_oc[i] = x[i];
        s.status = DSP_STATUS_GOOD;
        derive_iter_state(s, n_depth);
        if (s.status != DSP_STATUS_GOOD) {      // singular matrix: derive_iter_state left the derived state unset -- report the status alone
            for (int i = 0; i < 11; ++i) out[i] = 0.f;
            out[11] = (float)s.status;
            return;
        }
        float jrot[7], res;
        rotation_prior(s, jrot, res);
        for (int i = 0; i < 7; ++i) out[i] = jrot[i];
        out[7] = res; out[8] = s.scale; out[9] = s.dmin; out[10] = s.dmax;
        out[11] = (float)s.status;
        return;
    }
    float dT[16];
    exp_sim3_dev(x + 16, dT);
    for (int r = 0; r < 4; ++r)
        for (int cc = 0; cc < 4; ++cc) {
            float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc += dT[4 * r + k] * x[4 * k + cc];
            out[4 * r + cc] = acc;
        }
}
hipError_t launch_debug_lie(int kind, const float* x_dev, float* out_dev, int n_depth, hipStream_t s) {
    hipLaunchKernelGGL(k_debug_lie, dim3(1), dim3(64), 0, s, kind, x_dev, out_dev, n_depth);
    return hipGetLastError();
}

}  // namespace dsp
