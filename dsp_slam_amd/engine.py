"""Python face of libdspgn: one Engine per (decoder, GPU).  Thin: marshals numpy arrays into the C ABI.

Everything numeric happens in the HIP library; there is no CPU or PyTorch fallback.
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib as L


def gn_params(k1=1.0, k2=100.0, k3=0.25, k4=1e7, b1=0.2, b2=0.025, lr=1.0, s_damp=1.0, num_iterations=10,
              num_depth_samples=50, cut_off=0.01, pose_only_iterations=5):
    return L.GnParams(k1, k2, k3, k4, b1, b2, lr, s_damp, int(num_iterations), int(num_depth_samples), cut_off,
                      int(pose_only_iterations))


def params_from_configs(configs):
    """Hyper-parameters as Optimizer.__init__ reads them (reference reconstruct/optimizer.py:27-43)."""
    o = configs["optimizer"] if isinstance(configs, dict) else configs.optimizer
    get = (lambda d, k: d[k]) if isinstance(o, dict) else getattr
    j = get(o, "joint_optim")
    try:
        pose_it = get(get(o, "pose_only_optim"), "num_iterations")
    except (KeyError, AttributeError):
        pose_it = 5
    return gn_params(get(j, "k1"), get(j, "k2"), get(j, "k3"), get(j, "k4"), get(j, "b1"), get(j, "b2"),
                     get(j, "learning_rate"), get(j, "scale_damping"), get(j, "num_iterations"),
                     get(o, "num_depth_samples"), get(o, "cut_off_threshold"), pose_it)


def _ragged(arrays, width):
    """list of (n_i, width) arrays -> (offsets int64 (B+1), flat float32 (sum n_i, width))."""
    arrays = [L.f32(a).reshape(-1, width) if width else L.f32(a).reshape(-1) for a in arrays]
    off = np.zeros(len(arrays) + 1, np.int64)
    off[1:] = np.cumsum([a.shape[0] for a in arrays])
    flat = np.concatenate(arrays, 0) if arrays else np.zeros((0, width), np.float32)
    if flat.size == 0:
        flat = np.zeros((1, width) if width else (1,), np.float32)
    return off, np.ascontiguousarray(flat, np.float32)


class Batch(object):
    """Device-resident batch of objects (dsp_batch_*): upload once, run many times."""

    def __init__(self, engine, prm, t_cam_obj, pts, rays, depth, codes=None, trace=False):
        self.engine = engine
        self.n = len(pts)
        self._keep = (
            _ragged(pts, 3), _ragged(rays, 3), _ragged(depth, 0),
            L.f32(np.stack([np.asarray(t, np.float32).reshape(4, 4) for t in t_cam_obj])),
            None if codes is None else L.f32(np.stack([L.code64(c) for c in codes])),
        )
        (po, p), (ro, r), (do, d), t, c = self._keep
        self._h = C.c_void_p()
        lib = L.load()
        L.check(lib.dsp_batch_create(engine._h, C.byref(prm), self.n, L.ptr(po, L.c_i64p), L.ptr(p), L.ptr(ro, L.c_i64p),
                                     L.ptr(r), L.ptr(do, L.c_i64p), L.ptr(d), L.ptr(t), L.ptr(c), C.byref(self._h)),
                engine._h, "dsp_batch_create")
        self.iters = prm.num_iterations
        engine._batches.add(self)          # Engine.close() closes its live batches first: a batch must not outlive its handle
        if trace:
            L.check(lib.dsp_batch_enable_trace(self._h, 1), engine._h, "dsp_batch_enable_trace")

    # ---- the five settings of the C ABI (include/dsp_gn.h) ------------------------------------------------------------------------------
    def set_ray_passes(self, n):
        """0 = automatic, 1 = decode every in-sphere sample (reference behaviour), n = n front-to-back depth ranges."""
        L.check(L.load().dsp_batch_set_ray_passes(self._h, int(n)), self.engine._h, "dsp_batch_set_ray_passes")

    def set_prepass(self, mode=-1, delta=-1.0):
        """Low-precision pre-classification of the forward ray samples: -1 automatic, 0 off, 1 f16, 2 bf16; delta < 0 = default margin.
        Results are bit-identical for every setting whose delta exceeds the decoder's prepass error."""
        L.check(L.load().dsp_batch_set_prepass(self._h, int(mode), float(delta)), self.engine._h, "dsp_batch_set_prepass")

    def set_prepass_guard(self, on=True):
        """The always-on guard of the prepass (include/dsp_gn.h): off only to see what an unguarded run would return."""
        L.check(L.load().dsp_batch_set_prepass_guard(self._h, int(bool(on))), self.engine._h, "dsp_batch_set_prepass_guard")

    def set_kernel_timing(self, mode):
        """HIP events around every decoder launch (stats ms_mlp_*): -1 = automatic (batches of more than 16 objects), 0 = off, 1 = on."""
        L.check(L.load().dsp_batch_set_kernel_timing(self._h, int(mode)), self.engine._h, "dsp_batch_set_kernel_timing")

    def set_compute(self, mode):
        """0 = fp32 (default: the parity path), 1 = f16, 2 = bf16: the opt-in low-precision compute mode (dsp_batch_set_compute) -- NOT bit- or
        1e-4-comparable with the reference; see include/dsp_gn.h."""
        L.check(L.load().dsp_batch_set_compute(self._h, int(mode)), self.engine._h, "dsp_batch_set_compute")

    def set_iterations(self, n):
        L.check(L.load().dsp_batch_set_iterations(self._h, int(n)), self.engine._h, "dsp_batch_set_iterations")
        self.iters = int(n)

    # ---- testing: pin one of the bit-identical forms the library chooses between by itself (dsp_batch_set_debug) --------------------------
    def set_debug(self, key, value):
        L.check(L.load().dsp_batch_set_debug(self._h, int(key), int(value)), self.engine._h, "dsp_batch_set_debug(%d, %d)" % (key, value))

    def set_mask_reuse(self, mode):
        """-1 = automatic, 0 = render rows share the surface points' forward+backward launch, 1 = backward-only from exported masks."""
        self.set_debug(L.DBG_MASK_REUSE, mode)

    def set_split_rows(self, mode):
        """-1 = automatic, 0 = 64-point throughput tiles, 1 = 16-point latency tiles for the jacobian launch (when mask reuse is off)."""
        self.set_debug(L.DBG_SPLIT_ROWS, mode)

    def set_tail_split(self, mode):
        """-1 = automatic, 0 = off, 1 = the last partial round of the fp32 forward launch runs as 16-point latency-form tiles."""
        self.set_debug(L.DBG_TAIL_SPLIT, mode)

    def set_wave_bookkeeping(self, mode):
        """-1 = automatic, 0 = per-ray bookkeeping as count / scan / write launches (throughput form), 1 = one wave per ray over the whole chip."""
        self.set_debug(L.DBG_WAVE_BOOKKEEPING, mode)

    def set_speculative_band(self, mode):
        """-1 = automatic, 0 = band samples get a forward launch of their own, 1 = they go straight into the jacobian launch (latency path)."""
        self.set_debug(L.DBG_SPECULATIVE_BAND, mode)

    def set_mixed_reuse(self, mode):
        """-1 = automatic, 0 = off, 1 = kept render rows backward-only from exported masks INSIDE the latency-form jacobian launch."""
        self.set_debug(L.DBG_MIXED_REUSE, mode)

    def set_cluster_tiles(self, mode):
        """-1 = automatic, 0 = one workgroup per 16-point jacobian tile, 1 = four (cluster form) for lists of up to 128 tiles."""
        self.set_debug(L.DBG_CLUSTER_TILES, mode)

    def set_direct_tiles(self, mode):
        """-1 automatic / 1: a one-object batch's decoder kernels derive their tile lists themselves; 0: k_build_tiles launches."""
        self.set_debug(L.DBG_DIRECT_TILES, mode)

    def set_prepass_tile(self, points=-1):
        """-1 = automatic, 128 or 64 points per prepass tile.  Results are identical for either."""
        self.set_debug(L.DBG_PREPASS_TILE, points)

    def set_prepass_audit(self, on=True):
        self.set_debug(L.DBG_PREPASS_AUDIT, int(bool(on)))

    def set_lp_small_batches(self, mode):
        """-1 / 0 = a detection-sized batch keeps the fp32 latency path when the low-precision compute mode is set (faster there), 1 = the mode applies to it too."""
        self.set_debug(L.DBG_LP_SMALL_BATCHES, mode)

    def set_cluster_fault(self, on):
        """Fault injection: the following runs' cluster launches lose one workgroup's hand-off; False also ends the handle's cool-down."""
        self.set_debug(L.DBG_CLUSTER_FAULT, int(bool(on)))

    # ---- forensics ------------------------------------------------------------------------------------------------------------------------
    def set_ray_pass_bounds(self, bounds):
        bounds = np.ascontiguousarray(bounds, np.int32)
        L.check(L.load().dsp_batch_debug_ray_pass_bounds(self._h, L.ptr(bounds, L.c_i32p), bounds.shape[0] - 1), self.engine._h,
                "dsp_batch_debug_ray_pass_bounds")

    def set_start_state(self, t_obj_cam=None, codes=None, depths=None):
        """Testing / forensics: start the following runs from these camera->object matrices (taken bit for bit) and / or codes; depths
        (per object, num_depth_samples values): the first iteration samples the rays at exactly these depths."""
        t = None if t_obj_cam is None else L.f32(np.stack([np.asarray(x, np.float32).reshape(4, 4) for x in t_obj_cam]))
        c = None if codes is None else L.f32(np.stack([L.code64(x) for x in codes]))
        d = None
        if depths is not None:
            d = np.zeros((self.n, 64), np.float32)
            for i, row in enumerate(depths):
                row = np.asarray(row, np.float32).reshape(-1)
                d[i, :row.shape[0]] = row
        L.check(L.load().dsp_batch_debug_start_state(self._h, L.ptr(t), L.ptr(c), L.ptr(d)), self.engine._h, "dsp_batch_debug_start_state")

    def set_depth_schedule(self, depths=None):
        """Testing / forensics: depths[e][i] = the depth samples object i uses in iteration e (None = derive them from the pose again)."""
        if depths is None:
            L.check(L.load().dsp_batch_debug_depth_schedule(self._h, None, 0), self.engine._h, "dsp_batch_debug_depth_schedule")
            return
        n_it = len(depths)
        d = np.zeros((n_it, self.n, 64), np.float32)
        for e in range(n_it):
            for i in range(self.n):
                row = np.asarray(depths[e][i], np.float32).reshape(-1)
                d[e, i, :row.shape[0]] = row
        L.check(L.load().dsp_batch_debug_depth_schedule(self._h, L.ptr(d), n_it), self.engine._h, "dsp_batch_debug_depth_schedule")

    def debug_samples(self, obj, n_rays, n_depth):
        """(in-sphere mask (n_rays, n_depth) bool, sdf grid, de_ds grid) the last iteration of the last run left for object obj
        (NaN outside the sphere; de_ds != 0 marks a kept sample)."""
        rm = np.zeros(n_rays, np.uint64)
        sdf = np.zeros((n_rays, n_depth), np.float32)
        deds = np.zeros((n_rays, n_depth), np.float32)
        L.check(L.load().dsp_batch_debug_samples(self._h, int(obj), L.ptr(rm, C.POINTER(C.c_uint64)), L.ptr(sdf), L.ptr(deds), sdf.size),
                self.engine._h, "dsp_batch_debug_samples")
        mask = ((rm[:, None] >> np.arange(n_depth, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        return mask, sdf, deds

    def run(self):
        L.check(L.load().dsp_batch_run(self._h), self.engine._h, "dsp_batch_run")

    def results(self):
        n = self.n
        t = np.zeros((n, 4, 4), np.float32)
        code = np.zeros((n, L.CODE_LEN), np.float32)
        loss = np.zeros(n, np.float32)
        status = np.zeros(n, np.int32)
        L.check(L.load().dsp_batch_results(self._h, L.ptr(t), L.ptr(code), L.ptr(loss), L.ptr(status, L.c_i32p)),
                self.engine._h, "dsp_batch_results")
        return t, np.ascontiguousarray(code[:, :self.engine.code_len]), loss, status

    def results_packed_to_device(self, dst_ptr):
        """Copy the packed result rows (n x 82 float32, distributed.RESULT_WIDTH) device-to-device to dst_ptr (e.g. a torch tensor's data_ptr() on
        this batch's GPU): the results never touch the host in front of the multi-GPU gather."""
        L.check(L.load().dsp_batch_results_packed_dev(self._h, C.c_void_p(int(dst_ptr))), self.engine._h, "dsp_batch_results_packed_dev")

    def stats(self):
        s = L.Stats()
        L.check(L.load().dsp_batch_stats(self._h, C.byref(s)), self.engine._h, "dsp_batch_stats")
        return {k: getattr(s, k) for k, _ in L.Stats._fields_}

    def trace(self, iteration):
        n = self.n
        out = dict(H=np.zeros((n, 71, 71), np.float32), b=np.zeros((n, 71), np.float32), dx=np.zeros((n, 71), np.float32),
                   V=np.zeros(n, np.int64), m=np.zeros(n, np.int64), K=np.zeros(n, np.int64),
                   t_obj_cam=np.zeros((n, 4, 4), np.float32), code=np.zeros((n, L.CODE_LEN), np.float32),
                   set_sums=np.zeros((n, 2), np.uint32), depths=np.zeros((n, 64), np.float32))
        L.check(L.load().dsp_batch_trace(self._h, int(iteration), L.ptr(out["H"]), L.ptr(out["b"]), L.ptr(out["dx"]),
                                         L.ptr(out["V"], L.c_i64p), L.ptr(out["m"], L.c_i64p), L.ptr(out["K"], L.c_i64p),
                                         L.ptr(out["t_obj_cam"]), L.ptr(out["code"]), L.ptr(out["set_sums"], C.POINTER(C.c_uint32)), L.ptr(out["depths"])),
                self.engine._h, "dsp_batch_trace")
        return out

    def close(self):
        if self._h:
            L.load().dsp_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_results_c(engines, packed):
    """dsp_gather_results: one RCCL gather, inside ONE process, of the (n_i, 82) result blocks of several engines (one per GPU)
    to the first engine's GPU; returns them concatenated in engine order.  (Across processes use dsp_slam_amd.distributed.)"""
    n = len(engines)
    blocks = [L.f32(np.asarray(p, np.float32).reshape(-1, 82)) for p in packed]
    hs = (C.c_void_p * n)(*[e._h for e in engines])
    ptrs = (L.c_f32p * n)(*[L.ptr(b) for b in blocks])
    cnt = np.array([b.shape[0] for b in blocks], np.int32)
    out = np.zeros((int(cnt.sum()), 82), np.float32)
    L.check(L.load().dsp_gather_results(hs, n, ptrs, L.ptr(cnt, L.c_i32p), L.ptr(out)), engines[0]._h, "dsp_gather_results")
    return out


def gather_batches_c(batches):
    """dsp_gather_batch_results: the same gather straight from device-resident batches (one per GPU, each run before): device ->
    ncclGather -> host once."""
    n = len(batches)
    hs = (C.c_void_p * n)(*[b._h for b in batches])
    out = np.zeros((sum(b.n for b in batches), 82), np.float32)
    L.check(L.load().dsp_gather_batch_results(hs, n, L.ptr(out)), batches[0].engine._h, "dsp_gather_batch_results")
    return out


def pack_results_c(t_cam_obj, codes, loss, status):
    n = len(loss)
    out = np.zeros((n, 82), np.float32)
    codes = np.stack([L.code64(c) for c in codes]) if n else np.zeros((0, L.CODE_LEN), np.float32)
    L.load().dsp_pack_results(n, L.ptr(L.f32(t_cam_obj)), L.ptr(L.f32(codes)), L.ptr(L.f32(loss)), L.ptr(np.ascontiguousarray(status, np.int32), L.c_i32p),
                              L.ptr(out))
    return out


_last_engine = None
_live_engines = weakref.WeakSet()


@atexit.register
def _close_engines_at_exit():
    """Engines (and their batches) that are still alive when the interpreter exits -- a test that failed half way, a script that never called
    close() -- are closed HERE, while the interpreter and the HIP runtime are intact and in the right order (batches first), instead of by
    finalisers in no particular order."""
    for e in list(_live_engines):
        try:
            e.close()
        except Exception:
            pass


def last_engine():
    """The engine created last that is still alive (for module-level helpers of the reference API that take no decoder
    argument, such as reconstruct.utils.convert_sdf_voxels_to_mesh)."""
    e = _last_engine() if _last_engine is not None else None
    if e is None or not e._h:
        raise RuntimeError("no decoder is loaded on a GPU (get_decoder / config_decoder first)")
    return e


class Engine(object):
    """Owns a dsp_handle: packed decoder weights on one MI355X + a HIP stream."""

    def __init__(self, layers, latent_in, code_len=64, device=0):
        """layers: list of (W (out,in), b (out,)) float32 with weight-norm already folded."""
        lib = L.load()
        self._desc = L.DecoderDescHolder(layers, latent_in, code_len)
        self._batches = weakref.WeakSet()
        _live_engines.add(self)
        self._h = C.c_void_p()
        rc = lib.dsp_create(C.byref(self._desc.desc), int(device), C.byref(self._h))
        if rc != 0:
            msg = lib.dsp_last_error(None)
            raise L.DspError("dsp_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.device = int(device)
        self.code_len = int(code_len)
        global _last_engine
        _last_engine = weakref.ref(self)

    # -- decoder ------------------------------------------------------------------------------------
    def decode_sdf(self, code, pts):
        pts = L.f32(pts).reshape(-1, 3)
        code = L.code64(code)
        out = np.zeros(pts.shape[0], np.float32)
        L.check(L.load().dsp_decode_sdf(self._h, L.ptr(code), L.ptr(pts), pts.shape[0], L.ptr(out)), self._h, "dsp_decode_sdf")
        return out

    def decode_sdf_prepass(self, code, pts, dtype=L.PREPASS_F16):
        """The decoder through the low-precision prepass kernel (f16 / bf16 MFMA).  Calibration and tests only: the optimiser
        uses these values to classify samples, never as results."""
        pts = L.f32(pts).reshape(-1, 3)
        code = L.code64(code)
        out = np.zeros(pts.shape[0], np.float32)
        L.check(L.load().dsp_decode_sdf_prepass(self._h, int(dtype), L.ptr(code), L.ptr(pts), pts.shape[0], L.ptr(out)), self._h,
                "dsp_decode_sdf_prepass")
        return out

    def prepass_calibration(self, dtype=L.PREPASS_F16):
        """(largest |sdf_lp - sdf_fp32| measured at creation, margin derived from it) for this decoder."""
        err, delta = C.c_float(0), C.c_float(0)
        L.check(L.load().dsp_prepass_calibration(self._h, int(dtype), C.byref(err), C.byref(delta)), self._h, "dsp_prepass_calibration")
        return err.value, delta.value

    def prepass_calibration_table(self, dtype=L.PREPASS_F16):
        """dict(mags, max_err, delta: 5 entries each; guard_err): the margin as a function of the code's largest entry."""
        m, e, d = (np.zeros(5, np.float32) for _ in range(3))
        g = C.c_float(0)
        L.check(L.load().dsp_prepass_calibration_table(self._h, int(dtype), L.ptr(m), L.ptr(e), L.ptr(d), C.byref(g)), self._h,
                "dsp_prepass_calibration_table")
        return dict(mags=m, max_err=e, delta=d, guard_err=g.value)

    def prepass_reset_guard(self):
        """Forget what earlier guard trips on this engine left behind: the margins return to the decoder's calibration."""
        L.check(L.load().dsp_prepass_reset_guard(self._h), self._h, "dsp_prepass_reset_guard")

    def decode_sdf_multi(self, codes, pts):
        """(n_codes, 64) codes x one shared (n, 3) point set -> (n_codes, n) sdf, one kernel launch."""
        pts = L.f32(pts).reshape(-1, 3)
        codes = np.asarray(codes, np.float32)
        codes = L.f32(np.stack([L.code64(c) for c in codes.reshape(-1, codes.shape[-1])]))
        out = np.zeros((codes.shape[0], pts.shape[0]), np.float32)
        L.check(L.load().dsp_decode_sdf_multi(self._h, L.ptr(codes), codes.shape[0], L.ptr(pts), pts.shape[0], L.ptr(out)),
                self._h, "dsp_decode_sdf_multi")
        return out

    # -- mesh extraction ---------------------------------------------------------------------------
    def _fetch_mesh(self, nv, nf):
        verts = np.zeros((nv.value, 3), np.float32)
        faces = np.zeros((nf.value, 3), np.int32)
        L.check(L.load().dsp_mesh_fetch(self._h, L.ptr(verts), nv.value, L.ptr(faces, L.c_i32p), nf.value), self._h, "dsp_mesh_fetch")
        return verts, faces

    def extract_mesh(self, code, vol_dim, regular_grid=False):
        """Grid decode + marching cubes on the device (the SDF volume never leaves HBM): vertices (V,3) float32 in the
        decoder's [-1,1]^3 frame, faces (F,3) int32.  Empty when the surface does not cross the grid.  regular_grid=False
        samples the reference's (sheared) grid, see reconstruct.utils.create_voxel_grid."""
        code = L.code64(code)
        nv, nf = C.c_int64(0), C.c_int64(0)
        L.check(L.load().dsp_extract_mesh(self._h, L.ptr(code), int(vol_dim), 1 if regular_grid else 0, C.byref(nv), C.byref(nf)), self._h, "dsp_extract_mesh")
        return self._fetch_mesh(nv, nf)

    def marching_cubes(self, volume, level=0.0, spacing=1.0, origin=0.0):
        """Marching cubes of a host volume on the device: vertices = index * spacing + origin."""
        vol = L.f32(volume)
        if vol.ndim != 3:
            raise ValueError("volume must be 3-D")
        nv, nf = C.c_int64(0), C.c_int64(0)
        L.check(L.load().dsp_marching_cubes(self._h, L.ptr(vol), vol.shape[0], vol.shape[1], vol.shape[2], float(level), float(spacing),
                                            float(origin), C.byref(nv), C.byref(nf)), self._h, "dsp_marching_cubes")
        return self._fetch_mesh(nv, nf)

    def sdf_jacobian(self, code, pts):
        pts = L.f32(pts).reshape(-1, 3)
        code = L.code64(code)
        n = pts.shape[0]
        sdf = np.zeros(n, np.float32)
        grad = np.zeros((n, L.GRAD_DIM), np.float32)
        L.check(L.load().dsp_sdf_jacobian(self._h, L.ptr(code), L.ptr(pts), n, L.ptr(sdf), L.ptr(grad)), self._h, "dsp_sdf_jacobian")
        if self.code_len != L.CODE_LEN:     # d/d[code(code_len), xyz]: drop the unused code columns
            grad = np.ascontiguousarray(np.concatenate([grad[:, :self.code_len], grad[:, L.CODE_LEN:]], 1))
        return sdf, grad

    def sdf_jacobian_lp(self, code, pts, dtype=L.COMPUTE_F16):
        """sdf and d sdf / d [code, xyz] through the 16-bit jacobian kernels of the low-precision compute mode (accuracy measurements)."""
        pts = L.f32(pts).reshape(-1, 3)
        code = L.code64(code)
        n = pts.shape[0]
        sdf = np.zeros(n, np.float32)
        grad = np.zeros((n, L.GRAD_DIM), np.float32)
        L.check(L.load().dsp_sdf_jacobian_lp(self._h, int(dtype), L.ptr(code), L.ptr(pts), n, L.ptr(sdf), L.ptr(grad)), self._h, "dsp_sdf_jacobian_lp")
        if self.code_len != L.CODE_LEN:
            grad = np.ascontiguousarray(np.concatenate([grad[:, :self.code_len], grad[:, L.CODE_LEN:]], 1))
        return sdf, grad

    # -- residual terms -----------------------------------------------------------------------------
    def compute_sdf_loss(self, pts_cam, t_obj_cam, code):
        pts = L.f32(pts_cam).reshape(-1, 3)
        n = pts.shape[0]
        t = L.f32(t_obj_cam).reshape(4, 4)
        code = L.code64(code)
        j7 = np.zeros((n, 7), np.float32)
        jc = np.zeros((n, L.CODE_LEN), np.float32)
        r = np.zeros(n, np.float32)
        L.check(L.load().dsp_compute_sdf_loss(self._h, L.ptr(pts), n, L.ptr(t), L.ptr(code), L.ptr(j7), L.ptr(jc), L.ptr(r)),
                self._h, "dsp_compute_sdf_loss")
        return j7, jc[:, :self.code_len], r

    def compute_render_loss(self, rays, depth_obs, t_obj_cam, sampled_depth, code, th=0.01):
        rays = L.f32(rays).reshape(-1, 3)
        depth_obs = L.f32(depth_obs).reshape(-1)
        sampled = L.f32(sampled_depth).reshape(-1)
        t = L.f32(t_obj_cam).reshape(4, 4)
        code = L.code64(code)
        cap = rays.shape[0] * sampled.shape[0]
        j7 = np.zeros((cap, 7), np.float32)
        jc = np.zeros((cap, L.CODE_LEN), np.float32)
        r = np.zeros(cap, np.float32)
        k = C.c_int64(0)
        v = C.c_int64(0)
        m = C.c_int64(0)
        L.check(L.load().dsp_compute_render_loss(self._h, L.ptr(rays), rays.shape[0], L.ptr(depth_obs), L.ptr(t), L.ptr(sampled),
                                                 sampled.shape[0], L.ptr(code), float(th), C.byref(k), L.ptr(j7), L.ptr(jc),
                                                 L.ptr(r), C.byref(v), C.byref(m)), self._h, "dsp_compute_render_loss")
        stats = dict(V=v.value, m=m.value, K=k.value)
        if k.value < 0:
            return None, stats
        return (j7[:k.value].copy(), jc[:k.value, :self.code_len].copy(), r[:k.value].copy()), stats

    # -- optimiser ----------------------------------------------------------------------------------
    def fail_alloc(self, n):
        """Testing: the (n + 1)-th fresh device allocation of this handle's pool from now on fails like an exhausted HBM (n < 0: off)."""
        L.check(L.load().dsp_debug_fail_alloc(self._h, int(n)), self._h, "dsp_debug_fail_alloc")

    def trim(self):
        """Hand the handle's cached device blocks and pinned staging buffers back to the runtime (dsp_trim): for a process that shares the
        GPU with torch / RCCL / another handle and has just finished a large one-shot batch."""
        L.check(L.load().dsp_trim(self._h), self._h, "dsp_trim")

    def set_stream_priority(self, priority):
        """1 = highest, 0 = default, -1 = lowest queue priority of the handle's HIP stream (dsp_set_stream_priority): who yields on a shared GPU."""
        L.check(L.load().dsp_set_stream_priority(self._h, int(priority)), self._h, "dsp_set_stream_priority")

    def debug_lie(self, kind, x, n_depth=50):
        """Testing: exp_sim3 (kind 0, x[7]), exp_se3 (1, x[6]), the rotation prior + derived state (2, t_obj_cam 4x4) or the Sim(3) state
        update exp_sim3(dx) @ t_obj_cam (3, 16 + 7 floats) evaluated by the device functions the solve kernel calls.  Returns 16 floats."""
        x = L.f32(np.asarray(x, np.float32).reshape(-1))
        out = np.zeros(16, np.float32)
        L.check(L.load().dsp_debug_lie(self._h, int(kind), L.ptr(x), int(n_depth), L.ptr(out)), self._h, "dsp_debug_lie")
        return out

    def batch(self, prm, t_cam_obj, pts, rays, depth, codes=None, trace=False):
        return Batch(self, prm, t_cam_obj, pts, rays, depth, codes, trace)

    def reconstruct_batch(self, prm, t_cam_obj, pts, rays, depth, codes=None, compute=L.COMPUTE_F32):
        """compute: L.COMPUTE_F32 (default, the parity path) or the opt-in low-precision mode L.COMPUTE_F16 / _BF16 (dsp_batch_set_compute)."""
        if len(pts) == 0:      # an empty shard (more ranks than objects): nothing to run, but the caller still joins the gather
            return (np.zeros((0, 4, 4), np.float32), np.zeros((0, self.code_len), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32))
        b = Batch(self, prm, t_cam_obj, pts, rays, depth, codes)
        try:
            if compute != L.COMPUTE_F32:
                b.set_compute(compute)
            b.run()
            return b.results()
        finally:
            b.close()

    def estimate_pose_batch(self, prm, t_co_se3, scale, pts, codes):
        n = len(pts)
        if n == 0:
            return np.zeros((0, 4, 4), np.float32)
        po, p = _ragged(pts, 3)
        t = L.f32(np.stack([np.asarray(x, np.float32).reshape(4, 4) for x in t_co_se3]))
        sc = L.f32(np.asarray(scale, np.float32).reshape(n))
        cd = L.f32(np.stack([L.code64(c) for c in codes]))
        out = np.zeros((n, 4, 4), np.float32)
        L.check(L.load().dsp_estimate_pose_batch(self._h, C.byref(prm), n, L.ptr(po, L.c_i64p), L.ptr(p), L.ptr(t), L.ptr(sc),
                                                 L.ptr(cd), L.ptr(out)), self._h, "dsp_estimate_pose_batch")
        return out

    def close(self):
        if self._h:
            for b in list(self._batches):      # dsp_batch_destroy takes the handle's mutex: never after dsp_destroy
                b.close()
            L.load().dsp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
