"""ctypes binding of libdspgn.so (the C ABI in include/dsp_gn.h).

No fallback: if the library is missing or a call fails this raises -- the product path never runs on
the CPU oracle or on PyTorch ops.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)

CODE_LEN = 64
GRAD_DIM = 67

OBJ_GOOD, OBJ_FEW_SAMPLES, OBJ_NAN = 0, 1, 2
PREPASS_OFF, PREPASS_F16, PREPASS_BF16 = 0, 1, 2
PREPASS_SMALL_TILES = 0x100
COMPUTE_F32, COMPUTE_F16, COMPUTE_BF16 = 0, 1, 2
# keys of dsp_batch_set_debug (include/dsp_gn.h: DSP_DBG_*)
(DBG_MASK_REUSE, DBG_SPLIT_ROWS, DBG_TAIL_SPLIT, DBG_WAVE_BOOKKEEPING, DBG_SPECULATIVE_BAND, DBG_MIXED_REUSE, DBG_CLUSTER_TILES, DBG_DIRECT_TILES,
 DBG_PREPASS_TILE, DBG_PREPASS_AUDIT, DBG_CLUSTER_FAULT, DBG_LP_SMALL_BATCHES) = range(1, 13)
ABI_VERSION = 6


class DecoderDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("code_len", C.c_int32), ("latent_in", C.c_int32),
                ("out_dims", c_i32p), ("in_dims", c_i32p),
                ("weights", C.POINTER(c_f32p)), ("biases", C.POINTER(c_f32p))]


class GnParams(C.Structure):
    _fields_ = [("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float), ("k4", C.c_float),
                ("b1", C.c_float), ("b2", C.c_float), ("lr", C.c_float), ("s_damp", C.c_float),
                ("num_iterations", C.c_int32), ("num_depth_samples", C.c_int32), ("cut_off", C.c_float),
                ("pose_only_iterations", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_fwd_points", C.c_double), ("n_jac_points", C.c_double), ("ms_total", C.c_double),
                ("ms_mlp_fwd", C.c_double), ("ms_mlp_jac", C.c_double),
                ("n_mlp_fwd_launches", C.c_int32), ("n_mlp_jac_launches", C.c_int32), ("n_insphere_points", C.c_double), ("n_render_rows", C.c_double),
                ("n_prepass_points", C.c_double), ("ms_mlp_prepass", C.c_double), ("n_mlp_prepass_launches", C.c_int32),
                ("prepass_mode", C.c_int32), ("prepass_delta", C.c_float), ("prepass_max_err", C.c_float),
                ("prepass_misclassified", C.c_double), ("prepass_audited", C.c_double),
                ("prepass_guard_trips", C.c_double), ("prepass_guard_objects", C.c_double), ("prepass_guard_max_err", C.c_float),
                ("prepass_guard_rerun", C.c_int32), ("n_cluster_tiles", C.c_double), ("cluster_fallback", C.c_int32), ("cluster_cooldown", C.c_int32)]


class DspError(RuntimeError):
    pass


_lib = None

# every symbol include/dsp_gn.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ("dsp_abi_version", C.c_int, []),
    ("dsp_device_count", C.c_int, []),
    ("dsp_build_info", C.c_char_p, []),
    ("dsp_runtime_versions", C.c_int, [c_i32p, c_i32p]),
    ("dsp_create", C.c_int, [C.POINTER(DecoderDesc), C.c_int, C.POINTER(_VP)]),
    ("dsp_destroy", None, [_VP]),
    ("dsp_last_error", C.c_char_p, [_VP]),
    ("dsp_decode_sdf", C.c_int, [_VP, c_f32p, c_f32p, C.c_int64, c_f32p]),
    ("dsp_decode_sdf_multi", C.c_int, [_VP, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p]),
    ("dsp_decode_sdf_prepass", C.c_int, [_VP, C.c_int, c_f32p, c_f32p, C.c_int64, c_f32p]),
    ("dsp_debug_pack_prepass", C.c_int, [C.POINTER(DecoderDesc), C.c_int, C.POINTER(C.c_uint16), c_i64p, c_i32p, c_i32p]),
    ("dsp_debug_pack_lpj", C.c_int, [C.POINTER(DecoderDesc), C.c_int, C.POINTER(C.c_uint16), c_i64p, c_i32p, c_i32p]),
    ("dsp_sdf_jacobian", C.c_int, [_VP, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p]),
    ("dsp_compute_sdf_loss", C.c_int, [_VP, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]),
    ("dsp_compute_render_loss", C.c_int, [_VP, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int32, c_f32p, C.c_float,
                                          c_i64p, c_f32p, c_f32p, c_f32p, c_i64p, c_i64p]),
    ("dsp_reconstruct_batch", C.c_int, [_VP, C.POINTER(GnParams), C.c_int32, c_i64p, c_f32p, c_i64p, c_f32p, c_i64p, c_f32p,
                                        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p]),
    ("dsp_estimate_pose_batch", C.c_int, [_VP, C.POINTER(GnParams), C.c_int32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]),
    ("dsp_batch_create", C.c_int, [_VP, C.POINTER(GnParams), C.c_int32, c_i64p, c_f32p, c_i64p, c_f32p, c_i64p, c_f32p,
                                   c_f32p, c_f32p, C.POINTER(_VP)]),
    ("dsp_batch_run", C.c_int, [_VP]),
    ("dsp_batch_results", C.c_int, [_VP, c_f32p, c_f32p, c_f32p, c_i32p]),
    ("dsp_batch_stats", C.c_int, [_VP, C.POINTER(Stats)]),
    ("dsp_batch_set_ray_passes", C.c_int, [_VP, C.c_int]),
    ("dsp_batch_set_prepass", C.c_int, [_VP, C.c_int, C.c_float]),
    ("dsp_batch_set_prepass_guard", C.c_int, [_VP, C.c_int]),
    ("dsp_batch_set_kernel_timing", C.c_int, [_VP, C.c_int]),
    ("dsp_batch_set_iterations", C.c_int, [_VP, C.c_int32]),
    ("dsp_batch_set_compute", C.c_int, [_VP, C.c_int]),
    ("dsp_sdf_jacobian_lp", C.c_int, [_VP, C.c_int, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p]),
    ("dsp_batch_set_debug", C.c_int, [_VP, C.c_int, C.c_int]),
    ("dsp_prepass_calibration", C.c_int, [_VP, C.c_int, c_f32p, c_f32p]),
    ("dsp_prepass_calibration_table", C.c_int, [_VP, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p]),
    ("dsp_batch_debug_ray_pass_bounds", C.c_int, [_VP, c_i32p, C.c_int]),
    ("dsp_batch_debug_start_state", C.c_int, [_VP, c_f32p, c_f32p, c_f32p]),
    ("dsp_batch_debug_depth_schedule", C.c_int, [_VP, c_f32p, C.c_int32]),
    ("dsp_batch_debug_samples", C.c_int, [_VP, C.c_int32, C.POINTER(C.c_uint64), c_f32p, c_f32p, C.c_int64]),
    ("dsp_debug_fail_alloc", C.c_int, [_VP, C.c_int]),
    ("dsp_prepass_reset_guard", C.c_int, [_VP]),
    ("dsp_trim", C.c_int, [_VP]),
    ("dsp_set_stream_priority", C.c_int, [_VP, C.c_int]),
    ("dsp_debug_lie", C.c_int, [_VP, C.c_int, c_f32p, C.c_int32, c_f32p]),
    ("dsp_batch_enable_trace", C.c_int, [_VP, C.c_int]),
    ("dsp_batch_trace", C.c_int, [_VP, C.c_int32, c_f32p, c_f32p, c_f32p, c_i64p, c_i64p, c_i64p, c_f32p, c_f32p, C.POINTER(C.c_uint32), c_f32p]),
    ("dsp_batch_destroy", None, [_VP]),
    ("dsp_pack_results", None, [C.c_int32, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p]),
    ("dsp_gather_results", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(c_f32p), c_i32p, c_f32p]),
    ("dsp_gather_batch_results", C.c_int, [C.POINTER(_VP), C.c_int32, c_f32p]),
    ("dsp_batch_results_packed_dev", C.c_int, [_VP, C.c_void_p]),
    ("dsp_extract_mesh", C.c_int, [_VP, c_f32p, C.c_int32, C.c_int32, c_i64p, c_i64p]),
    ("dsp_marching_cubes", C.c_int, [_VP, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, c_i64p, c_i64p]),
    ("dsp_mesh_fetch", C.c_int, [_VP, c_f32p, C.c_int64, c_i32p, C.c_int64]),
    ("dsp_debug_split_layout", C.c_int, [C.POINTER(DecoderDesc), c_i32p, c_i32p, c_i64p]),
    ("dsp_debug_mc_table", C.c_int, [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]),
    ("dsp_debug_code_bias", C.c_int, [C.POINTER(DecoderDesc), c_f32p, c_f32p]),
    ("dsp_debug_pack", C.c_int, [C.POINTER(DecoderDesc), c_f32p, c_i64p, c_f32p, c_i64p, c_i32p, c_i32p, c_f32p]),
    # include/dsp_pose_graph.h (host fp64; no handle)
    ("dsp_pg_from_matrix", C.c_int, [C.c_int64, c_f64p, c_f64p]),
    ("dsp_pg_to_matrix", C.c_int, [C.c_int64, c_f64p, c_f64p]),
    ("dsp_pg_log", C.c_int, [C.c_int64, c_f64p, c_f64p]),
    ("dsp_pg_exp", C.c_int, [C.c_int64, c_f64p, c_f64p]),
    ("dsp_pg_edge_error", C.c_int, [C.c_int64, c_f64p, c_f64p, c_f64p, c_f64p]),
    ("dsp_pg_edge_linearize", C.c_int, [C.c_int64, c_f64p, c_f64p, c_f64p, c_f64p]),
    ("dsp_pg_edge_chi2", C.c_int, [C.c_int64, c_f64p, C.c_double, C.c_double, c_f64p, c_f64p, c_f64p]),
    ("dsp_pg_vertex_oplus", C.c_int, [C.c_int64, C.c_int, c_f64p, c_f64p, c_f64p]),
]


def lib_path():
    return os.environ.get("DSPGN_LIB", _build.LIB_PATH)


def load():
    """Load libdspgn.so (building it first if hipcc is available and the .so is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if path == _build.LIB_PATH and _build.is_stale():
        try:
            _build.build_locked()     # one rank builds, the others wait on the lock file and find it fresh
        except Exception as e:  # stale-but-present is usable on a box without hipcc
            if not os.path.exists(path):
                raise DspError("libdspgn.so is not built and cannot be built here: %s" % e)
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError => the .so does not match include/dsp_gn.h
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def code64(code):
    """A shape code as the C ABI carries it: DSP_CODE_LEN (64) floats; a 32-D code occupies the first 32 entries, the rest is zero
    (the decoder has no weights for them and the optimiser leaves them alone)."""
    c = np.asarray(code, np.float32).reshape(-1)
    out = np.zeros(CODE_LEN, np.float32)
    n = min(c.shape[0], CODE_LEN)
    out[:n] = c[:n]
    return out


def ptr(a, typ=c_f32p):
    return None if a is None else a.ctypes.data_as(typ)


def check(rc, handle=None, what=""):
    if rc != 0:
        msg = load().dsp_last_error(handle)
        raise DspError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


class DecoderDescHolder(object):
    """Keeps the numpy arrays referenced by a DecoderDesc alive."""

    def __init__(self, layers, latent_in, code_len):
        self.w = [f32(w) for w, _ in layers]
        self.b = [f32(b) for _, b in layers]
        n = len(layers)
        self.out_dims = (C.c_int32 * n)(*[w.shape[0] for w in self.w])
        self.in_dims = (C.c_int32 * n)(*[w.shape[1] for w in self.w])
        self.wp = (c_f32p * n)(*[ptr(w) for w in self.w])
        self.bp = (c_f32p * n)(*[ptr(b) for b in self.b])
        lat = [int(x) for x in latent_in]
        self.desc = DecoderDesc(n, int(code_len), lat[0] if len(lat) == 1 else -1,
                                C.cast(self.out_dims, c_i32p), C.cast(self.in_dims, c_i32p),
                                C.cast(self.wp, C.POINTER(c_f32p)), C.cast(self.bp, C.POINTER(c_f32p)))
