"""vors_amd — Python host-side mirror of the reference's tracking interface over libvors_hip.so.

This is plumbing (ctypes over the C ABI of include/vors_hip.h), used by tests/ and bench.py; the product is the HIP
library. Names follow the reference crate (paths relative to the reference repository root):

    Intrinsics                 src/core/camera.rs:84-91
    Config / Config.init       src/core/track/inverse_compositional.rs:37-49, 74-100
    Tracker.track / .current_frame                      inverse_compositional.rs:170-248
    State / Continue / iterative_solve                  src/math/optimizer.rs:9-70   (the "optimizer trait")
    LMOptimizerState                                    src/core/track/lm_optimizer.rs:16-193
    se3_exp / se3_log / so3_exp / so3_log               src/math/se3.rs, src/math/so3.rs

There is no CPU fallback: importing works anywhere (so symbol checks can run without a GPU) but every compute call
raises VorsError when the library or a HIP device is missing.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VORS_HIP_LIB") or os.path.join(_HERE, "libvors_hip.so")  # VORS_HIP_LIB: development builds (ablations)
MAX_LEVELS = 8

ROW_MAJOR, COL_MAJOR = 0, 1
CANDIDATES_COARSE_TO_FINE, CANDIDATES_DENSE, CANDIDATES_DSO = 0, 1, 2
TRACK_OK, TRACK_OPTIMIZER_FAILED_POSE_KEPT = 0, 1
ARITH_REFERENCE, ARITH_EXACT, ARITH_FUSED = 0, 1, 2  # 0 = the reference's own arithmetic AND summation order (bit-identical poses)


class VorsError(RuntimeError):
    pass


class vors_config(C.Structure):
    _fields_ = [
        ("nb_levels", C.c_int32),
        ("candidates_diff_threshold", C.c_int32),
        ("depth_scale", C.c_float),
        ("cu", C.c_float),
        ("cv", C.c_float),
        ("fu", C.c_float),
        ("fv", C.c_float),
        ("skew", C.c_float),
        ("idepth_variance", C.c_float),
        ("candidates_mode", C.c_int32),
        ("huber_delta", C.c_float),
        ("arithmetic", C.c_int32),
    ]


class vors_pair_stats(C.Structure):
    _fields_ = [
        ("lm_model", C.c_float * 7),
        ("optical_flow", C.c_float),
        ("change_keyframe", C.c_int32),
        ("nb_iter", C.c_int32 * MAX_LEVELS),
        ("n_points", C.c_int32 * MAX_LEVELS),
        ("energy", C.c_float * MAX_LEVELS),
        ("nb_grad_evals", C.c_int32 * MAX_LEVELS),
    ]


PAIR_STATS_DTYPE = np.dtype([
    ("lm_model", np.float32, 7),
    ("optical_flow", np.float32),
    ("change_keyframe", np.int32),
    ("nb_iter", np.int32, MAX_LEVELS),
    ("n_points", np.int32, MAX_LEVELS),
    ("energy", np.float32, MAX_LEVELS),
    ("nb_grad_evals", np.int32, MAX_LEVELS),
])
assert PAIR_STATS_DTYPE.itemsize == C.sizeof(vors_pair_stats)


class vors_obs(C.Structure):
    _fields_ = [
        ("cu", C.c_float), ("cv", C.c_float), ("fu", C.c_float), ("fv", C.c_float), ("skew", C.c_float),
        ("rows", C.c_int32), ("cols", C.c_int32),
        ("template_", C.POINTER(C.c_uint8)),
        ("image", C.POINTER(C.c_uint8)),
        ("n", C.c_int32),
        ("coordinates", C.POINTER(C.c_int32)),
        ("_z_candidates", C.POINTER(C.c_float)),
        ("jacobians", C.POINTER(C.c_float)),
        ("huber_delta", C.c_float),
        ("arithmetic", C.c_int32),
    ]


# every symbol include/vors_hip.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "vors_last_error", "vors_device_count", "vors_device_info", "vors_abi_version", "vors_selfcheck_isqrt",
    "vors_tracker_create", "vors_tracker_track", "vors_tracker_track_checked", "vors_tracker_current_frame", "vors_tracker_last_stats",
    "vors_tracker_keyframe", "vors_tracker_destroy",
    "vors_track_pairs",
    "vors_batch_create", "vors_batch_create_on", "vors_batch_device", "vors_batch_track_pairs", "vors_batch_prepare_keyframes", "vors_batch_track_current",
    "vors_batch_workspace_bytes", "vors_batch_enable_kernel_timing", "vors_batch_kernel_times", "vors_batch_last_kernel_ms",
    "vors_batch_destroy",
    "vors_batch_get_keyframe_image", "vors_batch_get_current_image", "vors_batch_get_points", "vors_batch_eval_level",
    "vors_lm_eval", "vors_lm_step", "vors_lm_solve",
    "vors_ref_sincos", "vors_se3_exp", "vors_se3_log", "vors_so3_exp", "vors_so3_log", "vors_iso_mul", "vors_iso_inverse",
    "vors_synth_render_pairs",
    "vors_multi_create", "vors_multi_device_count", "vors_multi_shard", "vors_multi_track_pairs", "vors_multi_track_pairs_host",
    "vors_multi_destroy", "vors_multi_rccl_version",
    "vors_trackers_create", "vors_trackers_create_on", "vors_trackers_count", "vors_trackers_init", "vors_trackers_track", "vors_trackers_state",
    "vors_trackers_current_frames", "vors_trackers_last_stats", "vors_trackers_enable_kernel_timing", "vors_trackers_kernel_times", "vors_trackers_destroy",
    "vors_synth_render_frames",
    "vors_pipeline_create", "vors_pipeline_submit", "vors_pipeline_wait", "vors_pipeline_drain", "vors_pipeline_destroy",
]

_lib = None


def lib():
    """Load libvors_hip.so. Import torch first when sharing device memory with it (same HIP runtime SONAME)."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime: load it FIRST so that libvors_hip.so binds to the same one (two runtimes in one process
        # do not see each other's devices / allocations). The library itself does not depend on torch.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise VorsError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.vors_last_error.restype = C.c_char_p
        vp, i, d, f = C.c_void_p, C.c_int, C.c_double, C.c_float
        _lib.vors_tracker_create.argtypes = [C.POINTER(vors_config), d, vp, d, vp, i, i, i, C.POINTER(vp)]
        _lib.vors_tracker_track.argtypes = [vp, d, vp, d, vp, C.POINTER(i)]
        _lib.vors_tracker_track_checked.argtypes = [vp, d, vp, d, vp, i, i, C.POINTER(i)]
        _lib.vors_tracker_current_frame.argtypes = [vp, C.POINTER(d), vp]
        _lib.vors_tracker_keyframe.argtypes = [vp, C.POINTER(d), vp]
        _lib.vors_tracker_last_stats.argtypes = [vp, C.POINTER(vors_pair_stats)]
        _lib.vors_tracker_destroy.argtypes = [vp]
        _lib.vors_tracker_destroy.restype = None
        _lib.vors_track_pairs.argtypes = [C.POINTER(vors_config), i, vp, vp, vp, i, i, i, vp, vp, vp, vp]
        _lib.vors_batch_create.argtypes = [C.POINTER(vors_config), i, i, i, C.POINTER(vp)]
        _lib.vors_batch_create_on.argtypes = [i, C.POINTER(vors_config), i, i, i, C.POINTER(vp)]
        _lib.vors_batch_device.argtypes = [vp, C.POINTER(i)]
        _lib.vors_multi_create.argtypes = [C.POINTER(vors_config), i, vp, i, i, i, C.POINTER(vp)]
        _lib.vors_multi_device_count.argtypes = [vp]
        _lib.vors_multi_shard.argtypes = [vp, i, i, C.POINTER(i), C.POINTER(i)]
        _lib.vors_multi_track_pairs.argtypes = [vp, i, vp, vp, vp, vp, vp]
        _lib.vors_multi_track_pairs_host.argtypes = [vp, i, vp, vp, vp, vp, vp]
        _lib.vors_multi_destroy.argtypes = [vp]
        _lib.vors_multi_destroy.restype = None
        _lib.vors_multi_rccl_version.argtypes = [vp]
        _lib.vors_trackers_create.argtypes = [C.POINTER(vors_config), i, i, i, C.POINTER(vp)]
        _lib.vors_trackers_create_on.argtypes = [i, C.POINTER(vors_config), i, i, i, C.POINTER(vp)]
        _lib.vors_trackers_count.argtypes = [vp]
        _lib.vors_trackers_init.argtypes = [vp, vp, vp, vp]
        _lib.vors_trackers_track.argtypes = [vp, vp, vp, vp]
        _lib.vors_trackers_state.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        _lib.vors_trackers_current_frames.argtypes = [vp, vp, vp, vp, vp]
        _lib.vors_trackers_last_stats.argtypes = [vp, vp, vp]
        _lib.vors_trackers_enable_kernel_timing.argtypes = [vp, i]
        _lib.vors_trackers_kernel_times.argtypes = [vp, i, vp, i, C.POINTER(i)]
        _lib.vors_trackers_destroy.argtypes = [vp]
        _lib.vors_trackers_destroy.restype = None
        _lib.vors_synth_render_frames.argtypes = [i, vp, vp, vp, i, i, vp, i, vp, vp, vp]
        _lib.vors_batch_track_pairs.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib.vors_pipeline_create.argtypes = [i, C.POINTER(vors_config), i, i, i, i, C.POINTER(vp)]
        _lib.vors_pipeline_submit.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int64)]
        _lib.vors_pipeline_wait.argtypes = [vp, C.c_int64, vp, i]
        _lib.vors_pipeline_drain.argtypes = [vp, vp, i]
        _lib.vors_pipeline_destroy.argtypes = [vp]
        _lib.vors_pipeline_destroy.restype = None
        _lib.vors_batch_prepare_keyframes.argtypes = [vp, i, vp, vp, vp]
        _lib.vors_batch_track_current.argtypes = [vp, i, vp, vp, vp, vp, vp, vp]
        _lib.vors_batch_workspace_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
        _lib.vors_batch_enable_kernel_timing.argtypes = [vp, i]
        _lib.vors_batch_kernel_times.argtypes = [vp, i, vp, i, C.POINTER(i)]
        _lib.vors_batch_last_kernel_ms.argtypes = [vp, C.POINTER(f), C.POINTER(f), C.POINTER(f)]
        _lib.vors_batch_destroy.argtypes = [vp]
        _lib.vors_batch_destroy.restype = None
        _lib.vors_batch_get_keyframe_image.argtypes = [vp, i, i, vp, C.POINTER(i), C.POINTER(i)]
        _lib.vors_batch_get_current_image.argtypes = [vp, i, i, vp, C.POINTER(i), C.POINTER(i)]
        _lib.vors_batch_get_points.argtypes = [vp, i, i, i, vp, vp, vp, vp, C.POINTER(i)]
        _lib.vors_batch_eval_level.argtypes = [vp, i, i, vp, i, vp]
        _lib.vors_lm_eval.argtypes = [C.POINTER(vors_obs), vp, C.POINTER(f), C.POINTER(C.c_int32), vp, vp, vp]
        _lib.vors_ref_sincos.argtypes = [vp, i, vp, vp]
        _lib.vors_ref_sincos.restype = None
        _lib.vors_lm_step.argtypes = [vp, vp, vp, f, vp, C.POINTER(i)]
        _lib.vors_lm_solve.argtypes = [C.POINTER(vors_obs), vp, vp, C.POINTER(C.c_int32), C.POINTER(f), C.POINTER(f), C.POINTER(i)]
        _lib.vors_synth_render_pairs.argtypes = [C.c_uint64, i, i, i, vp, d, i, vp, vp, vp, vp, vp, vp]
        for name in ("vors_se3_exp", "vors_se3_log", "vors_so3_exp", "vors_so3_log", "vors_iso_inverse"):
            getattr(_lib, name).argtypes = [vp, vp]
            getattr(_lib, name).restype = None
        _lib.vors_iso_mul.argtypes = [vp, vp, vp]
        _lib.vors_iso_mul.restype = None
    return _lib


def _check(st):
    if st != 0:
        raise VorsError(f"vors_hip error {st}: {lib().vors_last_error().decode()}")


def device_count():
    return lib().vors_device_count()


def device_info(device=0):
    """-> dict(clock_khz = peak shader clock, compute_units, memory_bytes) of a HIP device."""
    clk, cu, mem = C.c_int(), C.c_int(), C.c_uint64()
    lib().vors_device_info.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    _check(lib().vors_device_info(int(device), C.byref(clk), C.byref(cu), C.byref(mem)))
    return dict(clock_khz=clk.value, compute_units=cu.value, memory_bytes=mem.value)


def selfcheck_isqrt():
    """Arguments 0 .. 65535 for which the DSO selector's four-instruction integer root differs from floor(sqrt(n)) on this device (0)."""
    n = C.c_int(-1)
    lib().vors_selfcheck_isqrt.argtypes = [C.POINTER(C.c_int)]
    _check(lib().vors_selfcheck_isqrt(C.byref(n)))
    return n.value


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ----------------------------------------------------------------------------------------------- reference mirror
class Intrinsics:
    """src/core/camera.rs:84-91."""

    def __init__(self, principal_point, focal, skew=0.0):
        self.principal_point = (float(principal_point[0]), float(principal_point[1]))
        self.focal = (float(focal[0]), float(focal[1]))
        self.skew = float(skew)


# src/dataset/tum_rgbd.rs:15-52
DEPTH_SCALE = 5000.0
INTRINSICS_ICL_NUIM = Intrinsics((319.5, 239.5), (481.20, -480.00), 0.0)
INTRINSICS_FR1 = Intrinsics((318.643040, 255.313989), (517.306408, 516.469215), 0.0)
INTRINSICS_FR2 = Intrinsics((325.141442, 249.701764), (520.908620, 521.007327), 0.0)
INTRINSICS_FR3 = Intrinsics((320.106653, 247.632132), (535.433105, 539.212524), 0.0)


def scaled_intrinsics(rows, cols, base=INTRINSICS_FR1):
    """Intrinsics of the synthetic scenes: FR1 for 640x480; other sizes scale by s = cols / 640:
    f' = s f, c' = s (c + 0.5) - 0.5. Returns (cu, cv, fu, fv, skew)."""
    s = cols / 640.0
    cu, cv = base.principal_point
    fu, fv = base.focal
    return (s * (cu + 0.5) - 0.5, s * (cv + 0.5) - 0.5, s * fu, s * fv, base.skew)


class Config:
    """src/core/track/inverse_compositional.rs:37-49 (+ two extension fields, zero = reference behaviour)."""

    def __init__(self, nb_levels=6, candidates_diff_threshold=7, depth_scale=DEPTH_SCALE, intrinsics=INTRINSICS_FR1,
                 idepth_variance=0.0001, candidates_mode=CANDIDATES_COARSE_TO_FINE, huber_delta=0.0, arithmetic=ARITH_REFERENCE):
        self.nb_levels = nb_levels
        self.candidates_diff_threshold = candidates_diff_threshold
        self.depth_scale = depth_scale
        self.intrinsics = intrinsics
        self.idepth_variance = idepth_variance
        self.candidates_mode = candidates_mode
        self.huber_delta = huber_delta
        self.arithmetic = arithmetic

    def to_c(self):
        k = self.intrinsics
        return vors_config(self.nb_levels, self.candidates_diff_threshold, self.depth_scale, k.principal_point[0],
                           k.principal_point[1], k.focal[0], k.focal[1], k.skew, self.idepth_variance,
                           self.candidates_mode, self.huber_delta, self.arithmetic)

    def init(self, keyframe_depth_timestamp, depth_map, keyframe_img_timestamp, img, layout=ROW_MAJOR):
        """Config::init (inverse_compositional.rs:74-100) -> Tracker."""
        return Tracker(self, keyframe_depth_timestamp, depth_map, keyframe_img_timestamp, img, layout)


class Tracker:
    """core::track::inverse_compositional::Tracker. Construct through Config.init."""

    def __init__(self, config, depth_t, depth_map, img_t, img, layout=ROW_MAJOR):
        img = np.ascontiguousarray(img, np.uint8)
        depth_map = np.ascontiguousarray(depth_map, np.uint16)
        rows, cols = img.shape if layout == ROW_MAJOR else img.shape[::-1]
        self.config = config
        self._shape = (rows, cols)
        self._layout = layout
        self._h = C.c_void_p()
        cfg = config.to_c()
        _check(lib().vors_tracker_create(C.byref(cfg), depth_t, _ptr(depth_map), img_t, _ptr(img), rows, cols, layout,
                                         C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vors_tracker_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def track(self, depth_time, depth_map, img_time, img):
        """Tracker::track (inverse_compositional.rs:170-240). Returns the VORS_TRACK_* status (the reference returns ())."""
        img = np.ascontiguousarray(img, np.uint8)
        depth_map = np.ascontiguousarray(depth_map, np.uint16)
        want = self._shape if self._layout == ROW_MAJOR else self._shape[::-1]
        if img.shape != want or depth_map.shape != want:  # the C ABI carries no dimensions after create(): check them here
            raise VorsError(f"Tracker.track: frame shape {img.shape} / depth shape {depth_map.shape} differ from the {want} "
                            "this tracker was created with")
        st = C.c_int()
        rows, cols = img.shape if self._layout == ROW_MAJOR else img.shape[::-1]
        _check(lib().vors_tracker_track_checked(self._h, depth_time, _ptr(depth_map), img_time, _ptr(img), rows, cols, C.byref(st)))
        return st.value

    def current_frame(self):
        """Tracker::current_frame (inverse_compositional.rs:243-248) -> (depth timestamp, pose7)."""
        t = C.c_double()
        p = np.zeros(7, np.float32)
        _check(lib().vors_tracker_current_frame(self._h, C.byref(t), _ptr(p)))
        return t.value, p

    def keyframe(self):
        t = C.c_double()
        p = np.zeros(7, np.float32)
        _check(lib().vors_tracker_keyframe(self._h, C.byref(t), _ptr(p)))
        return t.value, p

    def last_stats(self):
        s = vors_pair_stats()
        _check(lib().vors_tracker_last_stats(self._h, C.byref(s)))
        return np.frombuffer(bytes(s), PAIR_STATS_DTYPE)[0]


def track_pairs(config, kf_gray, kf_depth, cur_gray, prev_poses7=None, layout=ROW_MAJOR, want_stats=True):
    """Host-buffer batch entry (vors_track_pairs): per pair Config::init(keyframe) + Tracker::track(current)."""
    kf_gray = np.ascontiguousarray(kf_gray, np.uint8)
    kf_depth = np.ascontiguousarray(kf_depth, np.uint16)
    cur_gray = np.ascontiguousarray(cur_gray, np.uint8)
    n = kf_gray.shape[0]
    rows, cols = kf_gray.shape[1:] if layout == ROW_MAJOR else kf_gray.shape[1:][::-1]
    poses = np.zeros((n, 7), np.float32)
    status = np.zeros(n, np.int32)
    stats = np.zeros(n, PAIR_STATS_DTYPE) if want_stats else None
    if prev_poses7 is not None:
        prev_poses7 = np.ascontiguousarray(prev_poses7, np.float32)
    cfg = config.to_c()
    _check(lib().vors_track_pairs(C.byref(cfg), n, _ptr(kf_gray), _ptr(kf_depth), _ptr(cur_gray), rows, cols, layout,
                                  _ptr(prev_poses7), _ptr(poses), _ptr(status), _ptr(stats)))
    return poses, status, stats


class MultiGpu:
    """vors_multi_*: ONE process, several devices, pairs sharded by contiguous blocks, one RCCL all-gather of pose + status."""

    def __init__(self, config, max_pairs_per_device, rows, cols, n_devices=0, device_ids=None):
        self.rows, self.cols = rows, cols
        self._h = C.c_void_p()
        cfg = config.to_c()
        ids = np.ascontiguousarray(device_ids, np.int32) if device_ids is not None else None
        self._device_ids = [int(x) for x in device_ids] if device_ids is not None else None
        _check(lib().vors_multi_create(C.byref(cfg), n_devices, _ptr(ids), max_pairs_per_device, rows, cols, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vors_multi_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def device_count(self):
        return lib().vors_multi_device_count(self._h)

    def rccl_version(self):
        """ncclGetVersion of the RCCL bound at run time; 0 for a one-device handle (RCCL is not loaded then)."""
        return lib().vors_multi_rccl_version(self._h)

    def _device_of(self, k):
        return self._device_ids[k] if self._device_ids is not None else k

    def shard(self, n_total, k):
        a, b = C.c_int(), C.c_int()
        _check(lib().vors_multi_shard(self._h, n_total, k, C.byref(a), C.byref(b)))
        return a.value, b.value

    def track_pairs_host(self, kf_gray, kf_depth, cur_gray):
        kf_gray = np.ascontiguousarray(kf_gray, np.uint8)
        kf_depth = np.ascontiguousarray(kf_depth, np.uint16)
        cur_gray = np.ascontiguousarray(cur_gray, np.uint8)
        n = kf_gray.shape[0]
        for name, a in (("kf_gray", kf_gray), ("kf_depth", kf_depth), ("cur_gray", cur_gray)):  # the C ABI reads n * rows * cols of each
            if a.shape != (n, self.rows, self.cols):
                raise VorsError(f"{name}: expected [{n}, {self.rows}, {self.cols}] images, got {a.shape}")
        poses = np.zeros((n, 7), np.float32)
        status = np.zeros(n, np.int32)
        _check(lib().vors_multi_track_pairs_host(self._h, n, _ptr(kf_gray), _ptr(kf_depth), _ptr(cur_gray), _ptr(poses), _ptr(status)))
        return poses, status

    def track_pairs(self, shards_kf_gray, shards_kf_depth, shards_cur_gray, n_total):
        """Device-resident: one torch tensor per device slot (that device's block of pairs)."""
        nd = self.device_count()
        for ts in (shards_kf_gray, shards_kf_depth, shards_cur_gray):
            if len(ts) != nd:
                raise VorsError(f"expected {nd} shards, got {len(ts)}")
            for k, t in enumerate(ts):
                cnt = self.shard(n_total, k)[1]
                if cnt and (t is None or tuple(t.shape) != (cnt, self.rows, self.cols) or not t.is_contiguous() or t.device.index != self._device_of(k)):
                    raise VorsError(f"shard {k}: expected a contiguous [{cnt}, {self.rows}, {self.cols}] tensor on device slot {k}, got "
                                    f"{None if t is None else (tuple(t.shape), t.device)}")
        arr = lambda ts: (C.c_void_p * nd)(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])
        poses = np.zeros((n_total, 7), np.float32)
        status = np.zeros(n_total, np.int32)
        _check(lib().vors_multi_track_pairs(self._h, n_total, arr(shards_kf_gray), arr(shards_kf_depth), arr(shards_cur_gray), _ptr(poses),
                                            _ptr(status)))
        return poses, status


class Batch:
    """Device-resident engine (vors_batch_*). Tensors are torch CUDA(HIP) tensors; work is enqueued on torch's current
    stream and not synchronised."""

    def __init__(self, config, max_pairs, rows, cols, device=None):
        self.config, self.max_pairs, self.rows, self.cols = config, max_pairs, rows, cols
        self._h = C.c_void_p()
        cfg = config.to_c()
        if device is None:
            _check(lib().vors_batch_create(C.byref(cfg), max_pairs, rows, cols, C.byref(self._h)))
        else:
            _check(lib().vors_batch_create_on(int(device), C.byref(cfg), max_pairs, rows, cols, C.byref(self._h)))

    def device(self):
        d = C.c_int()
        _check(lib().vors_batch_device(self._h, C.byref(d)))
        return d.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vors_batch_destroy(self._h)
            except Exception:
                pass
            self._h = None

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _dp(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def workspace_bytes(self):
        b = C.c_uint64()
        _check(lib().vors_batch_workspace_bytes(self._h, C.byref(b)))
        return b.value

    def enable_kernel_timing(self, ring=64):
        _check(lib().vors_batch_enable_kernel_timing(self._h, int(ring)))

    STAGES = {"pyramid_keyframe": 0, "keyframe": 1, "pyramid_current": 2, "lm": 3}

    def kernel_times(self, stage):
        """Durations (ms) of the last min(steps, ring) steps of a stage, oldest first (HIP events on the stream)."""
        out = np.zeros(4096, np.float32)
        n = C.c_int()
        _check(lib().vors_batch_kernel_times(self._h, self.STAGES[stage], _ptr(out), 4096, C.byref(n)))
        return out[:n.value].copy()

    def last_kernel_ms(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _check(lib().vors_batch_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(lm_ms=a.value, keyframe_ms=b.value, pyramid_ms=c.value)

    def _check_images(self, *tensors):
        for t in tensors:
            if t is None:
                continue
            if tuple(t.shape[-2:]) != (self.rows, self.cols) or not t.is_contiguous() or t.shape[0] > self.max_pairs:
                raise VorsError(f"expected contiguous [n <= {self.max_pairs}, {self.rows}, {self.cols}] images, got {tuple(t.shape)}")

    def prepare_keyframes(self, kf_gray, kf_depth):
        n = kf_gray.shape[0]
        self._check_images(kf_gray, kf_depth)
        # lifetime contract of vors_batch_prepare_keyframes: the handle keeps POINTERS to level 0 and the depth map (zero copy);
        # hold the tensors so that torch's caching allocator cannot hand their memory to someone else while the handle uses them
        self._kf_refs = (kf_gray, kf_depth)
        _check(lib().vors_batch_prepare_keyframes(self._h, n, self._dp(kf_gray), self._dp(kf_depth), self._stream()))

    def track_current(self, cur_gray, out_poses7, out_status, out_stats=None, prev_poses7=None):
        n = cur_gray.shape[0]
        self._check_images(cur_gray)
        self._cur_ref = cur_gray
        _check(lib().vors_batch_track_current(self._h, n, self._dp(cur_gray), self._dp(prev_poses7), self._dp(out_poses7),
                                              self._dp(out_status), self._dp(out_stats), self._stream()))

    def track_pairs(self, kf_gray, kf_depth, cur_gray, out_poses7, out_status, out_stats=None, prev_poses7=None):
        n = kf_gray.shape[0]
        self._check_images(kf_gray, kf_depth, cur_gray)
        self._kf_refs, self._cur_ref = (kf_gray, kf_depth), cur_gray
        _check(lib().vors_batch_track_pairs(self._h, n, self._dp(kf_gray), self._dp(kf_depth), self._dp(cur_gray),
                                            self._dp(prev_poses7), self._dp(out_poses7), self._dp(out_status),
                                            self._dp(out_stats), self._stream()))

    def eval_level(self, pair, level, model7, arithmetic):
        """One evaluation of a level of a pair at `model7` in the given arithmetic -> (sum r^2, n_inside, g[6], H[6,6])."""
        m = np.ascontiguousarray(model7, np.float32)
        out = np.zeros(29, np.float32)
        _check(lib().vors_batch_eval_level(self._h, pair, level, _ptr(m), int(arithmetic), _ptr(out)))
        H = np.zeros((6, 6), np.float32)
        H[np.triu_indices(6)] = out[8:29]
        H = H + np.triu(H, 1).T
        return float(out[0]), int(out[1]), out[2:8].copy(), H

    def keyframe_image(self, pair, level):
        out = np.empty(self.rows * self.cols, np.uint8)
        r, c = C.c_int(), C.c_int()
        _check(lib().vors_batch_get_keyframe_image(self._h, pair, level, _ptr(out), C.byref(r), C.byref(c)))
        return out[:r.value * c.value].reshape(r.value, c.value).copy()

    def current_image(self, pair, level):
        out = np.empty(self.rows * self.cols, np.uint8)
        r, c = C.c_int(), C.c_int()
        _check(lib().vors_batch_get_current_image(self._h, pair, level, _ptr(out), C.byref(r), C.byref(c)))
        return out[:r.value * c.value].reshape(r.value, c.value).copy()

    def points(self, pair, level):
        """Usable candidates of a level in device slot order: xy[n,2], idepth[n], jac[n,6], tmpl[n]."""
        cap = (self.rows >> level) * (self.cols >> level)
        xy = np.empty((cap, 2), np.int32)
        iz = np.empty(cap, np.float32)
        jac = np.empty((cap, 6), np.float32)
        tm = np.empty(cap, np.uint8)
        n = C.c_int()
        _check(lib().vors_batch_get_points(self._h, pair, level, cap, _ptr(xy), _ptr(iz), _ptr(jac), _ptr(tm), C.byref(n)))
        n = n.value
        return xy[:n].copy(), iz[:n].copy(), jac[:n].copy(), tm[:n].copy()


class Pipeline:
    """vors_pipeline_*: a ring of `depth` batch handles on internal streams for a continuous feed of independent batches (throughput mode).
    submit() orders the step after everything on torch's current stream and returns a ticket; wait() / drain() order torch's current
    stream after the step(s) (host=True: block the calling thread instead). Results are those of Batch.track_pairs bit for bit."""

    def __init__(self, config, max_pairs, rows, cols, depth=2, device=None):
        self.config, self.max_pairs, self.rows, self.cols, self.depth = config, max_pairs, rows, cols, depth
        self._h = C.c_void_p()
        self._refs = {}
        cfg = config.to_c()
        _check(lib().vors_pipeline_create(-1 if device is None else int(device), C.byref(cfg), int(depth), max_pairs, rows, cols,
                                          C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vors_pipeline_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def submit(self, kf_gray, kf_depth, cur_gray, out_poses7, out_status, out_stats=None, prev_poses7=None):
        n = kf_gray.shape[0]
        for t in (kf_gray, kf_depth, cur_gray):
            if tuple(t.shape) != (n, self.rows, self.cols) or not t.is_contiguous() or n > self.max_pairs:
                raise VorsError(f"expected contiguous [n <= {self.max_pairs}, {self.rows}, {self.cols}] images, got {tuple(t.shape)}")
        ticket = C.c_int64(-1)
        st = lib().vors_pipeline_submit(self._h, n, Batch._dp(kf_gray), Batch._dp(kf_depth), Batch._dp(cur_gray), Batch._dp(prev_poses7),
                                        Batch._dp(out_poses7), Batch._dp(out_status), Batch._dp(out_stats), Batch._stream(), C.byref(ticket))
        # the step reads and writes these buffers until it completes: keep them alive for as long as its slot can be running it. A call
        # that failed before it was given a ticket touches no slot's references (ticket -1 would otherwise evict those of slot depth - 1
        # while that slot's step may still be running); its buffers are kept aside instead.
        refs = (kf_gray, kf_depth, cur_gray, out_poses7, out_status, out_stats, prev_poses7)
        if ticket.value >= 0:
            self._refs[ticket.value % self.depth] = refs
        else:
            self._refs.setdefault("failed", []).append(refs)
        _check(st)
        return ticket.value

    def wait(self, ticket, host=False):
        _check(lib().vors_pipeline_wait(self._h, int(ticket), Batch._stream(), 1 if host else 0))

    def drain(self, host=False):
        _check(lib().vors_pipeline_drain(self._h, Batch._stream(), 1 if host else 0))


class Trackers:
    """N sequences in lock-step, device resident (vors_trackers_*): Config::init / Tracker::track / current_frame for every
    sequence with the whole tracker state machine (poses, keyframe test, per-sequence keyframe promotion) on the device."""

    def __init__(self, config, n_sequences, rows, cols, device=None):
        self.config, self.n, self.rows, self.cols = config, n_sequences, rows, cols
        self._h = C.c_void_p()
        cfg = config.to_c()
        if device is None:
            _check(lib().vors_trackers_create(C.byref(cfg), n_sequences, rows, cols, C.byref(self._h)))
        else:
            _check(lib().vors_trackers_create_on(int(device), C.byref(cfg), n_sequences, rows, cols, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vors_trackers_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def _check_frames(self, gray, depth):
        for t in (gray, depth):
            if tuple(t.shape) != (self.n, self.rows, self.cols) or not t.is_contiguous():
                raise VorsError(f"expected contiguous [{self.n}, {self.rows}, {self.cols}] frames, got {tuple(t.shape)}")

    def init(self, gray, depth):
        """Config::init for every sequence (frame 0). Only enqueues work on torch's current stream: the frames must stay alive (and unmodified)
        until that work has run — this object keeps a reference until the next call, like track()."""
        self._check_frames(gray, depth)
        self._last = (gray, depth)
        _check(lib().vors_trackers_init(self._h, Batch._dp(gray), Batch._dp(depth), Batch._stream()))

    def track(self, gray, depth):
        """Tracker::track for every sequence (one new frame each); only enqueues work on torch's current stream."""
        self._check_frames(gray, depth)
        self._last = (gray, depth)  # keep the frames alive until the next call (the enqueued work reads them)
        _check(lib().vors_trackers_track(self._h, Batch._dp(gray), Batch._dp(depth), Batch._stream()))

    def current_frames(self):
        """-> (poses7 [n,7], status [n], keyframe frame index [n]) on the host; synchronises the current stream."""
        poses = np.zeros((self.n, 7), np.float32)
        status = np.zeros(self.n, np.int32)
        kf = np.zeros(self.n, np.int32)
        _check(lib().vors_trackers_current_frames(self._h, _ptr(poses), _ptr(status), _ptr(kf), Batch._stream()))
        return poses, status, kf

    def stats(self):
        """Diagnostics of the last track() of every sequence (host copy; synchronises the current stream)."""
        out = np.zeros(self.n, PAIR_STATS_DTYPE)
        _check(lib().vors_trackers_last_stats(self._h, _ptr(out), Batch._stream()))
        return out

    def enable_kernel_timing(self, ring=64):
        _check(lib().vors_trackers_enable_kernel_timing(self._h, int(ring)))

    def kernel_times(self, stage):
        out = np.zeros(4096, np.float32)
        n = C.c_int()
        _check(lib().vors_trackers_kernel_times(self._h, Batch.STAGES[stage], _ptr(out), 4096, C.byref(n)))
        return out[:n.value].copy()


def synth_render_frames(seeds, salts, xis, rows, cols, cam5, invalid_percent=2, device="cuda"):
    """Frames of the synthetic scene at explicit twists (vors_synth_render_frames) -> gray u8 [n,rows,cols], depth (int16 payload u16)."""
    import torch
    seeds = np.ascontiguousarray(seeds, np.uint64)
    salts = np.ascontiguousarray(salts, np.uint64)
    xis = np.ascontiguousarray(xis, np.float64).reshape(-1, 6)
    n = len(seeds)
    assert len(salts) == n and len(xis) == n
    gray = torch.empty((n, rows, cols), dtype=torch.uint8, device=device)
    depth = torch.empty((n, rows, cols), dtype=torch.int16, device=device)
    cam = np.asarray(cam5, np.float64)
    _check(lib().vors_synth_render_frames(n, _ptr(seeds), _ptr(salts), _ptr(xis), rows, cols, _ptr(cam), invalid_percent,
                                          Batch._dp(gray), Batch._dp(depth), Batch._stream()))
    return gray, depth


def stats_tensor(n, device="cuda"):
    """Device buffer for n vors_pair_stats (as raw bytes); decode with decode_stats()."""
    import torch
    return torch.zeros(n * PAIR_STATS_DTYPE.itemsize, dtype=torch.uint8, device=device)


def decode_stats(t):
    return np.frombuffer(t.cpu().numpy().tobytes(), PAIR_STATS_DTYPE)


def synth_render_pairs(seed0, n_pairs, rows, cols, cam5, motion_scale=1.0, invalid_percent=2, want_cur_depth=False,
                       device="cuda"):
    """Render synthetic frame pairs on the device (vors_synth_render_pairs). Returns torch tensors."""
    import torch
    kg = torch.empty((n_pairs, rows, cols), dtype=torch.uint8, device=device)
    kd = torch.empty((n_pairs, rows, cols), dtype=torch.int16, device=device)  # u16 payload
    cg = torch.empty((n_pairs, rows, cols), dtype=torch.uint8, device=device)
    cd = torch.empty((n_pairs, rows, cols), dtype=torch.int16, device=device) if want_cur_depth else None
    gt = torch.empty((n_pairs, 7), dtype=torch.float32, device=device)
    cam = np.asarray(cam5, np.float64)
    _check(lib().vors_synth_render_pairs(int(seed0), n_pairs, rows, cols, _ptr(cam), float(motion_scale), invalid_percent,
                                         Batch._dp(kg), Batch._dp(kd), Batch._dp(cg), Batch._dp(cd), Batch._dp(gt),
                                         Batch._stream()))
    return kg, kd, cg, cd, gt


# ----------------------------------------------------------------------------------------------- operator level
class Obs:
    """lm_optimizer.rs:43-58 (hessians are recomputed on the device, not passed)."""

    def __init__(self, intrinsics5, template, image, coordinates, _z_candidates, jacobians, huber_delta=0.0, arithmetic=ARITH_REFERENCE):
        self.intrinsics = np.ascontiguousarray(intrinsics5, np.float32)
        self.template = np.ascontiguousarray(template, np.uint8)
        self.image = np.ascontiguousarray(image, np.uint8)
        self.coordinates = np.ascontiguousarray(coordinates, np.int32).reshape(-1, 2)
        self._z_candidates = np.ascontiguousarray(_z_candidates, np.float32)
        self.jacobians = np.ascontiguousarray(jacobians, np.float32).reshape(-1, 6)
        self.huber_delta = float(huber_delta)
        self.arithmetic = int(arithmetic)  # ARITH_REFERENCE: sequential sums in the order of `coordinates`

    def to_c(self):
        k = self.intrinsics
        rows, cols = self.template.shape
        u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
        return vors_obs(k[0], k[1], k[2], k[3], k[4], rows, cols, self.template.ctypes.data_as(u8p),
                        self.image.ctypes.data_as(u8p), len(self._z_candidates), self.coordinates.ctypes.data_as(i32p),
                        self._z_candidates.ctypes.data_as(f32p), self.jacobians.ctypes.data_as(f32p), self.huber_delta, self.arithmetic)


def lm_eval(obs, model7, want_residuals=False):
    """eval_energy + compute_eval_data (lm_optimizer.rs:68-107) on the device -> energy, n_inside, g[6], H[6,6]."""
    model7 = np.ascontiguousarray(model7, np.float32)
    e, n = C.c_float(), C.c_int32()
    g = np.zeros(6, np.float32)
    H = np.zeros((6, 6), np.float32)
    res = np.zeros(len(obs._z_candidates), np.float32) if want_residuals else None
    o = obs.to_c()
    _check(lib().vors_lm_eval(C.byref(o), _ptr(model7), C.byref(e), C.byref(n), _ptr(g), _ptr(H), _ptr(res)))
    return (e.value, n.value, g, H, res) if want_residuals else (e.value, n.value, g, H)


def ref_sincos(x):
    """sinf / cosf as se3::exp evaluates them on host and device (lie.h ref_sinf / ref_cosf) -> (sin, cos) float32 arrays."""
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().vors_ref_sincos(_ptr(x), x.size, _ptr(s), _ptr(c))
    return s, c


def lm_step(H, g, model7, lm_coef):
    """step() (lm_optimizer.rs:123-136). Returns (ok, model7)."""
    H = np.ascontiguousarray(H, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    model7 = np.ascontiguousarray(model7, np.float32)
    out = np.zeros(7, np.float32)
    ok = C.c_int()
    _check(lib().vors_lm_step(_ptr(H), _ptr(g), _ptr(model7), C.c_float(lm_coef), _ptr(out), C.byref(ok)))
    return bool(ok.value), out


def lm_solve(obs, model7):
    """State::iterative_solve for LMOptimizerState, whole loop on the device -> status, model7, nb_iter, energy, lm_coef."""
    model7 = np.ascontiguousarray(model7, np.float32)
    out = np.zeros(7, np.float32)
    it, e, lam, st = C.c_int32(), C.c_float(), C.c_float(), C.c_int()
    o = obs.to_c()
    _check(lib().vors_lm_solve(C.byref(o), _ptr(model7), _ptr(out), C.byref(it), C.byref(e), C.byref(lam), C.byref(st)))
    return st.value, out, it.value, e.value, lam.value


class Continue:
    """src/math/optimizer.rs:9-14."""
    Stop, Forward = 0, 1


class State:
    """The optimizer trait (src/math/optimizer.rs:32-70). Subclasses provide init / step / eval / stop_criterion;
    `iterative_solve` is the provided method. `step` signals Err by raising StepError."""

    class StepError(Exception):
        pass

    @classmethod
    def init(cls, obs, model):
        raise NotImplementedError

    def step(self):
        raise NotImplementedError

    def eval(self, obs, new_model):
        raise NotImplementedError

    def stop_criterion(self, nb_iter, eval_state):
        raise NotImplementedError

    @classmethod
    def iterative_solve(cls, obs, initial_model):
        state = cls.init(obs, initial_model)
        nb_iter = 0
        while True:
            nb_iter += 1
            new_model = state.step()
            eval_state = state.eval(obs, new_model)
            state, continuation = state.stop_criterion(nb_iter, eval_state)
            if continuation == Continue.Stop:
                return state, nb_iter


class EvalData:
    """lm_optimizer.rs:31-40."""

    def __init__(self, hessian, gradient, energy, model):
        self.hessian, self.gradient, self.energy, self.model = hessian, gradient, energy, model


class LMOptimizerState(State):
    """impl optimizer::State<Obs, EvalState, Iso3, String> for LMOptimizerState (lm_optimizer.rs:111-193), host-driven:
    eval runs on the device through vors_lm_eval, step through vors_lm_step. EvalState = EvalData | float (Err)."""

    def __init__(self, lm_coef, eval_data):
        self.lm_coef, self.eval_data = lm_coef, eval_data

    @staticmethod
    def _full_eval(obs, model):
        e, _, g, H = lm_eval(obs, model)
        return EvalData(H, g, np.float32(e), np.asarray(model, np.float32))

    @classmethod
    def init(cls, obs, model):
        return cls(np.float32(0.1), cls._full_eval(obs, model))

    def step(self):
        ok, model = lm_step(self.eval_data.hessian, self.eval_data.gradient, self.eval_data.model, float(self.lm_coef))
        if not ok:
            raise State.StepError("Error at Cholesky decomposition of hessian")
        return model

    def eval(self, obs, model):
        new = self._full_eval(obs, model)  # the device computes energy and (g, H) in one fused pass
        if new.energy > self.eval_data.energy:
            return float(new.energy)
        return new

    def stop_criterion(self, nb_iter, eval_state):
        too_many_iterations = nb_iter > 20
        is_err = not isinstance(eval_state, EvalData)
        if is_err and too_many_iterations:
            return self, Continue.Stop
        if too_many_iterations:
            return LMOptimizerState(self.lm_coef, eval_state), Continue.Stop
        if is_err:
            return LMOptimizerState(np.float32(self.lm_coef * np.float32(10.0)), self.eval_data), Continue.Forward
        d_energy = np.float32(self.eval_data.energy - eval_state.energy)
        cont = Continue.Forward if d_energy > 1.0 else Continue.Stop
        return LMOptimizerState(np.float32(np.float32(0.1) * self.lm_coef), eval_state), cont


def _vecfn(name, n_in, n_out):
    def f(x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        assert x.size == n_in
        out = np.zeros(n_out, np.float32)
        getattr(lib(), name)(_ptr(x), _ptr(out))
        return out
    f.__name__ = name
    return f


se3_exp = _vecfn("vors_se3_exp", 6, 7)
se3_log = _vecfn("vors_se3_log", 7, 6)
so3_exp = _vecfn("vors_so3_exp", 3, 4)
so3_log = _vecfn("vors_so3_log", 4, 3)
iso_inverse = _vecfn("vors_iso_inverse", 7, 7)


def iso_mul(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(7, np.float32)
    lib().vors_iso_mul(_ptr(a), _ptr(b), _ptr(out))
    return out
