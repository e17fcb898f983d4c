"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libvors_oracle.so (the CPU restatement of the reference hot path, see
oracle/vors_oracle.hpp). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvors_oracle.so")


class Config(C.Structure):
    """Same layout as `vors_config` (include/vors_hip.h) / `vo_config` (oracle)."""

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
    ]


# /root/reference/src/dataset/tum_rgbd.rs:23-52
INTRINSICS_FR1 = (318.643040, 255.313989, 517.306408, 516.469215, 0.0)
INTRINSICS_FR2 = (325.141442, 249.701764, 520.908620, 521.007327, 0.0)
INTRINSICS_FR3 = (320.106653, 247.632132, 535.433105, 539.212524, 0.0)
INTRINSICS_ICL_NUIM = (319.5, 239.5, 481.20, -480.00, 0.0)


def make_config(nb_levels=6, intr=INTRINSICS_FR1, thresh=7, depth_scale=5000.0, idepth_variance=1e-4,
                candidates_mode=0, huber_delta=0.0):
    """Defaults = /root/reference/src/bin/vors_track.rs:34-40."""
    return Config(nb_levels, thresh, depth_scale, intr[0], intr[1], intr[2], intr[3], intr[4], idepth_variance,
                  candidates_mode, huber_delta)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("vors_oracle.hpp", "vors_oracle_capi.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u16(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


_variants = {}


def _cpu_id():
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        model = [l for l in txt.splitlines() if l.startswith("model name")][:1]
        flags = [l for l in txt.splitlines() if l.startswith("flags")][:1]
        import hashlib
        return (model[0] if model else "?") + " " + hashlib.sha1((flags[0] if flags else "").encode()).hexdigest()[:12]
    except OSError:
        return "unknown"


def variant_lib(name):
    """A sensitivity / baseline build of the same sources (oracle/Makefile): "acc64" (f64 sums), "nalg<bit>" (one nalgebra
    assumption swapped), "native" (-march=native, for the CPU baseline; built on the host that runs it). NOT the oracle:
    only vo_track_pairs is bound."""
    if name not in _variants:
        path = os.path.join(_HERE, f"libvors_oracle_{name}.so")
        if name == "native":
            tag = os.path.join(_HERE, "libvors_oracle_native.cpu")
            have = open(tag).read() if os.path.exists(tag) and os.path.exists(path) else None
            if have != _cpu_id():
                subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
                with open(tag, "w") as f:
                    f.write(_cpu_id())
        elif not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _variants[name] = C.CDLL(path)
    return _variants[name]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.vo_tracker_create.restype = C.c_void_p
        _lib.vo_tracker_create.argtypes = [C.POINTER(Config), C.c_double, C.POINTER(C.c_uint16), C.c_double,
                                           C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int]
        _lib.vo_tracker_destroy.argtypes = [C.c_void_p]
        _lib.vo_tracker_track.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_uint16), C.c_double, C.POINTER(C.c_uint8)]
        _lib.vo_tracker_current_frame.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        _lib.vo_tracker_keyframe_pose.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        _lib.vo_tracker_num_levels.argtypes = [C.c_void_p]
        _lib.vo_tracker_level.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_float)]
        _lib.vo_tracker_get_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8)]
        _lib.vo_tracker_get_gradients.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                                                  C.POINTER(C.c_uint16)]
        _lib.vo_tracker_get_mask.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        _lib.vo_tracker_get_points.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                               C.POINTER(C.c_float)]
        _lib.vo_tracker_last.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        _lib.vo_synth_frame.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                        C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.c_int]
        _lib.vo_synth_pair_twist.argtypes = [C.c_uint64, C.c_double, C.POINTER(C.c_double)]
        _lib.vo_synth_gt_model7.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_float)]
    return _lib


# ----------------------------------------------------------------------------- synthetic scenes
def scaled_intrinsics(rows, cols, base=INTRINSICS_FR1):
    """FR1 intrinsics for 640x480; for other sizes scale by s = cols/640: f' = s f, c' = s (c + 0.5) - 0.5."""
    s = cols / 640.0
    cu, cv, fu, fv, sk = base
    return (s * (cu + 0.5) - 0.5, s * (cv + 0.5) - 0.5, s * fu, s * fv, sk)


def synth_frame(seed, xi, rows, cols, intr, invalid_percent=2, frame_salt=0, n_threads=4):
    gray = np.empty((rows, cols), np.uint8)
    depth = np.empty((rows, cols), np.uint16)
    cam = np.asarray(intr, np.float64)
    xi = np.asarray(xi, np.float64)
    lib().vo_synth_frame(int(seed), int(frame_salt), _f64(cam), _f64(xi), rows, cols, invalid_percent, _u8(gray), _u16(depth),
                         n_threads)
    return gray, depth


def pair_twist(seed, motion_scale=1.0):
    xi = np.zeros(6, np.float64)
    lib().vo_synth_pair_twist(int(seed), float(motion_scale), _f64(xi))
    return xi


def gt_model7(xi):
    out = np.zeros(7, np.float32)
    xi = np.asarray(xi, np.float64)
    lib().vo_synth_gt_model7(_f64(xi), _f32(out))
    return out


def synth_pair(seed, rows=480, cols=640, intr=None, motion_scale=1.0, invalid_percent=2, n_threads=4):
    """One frame pair of SURVEY.md §8d: keyframe at identity, current at exp(xi(seed)).
    Returns kf_gray, kf_depth, cur_gray, cur_depth, gt_model7 (keyframe->current camera)."""
    intr = intr or scaled_intrinsics(rows, cols)
    xi = pair_twist(seed, motion_scale)
    kg, kd = synth_frame(seed, np.zeros(6), rows, cols, intr, invalid_percent, 0, n_threads)
    cg, cd = synth_frame(seed, xi, rows, cols, intr, invalid_percent, 1, n_threads)
    return kg, kd, cg, cd, gt_model7(xi)


def synth_batch(n, rows=480, cols=640, seed0=0x5EED0000, intr=None, motion_scale=1.0, n_threads=8):
    kg = np.empty((n, rows, cols), np.uint8)
    kd = np.empty((n, rows, cols), np.uint16)
    cg = np.empty((n, rows, cols), np.uint8)
    cd = np.empty((n, rows, cols), np.uint16)
    gt = np.empty((n, 7), np.float32)
    for i in range(n):
        kg[i], kd[i], cg[i], cd[i], gt[i] = synth_pair(seed0 + i, rows, cols, intr, motion_scale, n_threads=n_threads)
    return kg, kd, cg, cd, gt


# ----------------------------------------------------------------------------- tracker mirror
class Tracker:
    """Oracle mirror of core::track::inverse_compositional::Tracker (inverse_compositional.rs:31)."""

    def __init__(self, cfg, depth_t, depth, img_t, gray, keep_debug=True):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        self.rows, self.cols = gray.shape
        self.cfg = cfg
        self._h = lib().vo_tracker_create(C.byref(cfg), depth_t, _u16(depth), img_t, _u8(gray), self.rows, self.cols,
                                          int(keep_debug))
        if not self._h:
            raise ValueError("oracle: pyramid shorter than nb_levels (the reference would panic)")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            try:
                _lib.vo_tracker_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def track(self, depth_t, depth, img_t, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        return lib().vo_tracker_track(self._h, depth_t, _u16(depth), img_t, _u8(gray))

    def current_frame(self):
        t = C.c_double()
        p = np.zeros(7, np.float32)
        lib().vo_tracker_current_frame(self._h, C.byref(t), _f32(p))
        return t.value, p

    def keyframe_pose(self):
        t = C.c_double()
        p = np.zeros(7, np.float32)
        lib().vo_tracker_keyframe_pose(self._h, C.byref(t), _f32(p))
        return t.value, p

    def num_levels(self):
        return lib().vo_tracker_num_levels(self._h)

    def level(self, lvl):
        r, c, n = C.c_int(), C.c_int(), C.c_int()
        intr = np.zeros(5, np.float32)
        lib().vo_tracker_level(self._h, lvl, C.byref(r), C.byref(c), C.byref(n), _f32(intr))
        return r.value, c.value, n.value, intr

    def image(self, lvl):
        r, c, _, _ = self.level(lvl)
        out = np.empty((r, c), np.uint8)
        lib().vo_tracker_get_image(self._h, lvl, _u8(out))
        return out

    def gradients(self, lvl):
        r, c, _, _ = self.level(lvl)
        gx = np.empty((r, c), np.int16)
        gy = np.empty((r, c), np.int16)
        g2 = np.empty((r, c), np.uint16)
        rc = lib().vo_tracker_get_gradients(self._h, lvl, gx.ctypes.data_as(C.POINTER(C.c_int16)),
                                            gy.ctypes.data_as(C.POINTER(C.c_int16)), _u16(g2))
        assert rc == 0
        return gx, gy, g2

    def mask(self):
        out = np.empty((self.rows, self.cols), np.uint8)
        assert lib().vo_tracker_get_mask(self._h, _u8(out)) == 0
        return out

    def points(self, lvl):
        """(xy int32[n,2], idepth f32[n], jac f32[n,6]) in the reference's column-major enumeration order."""
        _, _, n, _ = self.level(lvl)
        xy = np.empty((n, 2), np.int32)
        z = np.empty(n, np.float32)
        jac = np.empty((n, 6), np.float32)
        lib().vo_tracker_get_points(self._h, lvl, _i32(xy), _f32(z), _f32(jac))
        return xy, z, jac

    def last(self):
        L = self.cfg.nb_levels
        m = np.zeros(7, np.float32)
        flow = C.c_float()
        ch, ok = C.c_int(), C.c_int()
        it = np.zeros(L, np.int32)
        en = np.zeros(L, np.float32)
        lib().vo_tracker_last(self._h, _f32(m), C.byref(flow), C.byref(ch), C.byref(ok), _i32(it), _f32(en))
        return dict(lm_model=m, flow=flow.value, changed_keyframe=bool(ch.value), went_well=bool(ok.value), nb_iter=it,
                    energy=en)


def _opt(a, fn):
    return fn(a) if a is not None else None


def lm_eval(intr5, tmpl, img, xy, idepth, jac, model7, huber_delta=0.0, want_residuals=False):
    """eval_energy + compute_eval_data (lm_optimizer.rs:68-107) -> energy, n_inside, g[6], H[6,6] (, residuals)."""
    rows, cols = tmpl.shape
    n = len(idepth)
    intr5 = np.ascontiguousarray(intr5, np.float32)
    xy = np.ascontiguousarray(xy, np.int32)
    idepth = np.ascontiguousarray(idepth, np.float32)
    jac = np.ascontiguousarray(jac, np.float32)
    model7 = np.ascontiguousarray(model7, np.float32)
    e = C.c_float()
    ni = C.c_int32()
    g = np.zeros(6, np.float32)
    H = np.zeros((6, 6), np.float32)
    res = np.zeros(n, np.float32) if want_residuals else None
    f = lib().vo_lm_eval
    f.restype = None
    f(_f32(intr5), _u8(np.ascontiguousarray(tmpl)), _u8(np.ascontiguousarray(img)), rows, cols, n, _i32(xy), _f32(idepth),
      _f32(jac), C.c_float(huber_delta), _f32(model7), C.byref(e), C.byref(ni), _f32(g), _f32(H), _opt(res, _f32))
    if want_residuals:
        return e.value, ni.value, g, H, res
    return e.value, ni.value, g, H


def lm_solve(intr5, tmpl, img, xy, idepth, jac, model7, huber_delta=0.0):
    """iterative_solve at one level (optimizer.rs:57-70) -> status, model7, nb_iter, energy, lm_coef."""
    rows, cols = tmpl.shape
    n = len(idepth)
    intr5 = np.ascontiguousarray(intr5, np.float32)
    xy = np.ascontiguousarray(xy, np.int32)
    idepth = np.ascontiguousarray(idepth, np.float32)
    jac = np.ascontiguousarray(jac, np.float32)
    model7 = np.ascontiguousarray(model7, np.float32)
    out = np.zeros(7, np.float32)
    it = C.c_int32()
    e = C.c_float()
    lam = C.c_float()
    st = lib().vo_lm_solve(_f32(intr5), _u8(np.ascontiguousarray(tmpl)), _u8(np.ascontiguousarray(img)), rows, cols, n,
                           _i32(xy), _f32(idepth), _f32(jac), C.c_float(huber_delta), _f32(model7), _f32(out), C.byref(it),
                           C.byref(e), C.byref(lam))
    return st, out, it.value, e.value, lam.value


def lm_step(H, g, model7, lm_coef):
    H = np.ascontiguousarray(H, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    model7 = np.ascontiguousarray(model7, np.float32)
    out = np.zeros(7, np.float32)
    delta = np.zeros(6, np.float32)
    st = lib().vo_lm_step(_f32(H), _f32(g), _f32(model7), C.c_float(lm_coef), _f32(out), _f32(delta))
    return st, out, delta


def track_pairs(cfg, kf_gray, kf_depth, cur_gray, cur_depth=None, init_poses7=None, n_threads=1, variant=None):
    """Per pair: Config::init(keyframe) + Tracker::track(current). Returns dict of arrays.
    variant: None = the oracle; otherwise a sensitivity / baseline build (see variant_lib)."""
    n, rows, cols = kf_gray.shape
    L = cfg.nb_levels
    poses = np.zeros((n, 7), np.float32)
    models = np.zeros((n, 7), np.float32)
    status = np.zeros(n, np.int32)
    nb_iter = np.zeros((n, L), np.int32)
    n_points = np.zeros((n, L), np.int32)
    flow = np.zeros(n, np.float32)
    kf_gray = np.ascontiguousarray(kf_gray, np.uint8)
    kf_depth = np.ascontiguousarray(kf_depth, np.uint16)
    cur_gray = np.ascontiguousarray(cur_gray, np.uint8)
    if cur_depth is not None:
        cur_depth = np.ascontiguousarray(cur_depth, np.uint16)
    if init_poses7 is not None:
        init_poses7 = np.ascontiguousarray(init_poses7, np.float32)
    (variant_lib(variant) if variant else lib()).vo_track_pairs(C.byref(cfg), n, _u8(kf_gray), _u16(kf_depth), _u8(cur_gray), _opt(cur_depth, _u16), rows, cols,
                         _opt(init_poses7, _f32), _f32(poses), _i32(status), _f32(models), _i32(nb_iter), _i32(n_points),
                         _f32(flow), n_threads)
    return dict(poses=poses, models=models, status=status, nb_iter=nb_iter, n_points=n_points, flow=flow)


def track_sequences(cfg, gray, depth, n_threads=1, variant=None):
    """n_seq independent sequences, one Tracker each (vors_track.rs:46-62 per sequence). gray / depth: [n_frames, n_seq, rows, cols]
    (frame-major, as a lock-step host holds them). -> poses [n_seq, n_frames-1, 7], status, changed_keyframe [n_seq, n_frames-1]."""
    gray = np.ascontiguousarray(gray, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    F, n, rows, cols = gray.shape
    poses = np.zeros((n, F - 1, 7), np.float32)
    status = np.zeros((n, F - 1), np.int32)
    switch = np.zeros((n, F - 1), np.int32)
    (variant_lib(variant) if variant else lib()).vo_track_sequences(C.byref(cfg), n, F, _u8(gray), _u16(depth), rows, cols, _f32(poses), _i32(status),
                                                                     _i32(switch), n_threads)
    return dict(poses=poses, status=status, changed_keyframe=switch)


def mean_pyramid(img, max_levels):
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    out = np.empty(rows * cols * 2, np.uint8)
    r = np.zeros(max(max_levels, 1), np.int32)
    c = np.zeros(max(max_levels, 1), np.int32)
    n = lib().vo_mean_pyramid(_u8(img), rows, cols, max_levels, _u8(out), _i32(r), _i32(c))
    levels, off = [], 0
    for l in range(n):
        levels.append(out[off:off + r[l] * c[l]].reshape(r[l], c[l]).copy())
        off += r[l] * c[l]
    return levels


def dso_mask(img, nb_target=2000, seed=0x5EEDD50):
    """dso::select with the parameters of examples/candidates_dso.rs -> (mask uint8[rows, cols], block sizes of the rounds)."""
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    mask = np.zeros((rows, cols), np.uint8)
    bs = np.zeros(3, np.int32)
    f = lib().vo_dso_mask
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
    n = f(_u8(img), rows, cols, nb_target, seed, _u8(mask), _i32(bs))
    return mask, list(bs[:n])


def libm_sincos(x):
    """std::sin / std::cos on float32, as the oracle's se3::exp calls them (the platform libm) -> (sin, cos)."""
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().vo_libm_sincos(_f32(x), x.size, _f32(s), _f32(c))
    return s, c


def prune_with_thresh(thresh, a, b, c, d):
    out = np.zeros(4, np.uint8)
    lib().vo_prune_with_thresh(thresh, a, b, c, d, _u8(out))
    return [bool(x) for x in out]


def _vecfn(name, n_in, n_out):
    def f(x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        assert x.size == n_in
        out = np.zeros(n_out, np.float32)
        getattr(lib(), name)(_f32(x), _f32(out))
        return out
    return f


se3_exp = _vecfn("vo_se3_exp", 6, 7)
se3_log = _vecfn("vo_se3_log", 7, 6)
se3_hat = _vecfn("vo_se3_hat", 6, 16)
se3_vee = _vecfn("vo_se3_vee", 16, 6)
so3_exp = _vecfn("vo_so3_exp", 3, 4)
so3_log = _vecfn("vo_so3_log", 4, 3)
so3_hat = _vecfn("vo_so3_hat", 3, 9)
so3_hat_2 = _vecfn("vo_so3_hat_2", 3, 9)
so3_vee = _vecfn("vo_so3_vee", 9, 3)
iso_inverse = _vecfn("vo_iso_inverse", 7, 7)


def iso_mul(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(7, np.float32)
    lib().vo_iso_mul(_f32(a), _f32(b), _f32(out))
    return out


def intrinsics_multires(intr5, n):
    intr5 = np.ascontiguousarray(intr5, np.float32)
    out = np.zeros((n, 5), np.float32)
    lib().vo_intrinsics_multires(_f32(intr5), n, _f32(out))
    return out
