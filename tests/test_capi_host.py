"""CPU-side checks of the product library: it loads, exports every symbol include/vors_hip.h declares, refuses to
compute without a GPU, and its host-only arithmetic (Lie helpers, LM step) matches the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vors_amd as V
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "vors_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vors_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = V.lib()
    names = _declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(V.EXPORTED_SYMBOLS)
    assert lib.vors_abi_version() == 5   # 5: VORS_ARITH_* renumbered, 0 = REFERENCE


def test_struct_layouts_match_header():
    assert C.sizeof(V.vors_config) == 48
    assert C.sizeof(V.vors_pair_stats) == 7 * 4 + 4 + 4 + 4 * 8 * 4
    assert C.sizeof(O.Config) == C.sizeof(V.vors_config) - 4   # the oracle has ONE arithmetic (the reference's): no `arithmetic` field


def test_compute_entry_points_fail_loudly_without_gpu():
    if V.device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = V.Config(nb_levels=3)
    img = np.zeros((32, 32), np.uint8)
    dep = np.zeros((32, 32), np.uint16)
    with pytest.raises(V.VorsError, match="no HIP device"):
        cfg.init(0.0, dep, 0.0, img)
    with pytest.raises(V.VorsError, match="no HIP device"):
        V.track_pairs(cfg, img[None], dep[None], img[None])
    obs = V.Obs([16, 16, 30, 30, 0], img, img, np.zeros((1, 2), np.int32) + 5, np.ones(1, np.float32), np.ones((1, 6), np.float32))
    with pytest.raises(V.VorsError, match="no HIP device"):
        V.lm_eval(obs, [0, 0, 0, 0, 0, 0, 1])


def test_argument_validation_needs_no_gpu():
    img = np.zeros((1, 8, 8), np.uint8)
    dep = np.zeros((1, 8, 8), np.uint16)
    with pytest.raises(V.VorsError, match="image too small"):
        V.track_pairs(V.Config(nb_levels=6), img, dep, img)  # the reference panics here
    with pytest.raises(V.VorsError, match="nb_levels"):
        V.track_pairs(V.Config(nb_levels=0), img, dep, img)
    with pytest.raises(V.VorsError, match="candidates_mode"):
        V.track_pairs(V.Config(nb_levels=2, candidates_mode=7), img, dep, img)


def test_lie_helpers_bit_exact_vs_oracle():
    rng = np.random.default_rng(3)
    for scale in (1e-3, 5e-3, 0.2, 1.5):
        for _ in range(50):
            xi = (rng.uniform(-1, 1, 6) * scale).astype(np.float32)
            a = V.se3_exp(xi)
            assert (a.view(np.uint32) == O.se3_exp(xi).view(np.uint32)).all()
            assert (V.se3_log(a).view(np.uint32) == O.se3_log(a).view(np.uint32)).all()
            b = V.se3_exp((rng.uniform(-1, 1, 6) * scale).astype(np.float32))
            assert (V.iso_mul(a, b).view(np.uint32) == O.iso_mul(a, b).view(np.uint32)).all()
            assert (V.iso_inverse(a).view(np.uint32) == O.iso_inverse(a).view(np.uint32)).all()
            q = V.so3_exp(xi[3:])
            assert (q.view(np.uint32) == O.so3_exp(xi[3:]).view(np.uint32)).all()
            assert (V.so3_log(q).view(np.uint32) == O.so3_log(q).view(np.uint32)).all()


def test_lm_step_bit_exact_vs_oracle_and_cholesky_failure():
    rng = np.random.default_rng(4)
    for _ in range(100):
        J = rng.normal(size=(40, 6)).astype(np.float32) * rng.uniform(0.1, 100, 6).astype(np.float32)
        H = (J.T @ J).astype(np.float32)
        H = ((H + H.T) / 2).astype(np.float32)
        g = rng.normal(size=6).astype(np.float32) * 10
        model = V.se3_exp((rng.uniform(-1, 1, 6) * 0.05).astype(np.float32))
        ok, out = V.lm_step(H, g, model, 0.1)
        st, oout, _ = O.lm_step(H, g, model, 0.1)
        assert ok and st == 0
        assert (out.view(np.uint32) == oout.view(np.uint32)).all()
    # singular / NaN hessian -> "Error at Cholesky decomposition of hessian" (lm_optimizer.rs:131-133)
    for bad in (np.zeros((6, 6), np.float32), np.full((6, 6), np.nan, np.float32), -np.eye(6, dtype=np.float32)):
        ok, _ = V.lm_step(bad, np.ones(6, np.float32), [0, 0, 0, 0, 0, 0, 1], 0.1)
        assert not ok
        assert O.lm_step(bad, np.ones(6, np.float32), [0, 0, 0, 0, 0, 0, 1], 0.1)[0] == 1


def test_optimizer_trait_skeleton_on_a_toy_problem():
    """State.iterative_solve (optimizer.rs:57-70) drives any init/step/eval/stop_criterion implementation."""

    class Halver(V.State):
        def __init__(self, x, e):
            self.x, self.e = x, e

        @classmethod
        def init(cls, obs, model):
            return cls(model, (model - obs) ** 2)

        def step(self):
            return (self.x + obs_target) / 2

        def eval(self, obs, new_model):
            return (new_model, (new_model - obs) ** 2)

        def stop_criterion(self, nb_iter, ev):
            x, e = ev
            return Halver(x, e), (V.Continue.Forward if self.e - e > 1e-6 and nb_iter < 50 else V.Continue.Stop)

    obs_target = 3.0
    state, n = Halver.iterative_solve(obs_target, 11.0)
    assert abs(state.x - 3.0) < 2e-3 and 5 < n < 50


def test_multi_gpu_entry_fails_loudly_without_gpu():
    if V.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(V.VorsError, match="no HIP device"):
        V.MultiGpu(V.Config(nb_levels=3), 4, 32, 32)
    with pytest.raises(V.VorsError, match="no HIP device"):
        V.Batch(V.Config(nb_levels=3), 4, 32, 32, device=0)


def test_ref_sincos_equals_the_platform_libm_for_every_f32_in_its_range():
    """se3::exp's sinf / cosf (se3.rs:82-87; Rust's f32::sin / cos = the platform libm = glibc's algorithm) are restated in csrc/lie.h so
    that host and DEVICE evaluate them identically (ocml's sinf differs from glibc's in ~1 % of the arguments, and glibc's is not the
    correctly rounded sine either). The restatement must equal the oracle's std::sin / std::cos for EVERY float32 in [0, 4): the whole
    range is enumerated here (2^30 + ... values; ~1 s per 2^24)."""
    lo = np.array([0.0], np.float32).view(np.uint32)[0]
    hi = np.array([4.0], np.float32).view(np.uint32)[0]
    step = 1 << 24
    bad = 0
    for start in range(int(lo), int(hi), step):
        bits = np.arange(start, min(start + step, int(hi)), dtype=np.uint32)
        x = bits.view(np.float32)
        if x[-1] < 2.0 ** -14:   # tiny arguments: sin x = x, cos x = 1 in both (se3::exp never gets below 5e-3); sample them
            x = x[:: 4096]
        s1, c1 = V.ref_sincos(x)
        s2, c2 = O.libm_sincos(x)
        bad += int((s1.view(np.uint32) != s2.view(np.uint32)).sum()) + int((c1.view(np.uint32) != c2.view(np.uint32)).sum())
    assert bad == 0
    # beyond the restated range the platform function is used: still equal
    x = np.linspace(4.0, 50.0, 10001).astype(np.float32)
    s1, c1 = V.ref_sincos(x)
    s2, c2 = O.libm_sincos(x)
    assert (s1 == s2).all() and (c1 == c2).all()
