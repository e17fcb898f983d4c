"""TUM plumbing of the C++ host: parser / trajectory format / PNG codec self-test on CPU, and the
`vors_track` CLI end to end on a synthetic TUM-format sequence on the GPU (compared with the oracle tracker)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "visual-odometry-rs_amd", "host")


def _ensure_host_built():
    if not all(os.path.exists(os.path.join(HOST, b)) for b in ("host_plumbing_test", "vors_track", "host_selftest")):
        subprocess.check_call(["make", "-C", HOST, "-s"])


def _write_png(path, arr, level=6):
    """8-bit grey (uint8) or 16-bit big-endian grey (uint16) PNG, filter 0."""
    h, w = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    raw = arr.astype(">u2").tobytes() if depth == 16 else arr.tobytes()
    stride = len(raw) // h
    scan = b"".join(b"\x00" + raw[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(scan, level)) + chunk(b"IEND", b""))


def test_host_plumbing_selftest(tmp_path):
    _ensure_host_built()
    fixtures = os.path.join(ROOT, "tests", "golden", "png")   # written by Pillow (independent encoder, filters 0-4)
    out = subprocess.run([os.path.join(HOST, "host_plumbing_test"), str(tmp_path), fixtures], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "png fixtures: ok" in out.stdout and "host_plumbing_test: ok" in out.stdout


def test_cli_argument_errors_match_reference_messages():
    _ensure_host_built()
    exe = os.path.join(HOST, "vors_track")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert "Usage: ./vors_track [fr1|fr2|fr3|icl] associations_file" in r.stderr and "Wrong number of arguments" in r.stderr
    r = subprocess.run([exe, "fr9", "x"], capture_output=True, text=True)
    assert "Unknown camera id: fr9" in r.stderr
    r = subprocess.run([exe, "fr1", "/nonexistent/assoc.txt"], capture_output=True, text=True)
    assert "The association file does not exist or is not reachable" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_host_selftest_tracks_on_gpu():
    _ensure_host_built()
    r = subprocess.run([os.path.join(HOST, "host_selftest")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["exact", "fused", "reference"])
def test_vors_track_cli_on_synthetic_tum_sequence(tmp_path, arith):
    """BASELINE configs[0]'s shape (TUM-format sequence, first 20 frame pairs, the reference's coarse-to-fine candidates):
    associations.txt + 16-bit big-endian depth PNGs + 8-bit grey PNGs -> trajectory lines, vs the oracle Tracker, in both arithmetics.
    (The real fr1/xyz is not available offline: a synthetic sequence in the same on-disk format stands in.)"""
    _ensure_host_built()
    rows, cols, n = 480, 640, 21
    intr = O.INTRINSICS_FR1
    os.makedirs(tmp_path / "depth")
    os.makedirs(tmp_path / "rgb")
    step = np.array([0.010, -0.004, 0.003, 0.0015, -0.002, 0.001])
    frames, lines = [], ["# depth_timestamp depth_file_path rgb_timestamp rgb_file_path"]
    for k in range(n):
        g, d = O.synth_frame(4242, step * k, rows, cols, intr, frame_salt=k)
        td, tc = 1305031102.160407 + 0.033 * k, 1305031102.175304 + 0.033 * k
        _write_png(str(tmp_path / "depth" / f"{td:.6f}.png"), d)
        _write_png(str(tmp_path / "rgb" / f"{tc:.6f}.png"), g)
        lines.append(f"{td:.6f} depth/{td:.6f}.png {tc:.6f} rgb/{tc:.6f}.png")
        frames.append((float(f"{td:.6f}"), d, float(f"{tc:.6f}"), g))
    assoc = tmp_path / "associations.txt"
    assoc.write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(HOST, "vors_track"), "fr1", str(assoc), "--quiet", "--arith", arith], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(out) == n - 1                                   # poses for frames 1..n-1 only (vors_track.rs:49-64)
    ot = O.Tracker(O.make_config(6, intr), frames[0][0], frames[0][1], frames[0][2], frames[0][3])
    for k in range(1, n):
        td, d, tc, g = frames[k]
        ot.track(td, d, tc, g)
        t, pose = ot.current_frame()
        assert out[k - 1][0] == repr(td)                        # depth timestamp, shortest round-trip digits
        got = np.array([float(x) for x in out[k - 1][1:]], np.float32)
        assert np.abs(got - pose).max() < 1e-4, (k, got, pose)
        if arith == "reference":   # the reference's summation order: the printed trajectory IS the oracle's, digit for digit
            assert (got.view(np.uint32) == np.asarray(pose, np.float32).view(np.uint32)).all(), (k, got, pose)
        assert all("e" not in x.lower() for x in out[k - 1])    # positional notation, like Rust's Display


@pytest.mark.gpu
def test_vors_track_cli_config3_at_full_length_600_frames_dso_reference_arithmetic(tmp_path):
    """BASELINE configs[2] as written — a full-length TUM-format sequence (fr1/desk has ~600 frames; not available offline, so a synthetic
    sequence in the same on-disk format stands in) with DSO candidate selection — through the vors_track CLI (src/bin/vors_track.rs:46-64)
    in the REFERENCE arithmetic: all 600 trajectory lines equal the oracle tracker's poses bit for bit."""
    _ensure_host_built()
    import vors_amd as V
    rows, cols, n = 480, 640, 601
    intr = O.INTRINSICS_FR1
    os.makedirs(tmp_path / "depth")
    os.makedirs(tmp_path / "rgb")
    step = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
    frames, lines = [], ["# depth_timestamp depth_file_path rgb_timestamp rgb_file_path"]
    for k0 in range(0, n, 64):
        ks = list(range(k0, min(n, k0 + 64)))
        g, d = V.synth_render_frames([(1 << 63) | 4242] * len(ks), ks, [step * k for k in ks], rows, cols, intr)
        g, d = g.cpu().numpy(), d.cpu().numpy().view(np.uint16)
        for j, k in enumerate(ks):
            td, tc = 1305031102.160407 + 0.033 * k, 1305031102.175304 + 0.033 * k
            _write_png(str(tmp_path / "depth" / f"{td:.6f}.png"), d[j], level=1)
            _write_png(str(tmp_path / "rgb" / f"{tc:.6f}.png"), g[j], level=1)
            lines.append(f"{td:.6f} depth/{td:.6f}.png {tc:.6f} rgb/{tc:.6f}.png")
            frames.append((float(f"{td:.6f}"), d[j].copy(), float(f"{tc:.6f}"), g[j].copy()))
    assoc = tmp_path / "associations.txt"
    assoc.write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(HOST, "vors_track"), "fr1", str(assoc), "--quiet", "--arith", "reference", "--candidates", "dso"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(out) == n - 1
    ot = O.Tracker(O.make_config(6, intr, candidates_mode=2), frames[0][0], frames[0][1], frames[0][2], frames[0][3], keep_debug=False)
    switches = 0
    for k in range(1, n):
        td, d, tc, g = frames[k]
        ot.track(td, d, tc, g)
        switches += int(ot.last()["changed_keyframe"])
        got = np.array([float(x) for x in out[k - 1][1:]], np.float32)
        pose = np.asarray(ot.current_frame()[1], np.float32)
        assert out[k - 1][0] == repr(td)
        assert (got.view(np.uint32) == pose.view(np.uint32)).all(), f"frame {k}: {np.abs(got - pose).max():.3e}"
    assert switches >= 10


@pytest.mark.gpu
def test_vors_track_cli_stderr_lines_are_the_reference_s(tmp_path):
    """Without --quiet the CLI prints what the reference's eprintln!s print, with Rust's float Display (shortest round-trip digits,
    positional notation): `Optical_flow: {}` per tracked frame (inverse_compositional.rs:222) and `Changing keyframe after: {} seconds`
    on a switch (:228-229, f64 difference of the depth timestamps). Checked against the oracle tracker in the REFERENCE arithmetic,
    where the optical flow is bit-identical."""
    _ensure_host_built()
    rows, cols, n = 480, 640, 26
    intr = O.INTRINSICS_FR1
    os.makedirs(tmp_path / "depth")
    os.makedirs(tmp_path / "rgb")
    step = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    frames, lines = [], []
    for k in range(n):
        g, d = O.synth_frame(77, step * k, rows, cols, intr, frame_salt=k)
        td, tc = 1305031102.160407 + 0.033 * k, 1305031102.175304 + 0.033 * k
        _write_png(str(tmp_path / "depth" / f"{td:.6f}.png"), d)
        _write_png(str(tmp_path / "rgb" / f"{tc:.6f}.png"), g)
        lines.append(f"{td:.6f} depth/{td:.6f}.png {tc:.6f} rgb/{tc:.6f}.png")
        frames.append((float(f"{td:.6f}"), d, float(f"{tc:.6f}"), g))
    assoc = tmp_path / "associations.txt"
    assoc.write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(HOST, "vors_track"), "fr1", str(assoc), "--arith", "reference"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ot = O.Tracker(O.make_config(6, O.INTRINSICS_FR1), frames[0][0], frames[0][1], frames[0][2], frames[0][3])
    expect, kf_t, switches = [], frames[0][0], 0
    for k in range(1, n):
        td, d, tc, g = frames[k]
        st = ot.track(td, d, tc, g)
        last = ot.last()
        if st != 0:
            expect.append("Error at Cholesky decomposition of hessian")
        expect.append("Optical_flow: " + np.format_float_positional(np.float32(last["flow"]), unique=True, trim="-"))
        if last["changed_keyframe"]:
            expect.append("Changing keyframe after: " + np.format_float_positional(np.float64(td - kf_t), unique=True, trim="-") + " seconds")
            kf_t = td
            switches += 1
    got = [l for l in r.stderr.splitlines() if l.strip()]
    assert got == expect, "\n".join(f"{a!r} | {b!r}" for a, b in zip(got, expect) if a != b)
    assert switches >= 1
