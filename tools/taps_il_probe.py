"""Development aid (run through gpurun): the row-interleaved copy of the current pyramid (experiment, VORS_TAPS_IL=<first level that has one>) —
stage times of a REFERENCE step in the candidate-list modes and a hash of the poses / statuses / iteration counts (must not move).
usage: [VORS_TAPS_IL=1] python tools/taps_il_probe.py [pairs=4096]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
print(f"VORS_TAPS_IL={os.environ.get('VORS_TAPS_IL')}  {n} pairs", flush=True)
for mode, name in ((0, "c2f"), (2, "dso")):
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    poses, status, stats = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"), V.stats_tensor(n)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
    b = V.Batch(cfg, n, rows, cols)
    b.enable_kernel_timing(32)
    for _ in range(3):
        b.track_pairs(kg, kd, cg, poses, status, stats)
    torch.cuda.synchronize()
    h = zlib.crc32(poses.cpu().numpy().tobytes() + status.cpu().numpy().tobytes() + stats.cpu().numpy().tobytes())
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = {k: float(b.kernel_times(k)[-reps:].mean()) for k in ("pyramid_keyframe", "keyframe", "pyramid_current", "lm")}
    print(f"{name:5s} step {ms:7.3f} ms | pyr_kf {st['pyramid_keyframe']:.3f} keyframe {st['keyframe']:.3f} pyr_cur {st['pyramid_current']:.3f} "
          f"lm {st['lm']:.3f} | crc {h:08x} | workspace {b.workspace_bytes() / 2**30:.2f} GiB", flush=True)
    del b
