"""Development aid (gpurun, VORS_HIP_LIB=.../libvors_hip_rtiming.so from tools/build_ref_variant.sh timing -DVORS_REFW_TIMING): where the wavefronts of
lm_ref_track_coop_kernel (REFERENCE arithmetic, a workgroup per frame pair) spend their cycles — one producer and wavefront 0, accumulated in
registers (one atomic per bucket at the end of the kernel).  usage: VORS_REF_COOP=5 python tools/ref_profile_coop.py [pairs] [modes...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
prof = V.lib().vors_debug_refc_profile
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
modes = sys.argv[2:] or ["c2f", "dso", "dense"]
for name in modes:
    mode = {"c2f": 0, "dense": 1, "dso": 2}[name]
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
    b = V.Batch(cfg, n, rows, cols)
    b.enable_kernel_timing(32)
    for _ in range(2):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 10)()
    prof(out, 1)
    reps = 3
    for _ in range(reps):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    lm = float(b.kernel_times("lm")[-reps:].mean())
    prof(out, 1)
    o = [float(v) for v in out]
    wg = max(o[9], 1)
    print(f"{name} {n} pairs: lm {lm:.3f} ms | per workgroup {o[8] / wg / 1e3:.0f} kcyc | a producer: {o[3] / wg:.0f} groups, per group wait+warp {o[0] / max(o[3], 1):.0f} "
          f"products {o[1] / max(o[3], 1):.0f} at the barrier {o[2] / max(o[3], 1):.0f} cyc | wavefront 0: {o[7] / wg:.0f} chunks, per chunk chains {o[4] / max(o[7], 1):.0f} "
          f"verdict+step {o[5] / max(o[7], 1):.0f} at the barrier {o[6] / max(o[7], 1):.0f} cyc", flush=True)
    del b
