"""Development aid (gpurun, VORS_HIP_LIB=.../libvors_hip_rtiming.so from tools/build_ref_variant.sh timing -DVORS_REFW_TIMING): where wavefront 0
of lm_ref_track_coop_kernel (REFERENCE arithmetic, a workgroup per frame pair) spends its cycles.  usage: VORS_REF_COOP=4 python tools/ref_profile_coop.py [pairs]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
lib = V.lib()
prof = lib.vors_debug_refw_profile
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, mode in (("c2f", 0), ("dso", 2)):
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
    b = V.Batch(cfg, n, rows, cols)
    b.enable_kernel_timing(32)
    for _ in range(2):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    prof(out, 1)
    reps = 5
    for _ in range(reps):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    lm = float(b.kernel_times("lm")[-reps:].mean())
    prof(out, 1)
    o = [out[i] / reps / n / 1e3 for i in range(8)]
    print(f"{name} {n} pairs: lm {lm:.3f} ms | wavefront 0 per pair, kcyc: kernel {o[4]:.1f} = summing {o[0]:.1f} + verdict/step/publish {o[6]:.1f} (step alone {o[1]:.1f})"
          f" + at barriers {o[7]:.1f} + rest {o[4] - o[0] - o[6] - o[7]:.1f}; barriers {o[2] * 1e3:.0f}", flush=True)
