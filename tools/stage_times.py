"""Development aid (run through gpurun): stage times (HIP events on the stream) of a step for several batch sizes and the three candidate
modes, and the ratio of the step times — BASELINE config 4 as written is 4096 pairs over 8 GPUs, i.e. 512 pairs per GPU: the 512-pair
step must take at most 1/6 of the 4096-pair step for the >= 6x target.    usage: [MODES=c2f,dso] python tools/stage_times.py [arith] [batches...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

arith = {"fused": V.ARITH_FUSED, "exact": V.ARITH_EXACT, "reference": V.ARITH_REFERENCE}[sys.argv[1] if len(sys.argv) > 1 else "fused"]
batches = [int(x) for x in sys.argv[2:]] or [512, 4096]
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
for mode, name in ((0, "c2f"), (2, "dso"), (1, "dense")):
    if os.environ.get("MODES") and name not in os.environ["MODES"].split(","):
        continue
    res = {}
    for n in batches:
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
        poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
        b = V.Batch(cfg, n, rows, cols)
        b.enable_kernel_timing(32)
        for _ in range(3):
            b.track_pairs(kg, kd, cg, poses, status)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            b.track_pairs(kg, kd, cg, poses, status)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = {k: float(b.kernel_times(k)[-reps:].mean()) for k in ("pyramid_keyframe", "keyframe", "pyramid_current", "lm")}
        res[n] = ms
        print(f"{name:5s} {n:5d} pairs: step {ms:7.3f} ms ({n / ms:7.1f} k pairs/s) | pyr_kf {st['pyramid_keyframe']:.3f} keyframe {st['keyframe']:.3f} "
              f"pyr_cur {st['pyramid_current']:.3f} lm {st['lm']:.3f} | sum {sum(st.values()):.3f}", flush=True)
        del b
    if len(batches) >= 2:
        print(f"{name:5s} step-time ratio {batches[-1]} / {batches[0]} pairs: {res[batches[-1]] / res[batches[0]]:.2f} (target >= {batches[-1] / batches[0] * 0.75:.1f})")
