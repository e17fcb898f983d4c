"""Development aid (run through gpurun with VORS_HIP_LIB=.../libvors_hip_rtiming.so, built by tools/build_ref_variant.sh timing
-DVORS_REFW_TIMING): where a wavefront of lm_ref_track_kernel (REFERENCE arithmetic, one wavefront per frame pair) spends its cycles —
evaluations, step(), the rest — per pair, for a small batch (latency) and a large one (throughput).
usage: python tools/ref_profile.py [mode ...] [batches...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

lib = V.lib()
prof = getattr(lib, "vors_debug_refw_profile", None)
lprof = getattr(lib, "vors_debug_refw_level_profile", None)
modes = [a for a in sys.argv[1:] if not a.isdigit()] or ["c2f", "dso"]
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 4096]
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
for name in modes:
    mode = {"c2f": 0, "dense": 1, "dso": 2}[name]
    for n in batches:
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if {"dso": 1, "c2f": 0}.get(os.environ.get("SCENE"), mode == 2) else 0), n, rows, cols, intr)
        poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
        b = V.Batch(cfg, n, rows, cols)
        b.enable_kernel_timing(32)
        for _ in range(int(os.environ.get("WARM", "2"))):
            b.track_pairs(kg, kd, cg, poses, status)
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        lout = (ctypes.c_ulonglong * 24)()
        if prof:
            prof(out, 1)
        if lprof:
            lprof(lout, 1)
        reps = int(os.environ.get("REPS", "5"))
        for _ in range(reps):
            b.track_pairs(kg, kd, cg, poses, status)
        torch.cuda.synchronize()
        lm = float(b.kernel_times("lm")[-reps:].mean())
        line = f"{name:5s} {n:5d} pairs: lm {lm:7.3f} ms"
        if prof:
            prof(out, 1)
            ev, st, gr, ne, kc, nw = [out[i] / reps / n for i in range(6)]
            line += (f" | per pair: kernel {kc / 1e3:7.1f} kcyc = eval {ev / 1e3:7.1f} + step {st / 1e3:6.1f} + rest {(kc - ev - st) / 1e3:6.1f};"
                     f" {ne:5.1f} evals, {gr:7.1f} groups -> {ev / max(gr, 1):6.0f} cyc per group of 64, {st / max(ne, 1):6.0f} cyc per step")
            if out[7]:  # the longest wavefront's cycles (atomicMax over all launches) against the launch's duration = the shader clock
                line += f"; longest wavefront {out[7] / 1e3:7.1f} kcyc -> {out[7] / lm / 1e6:5.2f} GHz if it spans the launch"
        print(line, flush=True)
        if lprof:
            lprof(lout, 1)
            for l in range(8):
                cyc, gr, ne = [lout[l * 3 + i] / reps / n for i in range(3)]
                if ne:
                    print(f"        width 2^{l + 4}..: {ne:5.1f} evals of {gr / ne:5.1f} groups, {cyc / ne:8.0f} cyc per eval, {cyc / gr:6.0f} per group", flush=True)
        del b
