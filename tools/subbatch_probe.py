"""Development aid (gpurun): what splitting ONE step of n pairs into K sub-batches on K streams (each with a handle of its own, joined at the end
of the step) would gain — the in-call form of the throughput mode, emulated with torch streams.  usage: python tools/subbatch_probe.py [arith]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
aname = sys.argv[1] if len(sys.argv) > 1 else "reference"
arith = {"fused": V.ARITH_FUSED, "reference": V.ARITH_REFERENCE}[aname]
for mode, mname in ((0, "c2f"), (2, "dso"), (1, "dense")):
    for n in (4096, 512):
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
        poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
        row = []
        for K in (1, 2, 4):
            m = n // K
            hs = [V.Batch(cfg, m, rows, cols) for _ in range(K)]
            ss = [torch.cuda.Stream() for _ in range(K)]
            def step():
                cur = torch.cuda.current_stream()
                for k in range(K):
                    ss[k].wait_stream(cur)
                    with torch.cuda.stream(ss[k]):
                        sl = slice(k * m, (k + 1) * m)
                        hs[k].track_pairs(kg[sl], kd[sl], cg[sl], poses[sl], status[sl])
                for k in range(K):
                    cur.wait_stream(ss[k])
            for _ in range(3): step()
            torch.cuda.synchronize()
            reps = 20 if (mode != 1 or arith != V.ARITH_REFERENCE) else 5
            t0 = time.perf_counter()
            for _ in range(reps): step()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / reps * 1e3)
            del hs
        print(f"{aname} {mname:5s} {n:5d} pairs per step: one handle {row[0]:.3f} ms | 2 sub-batches {row[1]:.3f} | 4 sub-batches {row[2]:.3f}", flush=True)
