"""Development aid (gpurun): steps of a continuous feed alternating between `depth` batch handles on internal streams (vors_pipeline_*: the C
ABI's throughput mode) against the single-stream step, per candidate mode, arithmetic and batch size — what the small batches of BASELINE
config 4 (512 pairs per GPU) gain when consecutive steps overlap.    usage: python tools/pipeline_small_batches.py [arith ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
for aname in (sys.argv[1:] or ["reference", "fused"]):
    arith = {"fused": V.ARITH_FUSED, "reference": V.ARITH_REFERENCE, "exact": V.ARITH_EXACT}[aname]
    for mode, mname in ((0, "c2f"), (2, "dso"), (1, "dense")):
        res = {}
        for n in (512, 4096):
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
            sets = []
            for k in range(3):
                kg, kd, cg, _, _ = V.synth_render_pairs((0x5EED0000 + k * n) | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
                sets.append((kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")))
            K = 24 if (mode != 1 or arith != V.ARITH_REFERENCE) else 6
            row = []
            for depth in (1, 2, 3):
                if depth == 1:
                    b = V.Batch(cfg, n, rows, cols)
                    def step(i):
                        s = sets[i % 3]; b.track_pairs(s[0], s[1], s[2], s[3], s[4])
                    fin = torch.cuda.synchronize
                else:
                    pipe = V.Pipeline(cfg, n, rows, cols, depth=depth)
                    def step(i):
                        s = sets[i % 3]; pipe.submit(s[0], s[1], s[2], s[3], s[4])
                    def fin():
                        pipe.drain(); torch.cuda.synchronize()
                for i in range(6): step(i)
                fin()
                t0 = time.perf_counter()
                for i in range(K): step(i)
                fin()
                row.append((time.perf_counter() - t0) / K * 1e3)
                if depth == 1: del b
                else: del pipe
            res[n] = row
            print(f"{aname:9s} {mname:5s} {n:5d} pairs: single stream {row[0]:7.3f} ms | ring of 2 {row[1]:7.3f} ms | ring of 3 {row[2]:7.3f} ms per step", flush=True)
        print(f"{aname:9s} {mname:5s} 4096 / 512 step-time ratio: single {res[4096][0] / res[512][0]:.2f} | ring of 2 {res[4096][1] / res[512][1]:.2f} | ring of 3 {res[4096][2] / res[512][2]:.2f}", flush=True)
