"""Development aid (gpurun): does a ring measurement depend on what the process did before (sustained load, live handles)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
def cfg_of(mode, arith):
    return V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
def ring(tag, n=512, depth=3, K=100):
    cfg = cfg_of(0, V.ARITH_REFERENCE)
    sets = []
    for k in range(depth):
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 + k * n, n, rows, cols, intr)
        sets.append((kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")))
    pipe = V.Pipeline(cfg, n, rows, cols, depth=depth)
    def step(i):
        s = sets[i % depth]; pipe.submit(s[0], s[1], s[2], s[3], s[4])
    for i in range(2 * depth): step(i)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    pipe.drain(); torch.cuda.synchronize()
    print(f"{tag}: REFERENCE c2f 512 pairs ring {depth}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step", flush=True)
ring("fresh process")
n = 4096
kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000, n, rows, cols, intr)
poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
big = V.Batch(cfg_of(1, V.ARITH_FUSED), n, rows, cols)
big.enable_kernel_timing(32)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 20.0:
    for _ in range(20): big.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
ring("after 20 s of dense FUSED 4096-pair steps, that handle alive")
ring("again")
del big
ring("that handle destroyed")
time.sleep(10)
ring("after 10 s idle")
