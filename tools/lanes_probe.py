"""Development aid (run through gpurun): ONE step over a batch split into k lanes (k handles on k streams, fork / join with events),
with and without a stagger (lane i starts once lane i-1 has finished its keyframe stage, so that the bandwidth-bound stages of one lane
run under the VALU-bound LM stage of the previous one). Decides how vors_batch should split a batch internally.
usage: python tools/lanes_probe.py [dense|c2f|dso] [pairs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
mode_id = {"dense": 1, "c2f": 0, "dso": 2}[mode]
seed = 0x5EED0000 | ((1 << 63) if mode == "dso" else 0)
kg, kd, cg, _, _ = V.synth_render_pairs(seed, n, rows, cols, intr)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode_id, arithmetic=V.ARITH_FUSED)
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
ref = None
for k, stagger in ((1, 0), (2, 0), (2, 1), (3, 1), (4, 0), (4, 1), (8, 1)):
    per = (n + k - 1) // k
    lanes = [V.Batch(cfg, per, rows, cols) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    main = torch.cuda.current_stream()

    def step():
        fork = torch.cuda.Event(); fork.record(main)
        kf_done = None
        for i in range(k):
            lo, hi = i * per, min(n, (i + 1) * per)
            s = streams[i]
            s.wait_event(fork)
            if stagger and kf_done is not None:
                s.wait_event(kf_done)
            with torch.cuda.stream(s):
                lanes[i].prepare_keyframes(kg[lo:hi], kd[lo:hi])
                kf_done = torch.cuda.Event(); kf_done.record(s)
                lanes[i].track_current(cg[lo:hi], poses[lo:hi], status[lo:hi])
                j = torch.cuda.Event(); j.record(s)
            main.wait_event(j)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    p = poses.cpu().numpy()
    if ref is None:
        ref = p.copy()
    print(f"{mode} {n} pairs, {k} lane(s){' staggered' if stagger else ''}: {dt*1e3:.3f} ms/step = {n/dt:,.0f} pairs/s; "
          f"poses identical to 1 lane: {bool((p == ref).all())}", flush=True)
    del lanes
