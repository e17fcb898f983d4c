"""Development aid (run through gpurun): handle churn — batch handles, pipelines, batches on fresh streams — and the free device memory after
every 20: it must plateau (the HIP runtime keeps per-queue scratch; the handles themselves return everything)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 240, 320, 5
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1, arithmetic=V.ARITH_FUSED)
kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000, 64, rows, cols, intr)
outs = [(torch.zeros((64, 7), device="cuda"), torch.zeros(64, dtype=torch.int32, device="cuda")) for _ in range(6)]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for kind in ("batch", "pipeline", "batch_on_new_stream"):
    for it in range(80):
        if kind == "pipeline":
            h = V.Pipeline(cfg, 64, rows, cols, depth=2)
            for k in range(4): h.submit(kg, kd, cg, *outs[k])
            h.drain(host=True)
        elif kind == "batch":
            h = V.Batch(cfg, 64, rows, cols)
            for k in range(4): h.track_pairs(kg, kd, cg, *outs[k])
        else:
            s = torch.cuda.Stream()
            h = V.Batch(cfg, 64, rows, cols)
            with torch.cuda.stream(s):
                for k in range(4): h.track_pairs(kg, kd, cg, *outs[k])
            s.synchronize()
            del s
        torch.cuda.synchronize()
        del h
        if it % 20 == 19:
            torch.cuda.empty_cache()
            print(kind, it + 1, f"delta {(free0 - torch.cuda.mem_get_info()[0]) / 1e6:.1f} MB", flush=True)
