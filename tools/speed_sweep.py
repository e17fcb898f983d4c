"""Development aid (run through gpurun): do the scheduling defaults (threads per pair, evaluation rounds, chunks) hold away from the two
shapes they were fitted on?  For every (shape, batch size, candidates mode) the step time with the defaults and with each alternative
setting of one knob at a time; prints the best alternative and how far the default is from it.
usage: python tools/speed_sweep.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

SHAPES = [(240, 320, 5), (480, 640, 6), (960, 1280, 7)]
BATCHES = [64, 512, 4096]
KNOBS = {0: {"VORS_LM_BLOCK": ["64", "128", "256", "512"]},
         2: {"VORS_LM_BLOCK": ["128", "256", "512"]},
         1: {"VORS_LM_BLOCK": ["256", "512", "1024"], "VORS_LM_SPLIT_ROUNDS": ["10", "16", "26", "36"], "VORS_LM_CHUNKS": ["32", "64", "128", "256"]}}
if len(sys.argv) > 1:
    SHAPES, BATCHES = SHAPES[:2], [512, 4096]


def step_ms(cfg, n, rows, cols, data, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        b = V.Batch(cfg, n, rows, cols)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    kg, kd, cg, poses, status = data
    for _ in range(2):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rows, cols, L in SHAPES:
    intr = V.scaled_intrinsics(rows, cols)
    for n in BATCHES:
        if rows * cols * n > 640 * 480 * 4096:
            continue
        for mode in (0, 1, 2):
            kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
            data = (kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"))
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_FUSED)
            base = step_ms(cfg, n, rows, cols, data, {})
            best, best_ms = "default", base
            notes = []
            for knob, values in KNOBS[mode].items():
                for v in values:
                    try:
                        ms = step_ms(cfg, n, rows, cols, data, {knob: v})
                    except V.VorsError:
                        continue
                    notes.append(f"{knob[8:]}={v}: {ms:.3f}")
                    if ms < best_ms:
                        best, best_ms = f"{knob}={v}", ms
            print(f"{cols}x{rows} L{L} {n:5d} pairs mode {mode}: default {base:.3f} ms ({n / base:.0f} k pairs/s); best {best} {best_ms:.3f} ms "
                  f"({(base / best_ms - 1) * 100:+.1f} % over the default) | " + ", ".join(notes), flush=True)
            del data, kg, kd, cg
