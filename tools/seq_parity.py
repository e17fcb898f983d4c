"""Development aid (run through gpurun): the 64-sequence parity statistic of bench.py sequences_64 over SEVERAL draws of the sequences, for
the variants of the FUSED arithmetic's small-level rule — is "1 of 64 coarse-to-fine sequences beyond 1e-4" a property of the rule or of the draw?
Per draw and mode: sequences whose worst frame is beyond 1e-4 of the oracle tracker (and the max) for FUSED (default: the reference's warp
chain on levels of few points), FUSED with VORS_FUSED_SMALL=exact (round 3's rule: the whole EXACT evaluation there), EXACT, and the oracle
against its own f64-accumulation build.      usage: python tools/seq_parity.py [c2f,dso] [draws=6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
from oracle import oracle as O

modes = (sys.argv[1] if len(sys.argv) > 1 else "c2f").split(",")
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows, cols, L, n_seq, n_frames = 480, 640, 6, 64, 40
intr = V.scaled_intrinsics(rows, cols)
base = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])


def run(cfg, frames, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        tr = V.Trackers(cfg, n_seq, rows, cols)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    tr.init(*frames[0])
    traj = []
    for k in range(1, n_frames):
        tr.track(*frames[k])
        traj.append(tr.current_frames()[0])
    return np.stack(traj, axis=1)


tot = {}
for mode in modes:
    mode_id = {"c2f": V.CANDIDATES_COARSE_TO_FINE, "dso": V.CANDIDATES_DSO, "dense": V.CANDIDATES_DENSE}[mode]
    blocky = (1 << 63) if mode == "dso" else 0
    for d in range(draws):
        rng = np.random.default_rng(11 + d)
        speed = 0.5 + 1.0 * rng.random(n_seq)
        sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
        frames = [V.synth_render_frames([blocky | (4242 + 1000 * d + s) for s in range(n_seq)], [k] * n_seq,
                                        [base * sign[s] * speed[s] * k for s in range(n_seq)], rows, cols, intr) for k in range(n_frames)]
        gh = np.stack([g.cpu().numpy() for g, _ in frames]); dh = np.stack([x.cpu().numpy().view(np.uint16) for _, x in frames])
        ocfg = O.make_config(L, intr, candidates_mode=mode_id)
        nt = min(os.cpu_count() or 1, n_seq)
        ref = O.track_sequences(ocfg, gh, dh, n_threads=nt)["poses"]
        ref64 = O.track_sequences(ocfg, gh, dh, n_threads=nt, variant="acc64")["poses"]
        mk = lambda a: V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode_id, arithmetic=a)
        res = {"fused (warp chain on small levels)": run(mk(V.ARITH_FUSED), frames),
               "fused, VORS_FUSED_SMALL=exact": run(mk(V.ARITH_FUSED), frames, {"VORS_FUSED_SMALL": "exact"}),
               **{f"fused, VORS_FUSED_EXACT_POINTS={t}": run(mk(V.ARITH_FUSED), frames, {"VORS_FUSED_EXACT_POINTS": t})
                  for t in os.environ.get("THRESHOLDS", "").split(",") if t},
               "exact": run(mk(V.ARITH_EXACT), frames), "oracle f64 sums": ref64}
        line = []
        for name, traj in res.items():
            err = np.abs(traj - ref).max(axis=(1, 2))
            line.append(f"{name}: {int((err > 1e-4).sum())} (max {err.max():.2e})")
            t = tot.setdefault((mode, name), [0, 0.0]); t[0] += int((err > 1e-4).sum()); t[1] = max(t[1], float(err.max()))
        print(f"{mode} draw {d}: sequences beyond 1e-4 of {n_seq} | " + " | ".join(line), flush=True)
        del frames
for (mode, name), (n, mx) in tot.items():
    print(f"TOTAL {mode} over {draws * n_seq} sequences: {name}: {n} beyond 1e-4, max {mx:.2e}")
