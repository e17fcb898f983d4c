"""Development aid (run through gpurun): attribute the out-of-tolerance tail of the FUSED arithmetic.

The oracle (and its f64-accumulation build) runs ONCE on the bench's own pairs; then every library build named in --libs
(tools/build_variant.sh TAG ... -> libvors_hip_eTAG.so; "base" = libvors_hip.so; "base:exact" = base in the EXACT arithmetic)
tracks the same pairs in its own process and is compared with the cached oracle output: pairs beyond 1e-4, quantiles, the level at
which each outlier's LM path first forks (first level, coarsest first, whose iteration count differs from the oracle's), LM-stage time.

    python tools/parity_ablate.py --c2f 4096 --dense 1024 --libs base:exact,base,step,warp
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np

ROWS, COLS, L = (int(x) for x in os.environ.get("ABL_SHAPE", "480,640,6").split(","))  # rows, cols, levels
MODES = {"c2f": 0, "dense": 1, "dso": 2}


def seed_of(mode):
    return 0x5EED0000 | ((1 << 63) if mode == "dso" else 0)


def worker(args):
    import torch
    import vors_amd as V
    from oracle import oracle as O
    intr = O.scaled_intrinsics(ROWS, COLS)
    arith = {"exact": V.ARITH_EXACT, "fused": V.ARITH_FUSED, "reference": V.ARITH_REFERENCE}[args.arith]
    out = {"lib": args.tag}
    for mode, n in (("c2f", args.c2f), ("dense", args.dense), ("dso", args.dso)):
        if n <= 0:
            continue
        ref = np.load(f"/tmp/abl_oracle_{mode}_{n}_{ROWS}x{COLS}_L{L}.npz")
        kg, kd, cg, _, gt = V.synth_render_pairs(seed_of(mode), n, ROWS, COLS, intr)
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=MODES[mode], arithmetic=arith)
        b = V.Batch(cfg, n, ROWS, COLS)
        poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
        b.enable_kernel_timing(8)
        for _ in range(4):
            b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        st = V.decode_stats(stats)
        err = np.abs(poses.cpu().numpy() - ref["poses"]).max(axis=1)
        it_g, it_o = st["nb_iter"][:, :L], ref["nb_iter"]
        differs = it_g != it_o
        # first level, coarsest first, whose iteration count differs (-1 = none)
        fork = np.where(differs.any(axis=1), L - 1 - np.argmax(differs[:, ::-1], axis=1), -1)
        bad = np.nonzero(err > 1e-4)[0]
        e64 = np.abs(ref["poses64"] - ref["poses"]).max(axis=1)
        q = np.quantile(err, [0.5, 0.99, 1.0])
        out[mode] = {
            "n": n, "n_beyond_1e-4": int(len(bad)), "n_beyond_1e-5": int((err > 1e-5).sum()), "median": float(q[0]), "p99": float(q[1]), "max": float(q[2]),
            "acc64_beyond_1e-4": int((e64 > 1e-4).sum()), "also_acc64_outliers": int(((e64 > 1e-4) & (err > 1e-4)).sum()),
            "status_equal": bool((status.cpu().numpy() == ref["status"]).all()),
            "same_iters": float((~differs.any(axis=1)).mean()),
            "fork_level_hist_outliers": {int(k): int(v) for k, v in zip(*np.unique(fork[bad], return_counts=True))},
            "fork_level_hist_all": {int(k): int(v) for k, v in zip(*np.unique(fork, return_counts=True))},
            "outliers": [{"pair": int(i), "err": float(err[i]), "fork": int(fork[i]), "gpu": it_g[i].tolist(), "oracle": it_o[i].tolist()} for i in bad[:12]],
            "ms_per_step": round(ms, 3), "lm_ms": round(float(b.kernel_times("lm")[-5:].mean()), 3),
        }
        del b
    print("ABL " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c2f", type=int, default=4096)
    ap.add_argument("--dense", type=int, default=1024)
    ap.add_argument("--dso", type=int, default=0)
    ap.add_argument("--libs", default="base:exact,base")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--tag", default="base")
    ap.add_argument("--arith", default="fused")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import torch
    import vors_amd as V
    from oracle import oracle as O
    intr = O.scaled_intrinsics(ROWS, COLS)
    for mode, n in (("c2f", args.c2f), ("dense", args.dense), ("dso", args.dso)):
        path = f"/tmp/abl_oracle_{mode}_{n}_{ROWS}x{COLS}_L{L}.npz"
        if n <= 0 or os.path.exists(path):
            continue
        kg, kd, cg, _, _ = V.synth_render_pairs(seed_of(mode), n, ROWS, COLS, intr)
        kgn, kdn, cgn = kg.cpu().numpy(), kd.cpu().numpy().view(np.uint16), cg.cpu().numpy()
        ocfg = O.make_config(L, intr, candidates_mode=MODES[mode])
        t0 = time.time()
        ref = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=os.cpu_count())
        r64 = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=os.cpu_count(), variant="acc64")
        np.savez(path, poses=ref["poses"], status=ref["status"], nb_iter=ref["nb_iter"], poses64=r64["poses"], nb_iter64=r64["nb_iter"])
        print(f"oracle {mode} x{n}: {time.time() - t0:.1f} s on {os.cpu_count()} threads", flush=True)
        del kg, kd, cg
    torch.cuda.empty_cache()
    for spec in args.libs.split(","):
        parts = spec.split(":")  # tag[:arith[:ENV=value;ENV=value]]
        tag, arith = parts[0], (parts[1] if len(parts) > 1 else "")
        lib = os.path.join(ROOT, "visual-odometry-rs_amd", "vors_amd", "libvors_hip.so" if tag == "base" else f"libvors_hip_e{tag}.so")
        env = dict(os.environ, VORS_HIP_LIB=lib)
        if len(parts) > 2:
            env.update(kv.split("=", 1) for kv in parts[2].split(";") if kv)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--tag", spec, "--arith", arith or "fused", "--c2f", str(args.c2f),
                            "--dense", str(args.dense), "--dso", str(args.dso)], env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("ABL ")]
        if not lines:
            print(f"[{spec}] FAILED: {r.stderr[-600:]}", flush=True)
            continue
        d = json.loads(lines[0][4:])
        for mode in ("c2f", "dense", "dso"):
            if mode in d:
                m = d[mode]
                print(f"[{spec}] {COLS}x{ROWS} L{L} {mode} x{m['n']}: >1e-4: {m['n_beyond_1e-4']} (acc64: {m['acc64_beyond_1e-4']}, shared {m['also_acc64_outliers']}) >1e-5: {m['n_beyond_1e-5']} "
                      f"median {m['median']:.2e} p99 {m['p99']:.2e} max {m['max']:.2e}; same iters {m['same_iters']:.1%}; fork levels of outliers {m['fork_level_hist_outliers']}; "
                      f"{m['ms_per_step']} ms/step, LM {m['lm_ms']} ms; status_equal {m['status_equal']}", flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "ablate.jsonl"), "a") as f:
            f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
