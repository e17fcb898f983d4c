#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the MI355X direct-alignment hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic frame pairs already resident in HBM:
per pair  Config::init(keyframe) [mean pyramid, gradients, candidate selection, inverse-depth pyramid, Jacobians]
        + Tracker::track(current) [mean pyramid, coarse->fine Levenberg-Marquardt on the device, keyframe test],
then (N > 1) one RCCL all-gather of the poses. Pairs are independent: each rank (one process per GPU) owns
`--pairs` pairs (weak scaling), no data-path collective except that gather.

    python bench.py [--gpus N --steps K --warmup W] [--pairs P] [--candidates dense|c2f] [--rows R --cols C --levels L]

N > 1 is launched by the driver as: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.  The CPU oracle (oracle/) is used here only for the `cpu_baseline` leg and a sanity
check of the first pairs; it is never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--pairs", type=int, default=4096,
                   help="frame pairs per GPU per step (weak scaling; BASELINE config 4's batch size). Pairs need different numbers "
                        "of LM evaluations, so throughput grows with the batch until the workgroups balance (dense: 80 k pairs/s "
                        "at 256, 105 k at 1024, 140 k at 4096); SURVEY.md §8d's 256 = one pair per CU. See DESIGN.md §3/§7")
    p.add_argument("--candidates", choices=["dense", "c2f", "dso"], default="dense",
                   help="dense = BASELINE configs[1] (extension); c2f = the reference's coarse-to-fine selection; "
                        "dso = DSO-style selection (config 3; piecewise-constant synthetic texture)")
    p.add_argument("--rows", type=int, default=480)
    p.add_argument("--cols", type=int, default=640)
    p.add_argument("--levels", type=int, default=6)
    p.add_argument("--huber", type=float, default=0.0)
    p.add_argument("--arith", choices=["fused", "exact"], default="fused",
                   help="per-point arithmetic (include/vors_hip.h VORS_ARITH_*): fused = equivalent shorter f32 forms (poses within the 1e-4 "
                        "parity bar, gated by tests/test_gpu_fused.py); exact = the reference's evaluation order (parity anchor)")
    p.add_argument("--cpu-pairs", type=int, default=-1, help="pairs timed on the CPU oracle (-1 = auto, 0 = skip)")
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary (other candidate mode) measurement")
    p.add_argument("--graph", action="store_true", help="replay each step from a captured HIP graph (kernel timing off)")
    return p.parse_args()


def byte_model(stats, L, rows, cols, dense):
    """Algorithmic bytes: B_io = 4*S0 + 32 per pair; B_lm = sum_l E_l * N_l * b_pt with
    b_pt = 13 B dense / 17 B sparse, E_l = energy evaluations executed at level l = nb_iter_l + 1 when the level ran."""
    b_pt = 13 if dense else 17
    nb_iter = stats["nb_iter"][:, :L].astype(np.int64)
    n_pts = stats["n_points"][:, :L].astype(np.int64)
    evals = np.where(nb_iter > 0, nb_iter + 1, 0)
    b_lm = int((evals * n_pts).sum()) * b_pt
    b_io = (4 * rows * cols + 32) * len(stats)
    return b_io, b_lm, float(evals.sum(1).mean()), float((evals * n_pts).sum(1).mean())


def lm_traffic(args):
    """HBM bytes of the LM stage of one step (all its kernels) measured with rocprofv3 PMC passes (profiles/lm_traffic.json), or
    None when no profile of this exact workload has been committed."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "lm_traffic.json")))
    except Exception:
        return None
    key = f"{args.candidates}_{args.cols}x{args.rows}_L{args.levels}_{args.pairs}pairs"
    e = table.get(key)
    return int(e["traffic_bytes"]) if e and args.huber == 0.0 else None


class Workload:
    def __init__(self, V, args, mode, device, seed0):
        self.V, self.args, self.mode = V, args, mode
        n, rows, cols, L = args.pairs, args.rows, args.cols, args.levels
        from oracle import oracle as O  # only for the intrinsics helper constants (no compute)
        self.intr = O.scaled_intrinsics(rows, cols)
        self.mode_id = {"dense": V.CANDIDATES_DENSE, "c2f": V.CANDIDATES_COARSE_TO_FINE, "dso": V.CANDIDATES_DSO}[mode]
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(self.intr[:2], self.intr[2:4], self.intr[4]),
                       candidates_mode=self.mode_id, huber_delta=args.huber,
                       arithmetic=V.ARITH_FUSED if args.arith == "fused" else V.ARITH_EXACT)
        if mode == "dso":
            seed0 |= 1 << 63  # piecewise-constant texture: the DSO thresholds reject the smooth texture entirely
        self.cfg = cfg
        self.batch = V.Batch(cfg, n, rows, cols)
        self.kg, self.kd, self.cg, _, self.gt = V.synth_render_pairs(seed0, n, rows, cols, self.intr, device=device)
        self.poses = torch.zeros((n, 7), dtype=torch.float32, device=device)
        self.status = torch.zeros(n, dtype=torch.int32, device=device)
        self.stats = V.stats_tensor(n, device=device)
        self.graph = None

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.batch.track_pairs(self.kg, self.kd, self.cg, self.poses, self.status, self.stats)

    def capture(self):
        """Capture one step into a HIP graph (the launches are purely stream-ordered)."""
        self.step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                self.batch.track_pairs(self.kg, self.kd, self.cg, self.poses, self.status, self.stats)
        self.graph = g


def timed_run(work, steps, warmup, world, gathered):
    import torch.distributed as dist
    from vors_amd.distributed import gather_poses

    def one():
        work.step()
        if world > 1:
            gather_poses(work.poses, out=gathered)  # the single RCCL all-gather of poses

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    import vors_amd as V

    seed0 = 0x5EED0000 + rank * args.pairs
    main_w = Workload(V, args, args.candidates, device, seed0)
    gathered = torch.zeros((world * args.pairs, 7), dtype=torch.float32, device=device) if world > 1 else None

    ring = min(max(args.steps, 1), 4096)
    if args.graph:
        main_w.capture()
        dt = timed_run(main_w, args.steps, args.warmup, world, gathered)
        main_w.graph = None
        main_w.batch.enable_kernel_timing(ring)   # kernel durations from a few eager steps after the timed region
        for _ in range(min(args.steps, 5)):
            main_w.step()
        torch.cuda.synchronize()
    else:
        main_w.batch.enable_kernel_timing(ring)
        dt = timed_run(main_w, args.steps, args.warmup, world, gathered)
    lm_ms = main_w.batch.kernel_times("lm")[-args.steps:]
    kf_ms = main_w.batch.kernel_times("keyframe")[-args.steps:]
    pyr_ms = main_w.batch.kernel_times("pyramid_keyframe")[-args.steps:] + main_w.batch.kernel_times("pyramid_current")[-args.steps:]
    total_pairs = world * args.pairs * args.steps
    value = total_pairs / dt

    stats = V.decode_stats(main_w.stats)
    dense = args.candidates == "dense"
    b_io, b_lm, evals_per_pair, ptevals_per_pair = byte_model(stats, args.levels, args.rows, args.cols, dense)
    lm_avg_s = float(lm_ms.mean()) * 1e-3
    lm_bytes = b_lm + 32 * args.pairs  # algorithmic bytes of the LM stage of ONE step (this rank's batch)
    achieved = lm_bytes / lm_avg_s / 1e9
    job_gbps = (b_io + b_lm) * world * args.steps / dt / 1e9
    gt_err = np.abs(stats["lm_model"] - main_w.gt.cpu().numpy()).max(axis=1)

    out = {
        "metric": "frame-pairs/sec (640x480, 6 pyramid levels)" if (args.rows, args.cols, args.levels) == (480, 640, 6)
        else f"frame-pairs/sec ({args.cols}x{args.rows}, {args.levels} pyramid levels)",
        "value": round(value, 2),
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: synthetic 640x480 RGB-D (gray u8 + depth u16), 6-level pyramid, dense candidates"
                         if dense and (args.rows, args.cols, args.levels) == (480, 640, 6) else
                         f"synthetic {args.cols}x{args.rows} gray u8 + depth u16, {args.levels}-level pyramid, {args.candidates} candidates"),
            "pairs_per_gpu": args.pairs,
            "candidates": {"dense": "dense (all-true level-0 mask, extension)", "c2f": "coarse_to_fine (reference selection)",
                           "dso": "DSO-style selection (dso.rs, examples/candidates_dso.rs parameters)"}[args.candidates],
            "huber_delta": args.huber,
            "arithmetic": args.arith,
            "parallelism": f"pairs sharded over {world} GPU(s), one RCCL all-gather of poses per step" if world > 1 else "1 GPU",
            "launch": "hipGraph replay" if args.graph else "eager",
        },
        "roofline": {
            "bound": "hbm",
            # dense mode: the LM stage is a short sequence of launches (coarse levels per pair, then one launch per energy
            # evaluation round on the two finest levels + a per-pair step launch, then the per-pair epilogue); it is timed as a
            # whole with HIP events on its stream, and its algorithmic bytes are those of all its evaluations
            "kernel": ("LM stage: lm_track_kernel (coarse levels) + lm_split_eval_kernel / lm_split_step_kernel per evaluation round "
                       "+ lm_track_kernel (epilogue)") if dense else "lm_track_kernel",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5),
            "traffic": lm_traffic(args),  # PMC FETCH/WRITE bytes per launch from the committed rocprofv3 passes (profiles/)
            "algorithmic_bytes_per_launch": lm_bytes,
            "kernel_ms_avg": round(lm_avg_s * 1e3, 5),
            "whole_job_GBps": round(job_gbps, 2),
            "whole_job_frac": round(job_gbps / (HBM_PEAK_GBPS * world), 5),
            "io_only_GBps": round(b_io * world * args.steps / dt / 1e9, 2),
        },
        "stages_ms": {"pyramids": round(float(pyr_ms.mean()), 5), "keyframe": round(float(kf_ms.mean()), 5),
                      "lm": round(float(lm_ms.mean()), 5)},
        "lm_evals_per_pair": round(evals_per_pair, 2),
        "point_evals_per_pair": round(ptevals_per_pair, 1),
        "failed_pairs": int((main_w.status != 0).sum().item()),
        "pose_err_vs_ground_truth": {"median": float(np.median(gt_err)), "max": float(gt_err.max())},
    }

    if rank == 0 and world == 1:
        # ---- secondary measurement: the other candidate mode (reference selection when the headline is dense)
        if not args.no_secondary and args.candidates != "dso":
            other = "c2f" if dense else "dense"
            w2 = Workload(V, args, other, device, seed0)
            w2.batch.enable_kernel_timing(ring)
            dt2 = timed_run(w2, args.steps, args.warmup, 1, None)
            st2 = V.decode_stats(w2.stats)
            io2, lmb2, ev2, _ = byte_model(st2, args.levels, args.rows, args.cols, other == "dense")
            out["secondary"] = {
                "candidates": "coarse_to_fine (reference selection)" if other == "c2f" else "dense",
                "value": round(args.pairs * args.steps / dt2, 2), "unit": "frame-pairs/s",
                "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                "lm_kernel_ms": round(float(w2.batch.kernel_times("lm")[-args.steps:].mean()), 5),
                "lm_evals_per_pair": round(ev2, 2),
                "whole_job_GBps": round((io2 + lmb2) * args.steps / dt2 / 1e9, 2),
                "io_only_GBps": round(io2 * args.steps / dt2 / 1e9, 2),
            }
            del w2
        # ---- CPU baseline: the oracle (C++ restatement of the reference, single thread like the reference)
        n_cpu = args.cpu_pairs
        if n_cpu < 0:
            n_cpu = 256 if dense else 2048   # about 10 s (dense) / 5 s (sparse) of single-thread work, plus the all-cores leg
        n_cpu = min(n_cpu, args.pairs)
        if n_cpu > 0:
            from oracle import oracle as O
            kg = main_w.kg[:n_cpu].cpu().numpy()
            kd = main_w.kd[:n_cpu].cpu().numpy().view(np.uint16)
            cg = main_w.cg[:n_cpu].cpu().numpy()
            ocfg = O.make_config(args.levels, main_w.intr, candidates_mode=main_w.mode_id, huber_delta=args.huber)
            t0 = time.perf_counter()
            ref = O.track_pairs(ocfg, kg, kd, cg, n_threads=1)
            t_cpu = time.perf_counter() - t0
            ncores = os.cpu_count() or 1
            t0 = time.perf_counter()
            O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n_cpu))
            t_all = time.perf_counter() - t0
            gpu_poses = main_w.poses[:n_cpu].cpu().numpy()
            out["cpu_baseline"] = {
                "value": round(n_cpu / t_cpu, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                "sample": f"first {n_cpu} pairs of the same batch, same candidates mode, oracle/ C++ restatement "
                          f"(-O3, no FMA contraction), single thread like the reference",
                "all_cores": {"value": round(n_cpu / t_all, 3), "cores": min(ncores, n_cpu)},
                "gpu_over_cpu_1core": round(value / (n_cpu / t_cpu), 1),
                "max_pose_diff_gpu_vs_oracle": float(np.abs(gpu_poses - ref["poses"]).max()),
            }
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
