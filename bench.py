#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the MI355X direct-alignment hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic frame pairs already resident in HBM:
per pair  Config::init(keyframe) [mean pyramid, gradients, candidate selection, inverse-depth pyramid, Jacobians]
        + Tracker::track(current) [mean pyramid, coarse->fine Levenberg-Marquardt on the device, keyframe test],
then (N > 1) one RCCL all-gather of 8 f32 per pair (pose + status). Pairs are independent: each rank (one process per GPU)
owns `--pairs` pairs (weak scaling), no data-path collective except that gather.

    python bench.py [--gpus N --steps K --warmup W] [--pairs P] [--candidates dense|c2f|dso] [--arith fused|exact]
                    [--rows R --cols C --levels L --huber D]     (config 5: --rows 960 --cols 1280 --levels 7 --huber 10 --pairs 512)

N > 1 is launched by the driver as: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.  The CPU oracle (oracle/) is used here only for the `cpu_baseline` / `parity` leg (it is the
checker and the baseline, never the thing measured); everything timed on the GPU goes through libvors_hip.so alone.

Byte model (refined as VERDICT r01 asks): per point-evaluation 9 B dense / 13 B sparse (coordinates 4 when stored,
inverse depth 4, template 1, four taps 4) for every energy evaluation, + 4 B (gx, gy) for the evaluations whose g and H the reference
forms (the initial one of a level and every accepted candidate: vors_pair_stats.nb_grad_evals). The flat SURVEY figure (13 / 17 B for
every evaluation) is reported next to it.
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

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (≈6.3 TB/s achievable)
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; VALU issue peak = one wave64 instruction per 2 cycles per SIMD (157 TFLOP/s f32)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--pairs", type=int, default=4096,
                   help="frame pairs per GPU per step (weak scaling; BASELINE config 4's batch size). The headline uses 4096; the "
                        "SURVEY §8d count (256 = one pair per CU) is measured as well and reported under `batch_256`")
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
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other candidate mode, 256-pair batch)")
    p.add_argument("--graph", action="store_true", help="replay each step from a captured HIP graph (kernel timing off)")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="process-group backend for N > 1: nccl (= RCCL over xGMI, the measured configuration) or gloo (control-plane test of "
                        "the N > 1 code path on a box with fewer GPUs than ranks: ranks then share devices round-robin)")
    return p.parse_args()


def byte_model(stats, L, rows, cols, dense):
    """-> (B_io, B_lm refined, B_lm flat, mean evaluations per pair, mean point-evaluations per pair)."""
    b_energy, b_flat = (9, 13) if dense else (13, 17)
    nb_iter = stats["nb_iter"][:, :L].astype(np.int64)
    n_pts = stats["n_points"][:, :L].astype(np.int64)
    grad = stats["nb_grad_evals"][:, :L].astype(np.int64)
    evals = np.where(nb_iter > 0, nb_iter + 1, 0)          # E_l = energy evaluations executed = iterations + 1 when the level ran
    b_lm = int((evals * n_pts).sum()) * b_energy + int((grad * n_pts).sum()) * 4
    b_flat_total = int((evals * n_pts).sum()) * b_flat
    b_io = (4 * rows * cols + 32) * len(stats)
    return b_io, b_lm, b_flat_total, float(evals.sum(1).mean()), float((evals * n_pts).sum(1).mean())


def lm_counters(args):
    """Counters of the LM stage of one step of THIS workload from the committed rocprofv3 PMC passes (profiles/lm_counters.json,
    written by tools/make_lm_counters.py from `tools/profile.sh` + `tools/pmc_sq.sh` output), or {} when none was taken."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "lm_counters.json")))
    except Exception:
        return {}
    key = f"{args.candidates}_{args.arith}_{args.cols}x{args.rows}_L{args.levels}_{args.pairs}pairs" + (f"_huber{args.huber:g}" if args.huber > 0 else "")
    return table.get(key, {})


class Workload:
    def __init__(self, V, args, mode, device, seed0, pairs=None):
        self.V, self.args, self.mode = V, args, mode
        n, rows, cols, L = pairs or args.pairs, args.rows, args.cols, args.levels
        self.n = n
        self.intr = V.scaled_intrinsics(rows, cols)
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


def timed_run(work, steps, warmup, world, packed, gathered):
    import torch.distributed as dist
    from vors_amd.distributed import gather_packed

    def one():
        work.step()
        if world > 1:  # the single collective of a step: 8 f32 per pair (pose 7 + status), RCCL all-gather over xGMI
            packed[:, :7] = work.poses
            packed[:, 7] = work.status
            gather_packed(packed, out=gathered)

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
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline_and_parity(args, work, value):
    """the oracle (C++ restatement of the reference, `kind: port`) on a bounded sample of the same pairs — a
    -march=native build made on THIS host (FMA contraction stays off: the arithmetic is the oracle's), one pinned thread like the
    single-threaded reference, 1 warm-up + median of 5 runs; plus an all-cores figure. The same sample is the parity check of the run:
    GPU poses vs oracle, branch-flip rate, and the f64-accumulation probe of the oracle (how much the f32 summation order decides)."""
    from oracle import oracle as O
    dense = args.candidates == "dense"
    n_cpu = args.cpu_pairs
    if n_cpu < 0:
        n_cpu = 48 if dense else 1024      # ≈ 2 s (dense) / 2.5 s (sparse) per single-thread run at 640x480; x 6 runs + the all-cores leg
        scale = (args.rows * args.cols) / (480.0 * 640.0)
        n_cpu = max(4, int(n_cpu / scale))
    n_cpu = min(n_cpu, work.n)
    kg = work.kg[:n_cpu].cpu().numpy()
    kd = work.kd[:n_cpu].cpu().numpy().view(np.uint16)
    cg = work.cg[:n_cpu].cpu().numpy()
    ocfg = O.make_config(args.levels, work.intr, candidates_mode=work.mode_id, huber_delta=args.huber)
    ncores = os.cpu_count() or 1
    pinned = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {allowed[len(allowed) // 2]})
        pinned = allowed[len(allowed) // 2]
    except (AttributeError, OSError):
        allowed = None
    try:
        O.track_pairs(ocfg, kg[:max(1, n_cpu // 8)], kd[:max(1, n_cpu // 8)], cg[:max(1, n_cpu // 8)], n_threads=1, variant="native")  # warm-up
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            O.track_pairs(ocfg, kg, kd, cg, n_threads=1, variant="native")
            runs.append(time.perf_counter() - t0)
    finally:
        if allowed is not None:
            os.sched_setaffinity(0, set(allowed))
    t_cpu = float(np.median(runs))
    all_runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n_cpu), variant="native")
        all_runs.append(time.perf_counter() - t0)
    t_all = float(np.median(all_runs))
    # parity of this run (the oracle proper: baseline x86-64 build, no FMA)
    ref = O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n_cpu))
    ref64 = O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n_cpu), variant="acc64")
    gpu_poses = work.poses[:n_cpu].cpu().numpy()
    st = work.V.decode_stats(work.stats)[:n_cpu]
    L = args.levels
    ok = ref["status"] == 0
    cpu = {
        "value": round(n_cpu / t_cpu, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
        "sample": f"first {n_cpu} pairs of the same batch, same candidates mode; oracle/ C++ restatement of the reference built on this host with "
                  f"-O3 -march=native (no FMA contraction, no fast-math), ONE thread pinned to core {pinned} like the single-threaded reference; "
                  f"1 warm-up + median of 5 runs ({min(runs):.2f}-{max(runs):.2f} s)",
        "all_cores": {"value": round(n_cpu / t_all, 3), "cores": min(ncores, n_cpu), "note": "one thread per contiguous block of pairs, median of 3"},
        "gpu_over_cpu_1core": round(value / (n_cpu / t_cpu), 1),
    }
    parity = {
        "sample_pairs": n_cpu,
        "max_pose_diff_gpu_vs_oracle": float(np.abs(gpu_poses - ref["poses"])[ok].max(initial=0.0)),
        "status_equal": bool((work.status[:n_cpu].cpu().numpy() == ref["status"]).all()),
        "branch_flip_rate_gpu_vs_oracle": float((st["nb_iter"][:, :L] != ref["nb_iter"]).any(axis=1).mean()),
        "max_pose_diff_oracle_f32_vs_f64_accumulation": float(np.abs(ref64["poses"] - ref["poses"])[ok].max(initial=0.0)),
        "branch_flip_rate_oracle_f32_vs_f64_accumulation": float((ref64["nb_iter"] != ref["nb_iter"]).any(axis=1).mean()),
        "tolerance": 1e-4,
    }
    return cpu, parity


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
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (RCCL needs one GPU per rank)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    import vors_amd as V

    seed0 = 0x5EED0000 + rank * args.pairs
    main_w = Workload(V, args, args.candidates, device, seed0)
    packed = torch.zeros((args.pairs, 8), dtype=torch.float32, device=device) if world > 1 else None
    gathered = torch.zeros((world * args.pairs, 8), dtype=torch.float32, device=device) if world > 1 else None

    ring = min(max(args.steps, 1), 4096)
    if args.graph:
        main_w.capture()
        dt = timed_run(main_w, args.steps, args.warmup, world, packed, gathered)
        main_w.graph = None
        main_w.batch.enable_kernel_timing(ring)   # kernel durations from a few eager steps after the timed region
        for _ in range(min(args.steps, 5)):
            main_w.step()
        torch.cuda.synchronize()
    else:
        main_w.batch.enable_kernel_timing(ring)
        dt = timed_run(main_w, args.steps, args.warmup, world, packed, gathered)
    lm_ms = main_w.batch.kernel_times("lm")[-args.steps:]
    kf_ms = main_w.batch.kernel_times("keyframe")[-args.steps:]
    pyr_ms = main_w.batch.kernel_times("pyramid_keyframe")[-args.steps:] + main_w.batch.kernel_times("pyramid_current")[-args.steps:]
    total_pairs = world * args.pairs * args.steps
    value = total_pairs / dt

    stats = V.decode_stats(main_w.stats)
    dense = args.candidates == "dense"
    b_io, b_lm, b_lm_flat, evals_per_pair, ptevals_per_pair = byte_model(stats, args.levels, args.rows, args.cols, dense)
    lm_avg_s = float(lm_ms.mean()) * 1e-3
    lm_bytes = b_lm + 32 * args.pairs  # algorithmic bytes of the LM stage of ONE step (this rank's batch)
    achieved = lm_bytes / lm_avg_s / 1e9
    job_gbps = (b_io + b_lm) * world * args.steps / dt / 1e9
    gt_err = np.abs(stats["lm_model"] - main_w.gt.cpu().numpy()).max(axis=1)
    cnt = lm_counters(args)
    base_shape = (args.rows, args.cols, args.levels) == (480, 640, 6)

    roofline = {
        "bound": "hbm",
        # dense mode: the LM stage is a short sequence of launches (coarse levels per pair, then one launch per energy evaluation round on
        # the finest levels + a per-pair step launch, then the per-pair epilogue); it is timed as a whole with HIP events on its stream,
        # and its algorithmic bytes are those of all its evaluations
        "kernel": ("LM stage: lm_track_kernel (coarse levels) + lm_split_eval_kernel / lm_split_step_kernel per evaluation round "
                   "+ lm_track_kernel (epilogue)") if dense else "lm_track_kernel",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 5),
        # HBM bytes of the same stage from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; separate rocprofv3 --pmc passes of this
        # exact command, committed under profiles/): NOT measured in this run — null when no profile of this workload exists
        "traffic": cnt.get("traffic_bytes"),
        "traffic_source": cnt.get("profile", None) and f"from_profile: {cnt['profile']}",
        "algorithmic_bytes_per_launch": lm_bytes,
        "algorithmic_bytes_flat_model": b_lm_flat + 32 * args.pairs,
        "frac_flat_model": round((b_lm_flat + 32 * args.pairs) / lm_avg_s / 1e9 / HBM_PEAK_GBPS, 5),
        "kernel_ms_avg": round(lm_avg_s * 1e3, 5),
        "whole_job_GBps": round(job_gbps, 2),
        "whole_job_frac": round(job_gbps / (HBM_PEAK_GBPS * world), 5),
        "io_only_GBps": round(b_io * world * args.steps / dt / 1e9, 2),
        # what actually binds the stage: wave64 VALU instructions issued (SQ_INSTS_VALU of the same profile) x 2 cycles over the
        # SIMD-cycles of the measured stage time (the instruction mix is ~40 % half-rate ops)
        "binding_resource": "VALU issue",
        "valu_issue_frac": (round(cnt["sq_insts_valu"] * 2.0 / (N_SIMD * lm_avg_s * CLOCK_HZ), 4) if "sq_insts_valu" in cnt else None),
        "valu_busy_frac": (round(cnt["sq_active_inst_valu"] * 4.0 / (N_SIMD * lm_avg_s * CLOCK_HZ), 4) if "sq_active_inst_valu" in cnt else None),
    }
    out = {
        "metric": "frame-pairs/sec (640x480, 6 pyramid levels)" if base_shape else f"frame-pairs/sec ({args.cols}x{args.rows}, {args.levels} pyramid levels)",
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
                         if dense and base_shape and args.huber == 0 else
                         ("BASELINE configs[4] shape: synthetic 1280x960 RGB-D, 7-level pyramid, Huber weighting, dense candidates"
                          if dense and (args.rows, args.cols, args.levels) == (960, 1280, 7) and args.huber > 0 else
                          f"synthetic {args.cols}x{args.rows} gray u8 + depth u16, {args.levels}-level pyramid, {args.candidates} candidates")),
            "pairs_per_gpu": args.pairs,
            "candidates": {"dense": "dense (all-true level-0 mask, extension)", "c2f": "coarse_to_fine (reference selection)",
                           "dso": "DSO-style selection (dso.rs, examples/candidates_dso.rs parameters)"}[args.candidates],
            "huber_delta": args.huber,
            "arithmetic": args.arith,
            "parallelism": (f"pairs sharded over {world} ranks, one all-gather of pose+status per step "
                            f"({'RCCL over xGMI, one GPU per rank' if args.backend == 'nccl' else 'gloo TEST backend, ranks share ' + str(n_dev) + ' GPU(s)'})"
                            if world > 1 else "1 GPU"),
            "launch": "hipGraph replay" if args.graph else "eager",
        },
        "roofline": roofline,
        "stages_ms": {"pyramids": round(float(pyr_ms.mean()), 5), "keyframe": round(float(kf_ms.mean()), 5),
                      "lm": round(float(lm_ms.mean()), 5)},
        "lm_evals_per_pair": round(evals_per_pair, 2),
        "lm_grad_evals_per_pair": round(float(stats["nb_grad_evals"][:, :args.levels].sum(1).mean()), 2),
        "point_evals_per_pair": round(ptevals_per_pair, 1),
        "failed_pairs": int((main_w.status != 0).sum().item()),
        "pose_err_vs_ground_truth": {"median": float(np.median(gt_err)), "max": float(gt_err.max())},
    }

    if rank == 0 and world == 1:
        if not args.no_secondary:
            # ---- SURVEY §8d batch size: 256 pairs (one per CU) of the same workload
            if args.pairs != 256:
                ws = Workload(V, args, args.candidates, device, seed0, pairs=256)
                ws.batch.enable_kernel_timing(ring)
                dts = timed_run(ws, args.steps, args.warmup, 1, None, None)
                out["batch_256"] = {"value": round(256 * args.steps / dts, 2), "unit": "frame-pairs/s", "pairs_per_gpu": 256,
                                    "ms_per_step": round(dts / args.steps * 1e3, 4),
                                    "lm_kernel_ms": round(float(ws.batch.kernel_times("lm")[-args.steps:].mean()), 5)}
                del ws
            # ---- the other candidate mode (the reference's own selection when the headline is dense)
            if args.candidates != "dso":
                other = "c2f" if dense else "dense"
                w2 = Workload(V, args, other, device, seed0)
                w2.batch.enable_kernel_timing(ring)
                dt2 = timed_run(w2, args.steps, args.warmup, 1, None, None)
                st2 = V.decode_stats(w2.stats)
                io2, lmb2, _, ev2, _ = byte_model(st2, args.levels, args.rows, args.cols, other == "dense")
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
            # ---- the same workload with steps alternating between TWO handles on two HIP streams (each step is still one full pass over its
            # own batch of `pairs` pairs; the GPU overlaps the latency-bound tail of one step — straggler rounds, tree descent — with the
            # VALU-bound body of the next). `value` above stays the single-stream figure: its stage times and roofline are self-consistent.
            w3 = Workload(V, args, args.candidates, device, seed0 + args.pairs)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            both = [main_w, w3]

            def alt(i):
                with torch.cuda.stream(streams[i & 1]):
                    both[i & 1].step()
            for i in range(2 * max(args.warmup, 1)):
                alt(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                alt(i)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            out["pipelined_two_streams"] = {"value": round(args.pairs * args.steps / dt3, 2), "unit": "frame-pairs/s",
                                            "ms_per_step": round(dt3 / args.steps * 1e3, 4),
                                            "note": "steps alternate between two batch handles on two streams; not the headline"}
            del w3
        if args.cpu_pairs != 0:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(args, main_w, value)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
