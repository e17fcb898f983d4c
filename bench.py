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

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (ROCm default: 4) in creation order. This process creates
# many (every dense handle has a side-lane stream, every vors_pipeline ring one per slot): with 4 queues two slots of a ring end up on ONE
# queue and run one after the other — measured: 512 REFERENCE pairs per step through a ring of 3, 0.61 ms per step with 4 queues, 0.46 with 8
# (dense: 8.1 vs 5.0 ms). Must be set before the runtime initialises; the single-stream figures (`value`) do not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak: --pairs per GPU whatever N (the driver's default contract). strong: BASELINE configs[3] as written — a FIXED batch of "
                        "--total-pairs pairs sharded over the N ranks (ceil(total / N) each, the last rank takes the remainder)")
    p.add_argument("--total-pairs", type=int, default=4096, help="--scaling strong: the fixed batch (BASELINE config 4: 4096)")
    p.add_argument("--candidates", choices=["dense", "c2f", "dso"], default="dense",
                   help="dense = BASELINE configs[1] (extension); c2f = the reference's coarse-to-fine selection; "
                        "dso = DSO-style selection (config 3; piecewise-constant synthetic texture)")
    p.add_argument("--rows", type=int, default=480)
    p.add_argument("--cols", type=int, default=640)
    p.add_argument("--levels", type=int, default=6)
    p.add_argument("--huber", type=float, default=0.0)
    p.add_argument("--arith", choices=["fused", "exact", "reference"], default="fused",
                   help="per-point arithmetic (include/vors_hip.h VORS_ARITH_*): fused = equivalent shorter f32 forms (poses within the 1e-4 "
                        "parity bar, gated by tests/test_gpu_fused.py); exact = the reference's evaluation order per point; reference = exact + the "
                        "reference's sequential summation order: bit-identical to the oracle (the deterministic parity anchor)")
    p.add_argument("--cpu-pairs", type=int, default=-1, help="pairs timed on the CPU oracle (-1 = auto, 0 = skip)")
    p.add_argument("--parity-pairs", type=int, default=-1,
                   help="pairs of each measured workload compared with the oracle after the timed region (-1 = auto: 1024 dense / 4096 sparse on a "
                        "many-core host, 0 = skip)")
    p.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic then falls back to the committed profile)")
    p.add_argument("--no-sequences", action="store_true", help="skip the 64-sequence tracker measurement")
    p.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other candidate mode, 256-pair batch)")
    p.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] block (1280x960, 7 levels, Huber 10, 512 pairs)")
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


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel + engine sources: what a committed counter profile must have been taken from to describe
    THIS build (tools/summarize_prof.py stamps the same hash into profiles/lm_counters.json)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "visual-odometry-rs_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def lm_counters(args):
    """Counters of the LM stage of one step of THIS workload from the committed rocprofv3 PMC passes (profiles/lm_counters.json,
    written by tools/make_lm_counters.py from `tools/profile.sh` output) — only when that profile was taken from the very sources this
    build comes from (`source_sha16` stamp); {} otherwise."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "lm_counters.json")))
    except Exception:
        return {}
    key = f"{args.candidates}_{args.arith}_{args.cols}x{args.rows}_L{args.levels}_{args.pairs}pairs" + (f"_huber{args.huber:g}" if args.huber > 0 else "")
    e = table.get(key, {})
    if not e or e.get("source_sha16") != kernel_source_hash():
        return {}
    e = dict(e)
    e["source"] = f"from_profile: {e.get('profile')} (same kernel sources: sha16 {e['source_sha16']})"
    return e


LM_STAGE_KERNELS = ("lm_track_kernel", "lm_split_eval_kernel", "lm_split_step_kernel", "lm_ref_track_kernel", "lm_ref_track_coop_kernel")
ONCE_PER_STEP_KERNELS = ("dense_idepth_level1", "keyframe_sparse_kernel", "dso_rounds_kernel")  # launched exactly once per bench step


def live_counters(args, candidates=None):
    """HBM traffic and VALU counters of the LM stage of one step, MEASURED IN THIS RUN: three short rocprofv3 passes of this very command
    (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_*`; each its own process with --kernel-trace only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), 2 steps each, after the timed region.
    traffic = FETCH_SIZE KiB x 2 (the guide's gfx950 correction for wide coalesced reads) + WRITE_SIZE KiB. {} if rocprofv3 is missing / fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {}
    child = [sys.executable, os.path.abspath(__file__), "--pairs", str(args.pairs), "--steps", "2", "--warmup", "1", "--candidates", candidates or args.candidates,
             "--arith", args.arith, "--rows", str(args.rows), "--cols", str(args.cols), "--levels", str(args.levels), "--huber", str(args.huber),
             "--no-secondary", "--cpu-pairs", "0", "--parity-pairs", "0", "--no-pmc", "--no-sequences"]
    out, t0 = {}, time.perf_counter()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]):
        d = tempfile.mkdtemp(prefix="vors_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "b", "--"] + child,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=300, capture_output=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                continue
            agg, steps = {}, {}
            for r in csv.DictReader(open(files[0])):
                name, c = r["Kernel_Name"], r["Counter_Name"]
                if any(k in name for k in LM_STAGE_KERNELS):
                    agg[c] = agg.get(c, 0.0) + float(r["Counter_Value"])
                if any(k in name for k in ONCE_PER_STEP_KERNELS):
                    steps[c] = steps.get(c, 0) + 1
            for c, v in agg.items():
                if steps.get(c):
                    out[c.lower()] = v / steps[c]
        except Exception:
            pass
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "fetch_size" in out and "write_size" in out:
        out["traffic_bytes"] = int(out["fetch_size"] * 1024 * 2 + out["write_size"] * 1024)
    if out:
        out["source"] = (f"measured in this run: rocprofv3 --pmc passes of this command (2 steps each, {time.perf_counter() - t0:.0f} s); "
                         "FETCH_SIZE KiB x 2 + WRITE_SIZE KiB per MI355X_MICROARCH.md")
    return out


def arith_id(V, name):
    return {"fused": V.ARITH_FUSED, "exact": V.ARITH_EXACT, "reference": V.ARITH_REFERENCE}[name]


class Workload:
    def __init__(self, V, args, mode, device, seed0, pairs=None, batch=True):
        self.V, self.args, self.mode = V, args, mode
        n, rows, cols, L = pairs or args.pairs, args.rows, args.cols, args.levels
        self.n = n
        self.intr = V.scaled_intrinsics(rows, cols)
        self.mode_id = {"dense": V.CANDIDATES_DENSE, "c2f": V.CANDIDATES_COARSE_TO_FINE, "dso": V.CANDIDATES_DSO}[mode]
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(self.intr[:2], self.intr[2:4], self.intr[4]),
                       candidates_mode=self.mode_id, huber_delta=args.huber,
                       arithmetic=arith_id(V, args.arith))
        if mode == "dso":
            seed0 |= 1 << 63  # piecewise-constant texture: the DSO thresholds reject the smooth texture entirely
        self.cfg = cfg
        self.batch = V.Batch(cfg, n, rows, cols) if batch else None   # (batch=False: inputs / outputs only — the ring owns the handles)
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


class Ring:
    """A continuous feed through vors_pipeline_* (the C ABI's throughput mode): `depth` batch handles on internal streams, consecutive steps on
    consecutive handles; `depth` sets of inputs / outputs so that steps in flight share no buffer. Each step is still ONE full pass of the hot
    path over its own batch; what changes is that step k + 1's pyramids and keyframe stage run under step k's LM stage and its straggler tail."""

    def __init__(self, V, args, mode, device, seed0, pairs, depth):
        self.depth, self.n = depth, pairs
        self.works = [Workload(V, args, mode, device, seed0 + k * pairs, pairs=pairs, batch=False) for k in range(depth)]
        self.pipe = V.Pipeline(self.works[0].cfg, pairs, args.rows, args.cols, depth=depth)
        self.tickets = {}

    def submit(self, i):
        w = self.works[i % self.depth]
        self.tickets[i] = self.pipe.submit(w.kg, w.kd, w.cg, w.poses, w.status, w.stats)

    def run(self, steps, warmup, after_step=None, barrier=None):
        """-> seconds for `steps` steps (after `warmup` untimed ones); after_step(i, work) — e.g. the all-gather of a rank's results — is
        called for step i once it has completed in stream order, `depth - 1` steps behind the submissions."""
        def feed(first, count):
            for i in range(first, first + count):
                self.submit(i)
                j = i - (self.depth - 1)
                if after_step is not None and j >= first:
                    self.pipe.wait(self.tickets.pop(j))
                    after_step(j, self.works[j % self.depth])
            for j in range(max(first, first + count - (self.depth - 1)), first + count):
                if after_step is not None:
                    self.pipe.wait(self.tickets.pop(j))
                    after_step(j, self.works[j % self.depth])
            self.pipe.drain()
            self.tickets.clear()
        feed(0, max(warmup, self.depth))
        torch.cuda.synchronize()
        if barrier is not None:   # (N > 1: every rank starts and stops the clock behind a barrier, like timed_run)
            barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        feed(1000, steps)
        torch.cuda.synchronize()
        if barrier is not None:
            barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0


def ring_block(V, args, mode, device, seed0, pairs, steps, single_ms=None, depth=3):
    """Throughput of `pairs`-pair steps fed through a ring of `depth` handles (Ring) -> block for the JSON line."""
    r = Ring(V, args, mode, device, seed0, pairs, depth)
    # (a ring fills and drains once per measurement — about one step's worth: enough steps that the figure is the rate of the feed, within
    # ~3 s of GPU time)
    if single_ms:
        steps = max(steps, min(100, int(3000.0 / max(single_ms, 1e-3))))
    dt = r.run(steps, 2)
    blk = {"ring": depth, "steps": steps, "value": round(pairs * steps / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt / steps * 1e3, 4)}
    if single_ms:
        blk["gain_over_single_stream"] = round(single_ms / (dt / steps * 1e3), 3)
    del r
    return blk


def host_pairs(work, n):
    return (work.kg[:n].cpu().numpy(), work.kd[:n].cpu().numpy().view(np.uint16), work.cg[:n].cpu().numpy())


def cpu_baseline(args, work, value, n_override=None):
    """the oracle (C++ restatement of the reference, `kind: port`) on a bounded sample of the same pairs — a
    -march=native build made on THIS host (FMA contraction stays off: the arithmetic is the oracle's), one pinned thread like the
    single-threaded reference, 1 warm-up + median of 5 runs; plus an all-cores figure."""
    from oracle import oracle as O
    dense = work.mode == "dense"
    n_cpu = args.cpu_pairs if n_override is None else n_override
    if n_cpu < 0:
        n_cpu = 48 if dense else 1024      # ≈ 2 s (dense) / 2.5 s (sparse) per single-thread run at 640x480; x 6 runs + the all-cores leg
        scale = (args.rows * args.cols) / (480.0 * 640.0)
        n_cpu = max(4, int(n_cpu / scale))
    n_cpu = min(n_cpu, work.n)
    kg, kd, cg = host_pairs(work, n_cpu)
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
    return {
        "value": round(n_cpu / t_cpu, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
        "sample": f"first {n_cpu} pairs of the same batch, same candidates mode; oracle/ C++ restatement of the reference built on this host with "
                  f"-O3 -march=native (no FMA contraction, no fast-math), ONE thread pinned to core {pinned} like the single-threaded reference; "
                  f"1 warm-up + median of 5 runs ({min(runs):.2f}-{max(runs):.2f} s)",
        "all_cores": {"value": round(n_cpu / t_all, 3), "cores": min(ncores, n_cpu), "note": "one thread per contiguous block of pairs, median of 3"},
        "gpu_over_cpu_1core": round(value / (n_cpu / t_cpu), 1),
    }


def parity_block(args, work, n_want):
    """FULL-BATCH parity of the timed arithmetic: the poses the timed steps produced for the first n pairs of the
    batch against the oracle proper (baseline x86-64 build, no FMA) run on all host cores, with the tail made visible — pairs beyond
    the 1e-4 tolerance, p99, max — and beside it the oracle against its OWN f64-accumulation build on the same pairs: the floor set by
    the order of the reference's f32 sums, which no implementation that sums in another order can beat."""
    from oracle import oracle as O
    ncores = os.cpu_count() or 1
    n = min(n_want, work.n)
    kg, kd, cg = host_pairs(work, n)
    ocfg = O.make_config(args.levels, work.intr, candidates_mode=work.mode_id, huber_delta=args.huber)
    t0 = time.perf_counter()
    ref = O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n))
    t_oracle = time.perf_counter() - t0
    ref64 = O.track_pairs(ocfg, kg, kd, cg, n_threads=min(ncores, n), variant="acc64")
    gpu_poses = work.poses[:n].cpu().numpy()
    st = work.V.decode_stats(work.stats)[:n]
    L = args.levels
    ok = ref["status"] == 0
    err = np.abs(gpu_poses - ref["poses"]).max(axis=1)
    err64 = np.abs(ref64["poses"] - ref["poses"]).max(axis=1)
    err[~ok] = 0.0     # a failed pair keeps its previous pose in both (status equality is reported separately)
    err64[~ok] = 0.0
    q = np.quantile(err, [0.5, 0.99])
    return {
        "candidates": work.mode, "arithmetic": args.arith, "sample_pairs": int(n), "tolerance": 1e-4,
        "n_beyond_tol": int((err > 1e-4).sum()),
        "n_beyond_tol_oracle_f32_vs_f64_accumulation": int((err64 > 1e-4).sum()),
        "median_pose_diff": float(q[0]), "p99_pose_diff": float(q[1]), "max_pose_diff_gpu_vs_oracle": float(err.max(initial=0.0)),
        "max_pose_diff_oracle_f32_vs_f64_accumulation": float(err64.max(initial=0.0)),
        "status_equal": bool((work.status[:n].cpu().numpy() == ref["status"]).all()),
        "branch_flip_rate_gpu_vs_oracle": float((st["nb_iter"][:, :L] != ref["nb_iter"]).any(axis=1).mean()),
        "branch_flip_rate_oracle_f32_vs_f64_accumulation": float((ref64["nb_iter"] != ref["nb_iter"]).any(axis=1).mean()),
        "oracle_seconds_all_cores": round(t_oracle, 2), "oracle_threads": min(ncores, n),
    }


def multi_gpu_self_check(args, work, packed, gathered, rank, world):
    """First-run safety of the N > 1 path (it has never executed on real multi-GPU hardware before the driver's SCALE run): after the
    timed region every rank checks that block r of the gathered [world * P, 8] table equals rank r's local results — its own block
    against its own poses / statuses, and (through one more all-gather of a per-rank checksum) every other block against its owner's —
    and that the collective library answers its version query. Raises on any mismatch: a wrong gather must fail loudly, not time well."""
    import torch.distributed as dist
    P = work.n
    local = torch.empty((P, 8), dtype=torch.float32, device=packed.device)
    local[:, :7] = work.poses
    local[:, 7] = work.status
    mine = gathered[rank * P:(rank + 1) * P]
    if not torch.equal(mine, local):
        raise SystemExit(f"rank {rank}: its block of the gathered table differs from its local results")
    # a checksum of every block as each rank sees it vs the owner's checksum of its local table
    # (the f32 words summed as integers: exact and independent of the order of the reduction)
    sums_seen = gathered.contiguous().view(torch.int32).view(world, P * 8).to(torch.int64).sum(dim=1)
    own = local.view(torch.int32).to(torch.int64).sum().reshape(1)
    owners = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
    if dist.get_backend() == "nccl":
        dist.all_gather(owners, own)
    else:
        cpu_owners = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(cpu_owners, own.cpu())
        owners = [o.to(packed.device) for o in cpu_owners]
    owners = torch.cat(owners)
    if not torch.equal(sums_seen, owners):
        raise SystemExit(f"rank {rank}: gathered blocks differ from their owners' results: {sums_seen.tolist()} vs {owners.tolist()}")
    ver = None
    if dist.get_backend() == "nccl":
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())  # RCCL's ncclGetVersion through torch
        if not ver:
            raise SystemExit("RCCL version query failed")
    return {"gathered_blocks_equal_owners": True, "ranks": world, "rccl_version": ver, "backend": dist.get_backend()}


def sequences_bench(V, args, device, n_seq=64, n_frames=40):
    """configs[0] / configs[2]'s shape — SEQUENCES, not pairs: 64 sequences of 40 frames advancing in lock-step through vors_trackers_*
    (the whole Tracker::track state machine incl. per-sequence keyframe promotion on the device, no host round trip), frames resident in
    HBM; beside it the CPU oracle's Tracker on ONE pinned core over one of the sequences, and the pose agreement on all of them."""
    from oracle import oracle as O
    rows, cols, L = args.rows, args.cols, args.levels
    intr = V.scaled_intrinsics(rows, cols)
    base = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
    rng = np.random.default_rng(11)
    speed = 0.5 + 1.0 * rng.random(n_seq)
    sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
    out = {}
    for mode in ("c2f", "dso", "dense"):
        blocky = (1 << 63) if mode == "dso" else 0
        frames = []
        for k in range(n_frames):
            frames.append(V.synth_render_frames([blocky | (4242 + s) for s in range(n_seq)], [k] * n_seq,
                                                [base * sign[s] * speed[s] * k for s in range(n_seq)], rows, cols, intr, device=device))
        mode_id = {"c2f": V.CANDIDATES_COARSE_TO_FINE, "dso": V.CANDIDATES_DSO, "dense": V.CANDIDATES_DENSE}[mode]
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode_id,
                       arithmetic=arith_id(V, args.arith))
        tr = V.Trackers(cfg, n_seq, rows, cols)
        for _ in range(2):  # the first pass warms up; the second is timed
            tr.init(*frames[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(1, n_frames):
                tr.track(*frames[k])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        # one more pass that reads the poses back every frame (what a host that writes trajectories does): the parity trajectory
        tr.init(*frames[0])
        traj, switches, kf_prev = [], 0, np.zeros(n_seq, np.int32)
        for k in range(1, n_frames):
            tr.track(*frames[k])
            poses, status, kf = tr.current_frames()
            traj.append(poses)
            switches += int((kf != kf_prev).sum())
            kf_prev = kf
        traj = np.stack(traj, axis=1)  # [n_seq, n_frames-1, 7]
        # CPU oracle: every sequence on all cores for the parity (and its f64-accumulation build for the floor); ONE sequence on one
        # pinned core for the rate
        gh = np.stack([g.cpu().numpy() for g, _ in frames])
        dh = np.stack([d.cpu().numpy().view(np.uint16) for _, d in frames])
        ocfg = O.make_config(L, intr, candidates_mode=mode_id)
        nt = min(os.cpu_count() or 1, n_seq)
        ref = O.track_sequences(ocfg, gh, dh, n_threads=nt)
        ref64 = O.track_sequences(ocfg, gh, dh, n_threads=nt, variant="acc64")
        try:
            allowed = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, {allowed[len(allowed) // 2]})
        except (AttributeError, OSError):
            allowed = None
        try:
            O.track_sequences(ocfg, gh[:3, :1], dh[:3, :1], n_threads=1, variant="native")  # warm-up
            t0 = time.perf_counter()
            O.track_sequences(ocfg, gh[:, :1], dh[:, :1], n_threads=1, variant="native")
            t_cpu = time.perf_counter() - t0
        finally:
            if allowed is not None:
                os.sched_setaffinity(0, set(allowed))
        err = np.abs(traj - ref["poses"]).max(axis=(1, 2))        # per sequence: the worst frame of its accumulated trajectory
        err64 = np.abs(ref64["poses"] - ref["poses"]).max(axis=(1, 2))
        fps = n_seq * (n_frames - 1) / dt
        cpu_fps = (n_frames - 1) / t_cpu
        # the same sequences in the REFERENCE arithmetic (the reference's summation order): every pose of every frame must equal the
        # oracle tracker's BIT FOR BIT, keyframe decisions included
        ref_mode = None
        if args.arith != "reference":
            import copy
            rcfg = copy.copy(cfg)
            rcfg.arithmetic = V.ARITH_REFERENCE
            rt = V.Trackers(rcfg, n_seq, rows, cols)
            rt.init(*frames[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rtraj = []
            for k in range(1, n_frames):
                rt.track(*frames[k])
                rtraj.append(rt.current_frames()[0])
            dtr = time.perf_counter() - t0
            rtraj = np.stack(rtraj, axis=1)
            same = (rtraj.view(np.uint32) == ref["poses"].view(np.uint32)).all(axis=(1, 2))
            ref_mode = {"n_sequences_bit_identical_to_oracle_tracker": int(same.sum()),
                        "max_pose_diff_vs_oracle_tracker": float(np.abs(rtraj - ref["poses"]).max()),
                        "frames_per_s_incl_readback": round(n_seq * (n_frames - 1) / dtr, 1)}
            del rt
        out[mode] = {"frames_per_s": round(fps, 1), "ms_per_lockstep_frame": round(dt / (n_frames - 1) * 1e3, 4), "sequences": n_seq,
                     "frames_per_sequence": n_frames - 1, "keyframe_switches": switches,
                     "keyframe_switches_oracle": int(ref["changed_keyframe"].sum()),
                     "cpu_oracle_tracker_frames_per_s_1core": round(cpu_fps, 1), "gpu_over_cpu_1core": round(fps / cpu_fps, 1),
                     "max_pose_diff_vs_oracle_tracker": float(err.max()), "median_pose_diff_vs_oracle_tracker": float(np.median(err)),
                     "n_sequences_beyond_tol": int((err > 1e-4).sum()),
                     "n_sequences_beyond_tol_oracle_f32_vs_f64_accumulation": int((err64 > 1e-4).sum()), "sequences_compared": n_seq,
                     "reference_arithmetic": ref_mode}
        del tr, frames, gh, dh
    out["note"] = (f"{args.cols}x{args.rows}, {L} levels, {args.arith} arithmetic; frames resident in HBM; a lock-step frame = one vors_trackers_track call "
                   "for all 64 sequences; the trajectory error is the max over all frames of a sequence (errors accumulate along a sequence)")
    return out


def reference_parity(V, args, device, seed0, sizes):
    """VORS_ARITH_REFERENCE (the reference's summation order on the device) over full-size samples of the three candidate modes against
    the oracle: the expectation is EQUALITY OF BITS — every pose, every iteration count — so the block reports counts of identical
    pairs, not a tolerance; plus the mode's own throughput."""
    import copy
    from oracle import oracle as O
    out = {}
    ncores = os.cpu_count() or 1
    def one(mode, a, n, reps):
        w = Workload(V, a, mode, device, seed0, pairs=n)
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            w.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kg, kd, cg = host_pairs(w, n)
        ref = O.track_pairs(O.make_config(a.levels, w.intr, candidates_mode=w.mode_id, huber_delta=a.huber), kg, kd, cg, n_threads=min(ncores, n))
        poses = w.poses.cpu().numpy()
        st = V.decode_stats(w.stats)
        L = a.levels
        same_pose = (poses.view(np.uint32) == ref["poses"].view(np.uint32)).all(axis=1)
        same_model = (np.ascontiguousarray(st["lm_model"]).view(np.uint32) == ref["models"].view(np.uint32)).all(axis=1)
        same_iter = (st["nb_iter"][:, :L] == ref["nb_iter"]).all(axis=1)
        same_flow = np.ascontiguousarray(st["optical_flow"]).view(np.uint32) == ref["flow"].view(np.uint32)
        err = np.abs(poses - ref["poses"]).max(axis=1)
        res = {"sample_pairs": int(n), "n_poses_bit_identical": int(same_pose.sum()), "n_lm_models_bit_identical": int(same_model.sum()),
               "n_iteration_counts_equal_at_every_level": int(same_iter.sum()), "n_optical_flow_bit_identical": int(same_flow.sum()),
               "branch_flip_rate_gpu_vs_oracle": float((~same_iter).mean()), "n_beyond_tol": int((err > 1e-4).sum()),
               "max_pose_diff_gpu_vs_oracle": float(err.max(initial=0.0)),
               "status_equal": bool((w.status.cpu().numpy() == ref["status"]).all()),
               "frame_pairs_per_s": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 3)}
        del w
        return res

    for mode in ("c2f", "dso", "dense"):
        n = min(sizes[mode], args.pairs)
        if n <= 0:
            continue
        a = copy.copy(args)
        a.arith = "reference"
        out[mode] = one(mode, a, n, 3 if mode != "dense" else 1)
    if (args.rows, args.cols, args.levels) == (480, 640, 6) and min(sizes["dense"], args.pairs) >= 32:
        # BASELINE config 5's shape (1280x960, 7 levels, Huber 10 — the extension, defined by the oracle) and config 3's candidates at that
        # shape: a small sample each, the same equality of bits
        a = copy.copy(args)
        a.arith, a.rows, a.cols, a.levels, a.huber = "reference", 960, 1280, 7, 10.0
        out["config5_shape_dense_huber10"] = one("dense", a, 32, 1)
        out["config5_shape_c2f_huber10"] = one("c2f", a, 64, 1)
    out["note"] = ("arithmetic = reference: EXACT's per-point arithmetic + the reference's sequential f32 sums in extract_z's column-major order "
                   "(lm_reference.hip); compared with the oracle bit for bit")
    return out


def measure_block(V, a, mode, device, seed0, ring, steps, warmup, pmc, limited_by):
    """One workload (mode x arithmetic x shape of `a`) at a.pairs pairs: throughput, stage times (HIP events on the stream), the LM stage's
    algorithmic-byte rate against the HBM peak and — pmc — its live counter traffic (three short rocprofv3 passes of the same command)."""
    w = Workload(V, a, mode, device, seed0)
    w.batch.enable_kernel_timing(ring)
    dt = timed_run(w, steps, warmup, 1, None, None)
    st = V.decode_stats(w.stats)
    io, lmb, _, ev, _ = byte_model(st, a.levels, a.rows, a.cols, mode == "dense")
    lm = float(w.batch.kernel_times("lm")[-steps:].mean())
    kf = float(w.batch.kernel_times("keyframe")[-steps:].mean())
    py = float((w.batch.kernel_times("pyramid_keyframe")[-steps:] + w.batch.kernel_times("pyramid_current")[-steps:]).mean())
    lm_bytes = lmb + 32 * a.pairs
    cnt = live_counters(a, mode) if pmc else {}
    tr = cnt.get("traffic_bytes")
    kernel = ("lm_ref_track_kernel (+ lm_ref_track_coop_kernel for small batches and the pairs handed over)" if a.arith == "reference" else
              ("lm_track_kernel (coarse levels) + lm_split_eval_kernel / lm_split_step_kernel rounds" if mode == "dense" else "lm_track_kernel"))
    blk = {"value": round(a.pairs * steps / dt, 2), "unit": "frame-pairs/s", "pairs_per_gpu": a.pairs, "ms_per_step": round(dt / steps * 1e3, 4),
           "stages_ms": {"pyramids": round(py, 5), "keyframe": round(kf, 5), "lm": round(lm, 5)},
           "lm_evals_per_pair": round(ev, 2), "failed_pairs": int((w.status != 0).sum().item()),
           "roofline": {"bound": "hbm", "limited_by": limited_by, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "kernel": kernel,
                        "achieved": round(lm_bytes / (lm * 1e-3) / 1e9, 2), "frac": round(lm_bytes / (lm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                        "algorithmic_bytes_per_launch": lm_bytes, "kernel_ms_avg": round(lm, 5),
                        "traffic": tr, "traffic_source": cnt.get("source"), "traffic_over_algorithmic": (round(tr / lm_bytes, 3) if tr else None),
                        "memory_side_frac": (round(tr / (lm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if tr else None),
                        "valu_instructions_per_launch": cnt.get("sq_insts_valu"),
                        "io_only_GBps": round(io * steps / dt / 1e9, 2), "io_only_frac": round(io * steps / dt / 1e9 / HBM_PEAK_GBPS, 5),
                        "whole_job_frac": round((io + lmb) * steps / dt / 1e9 / HBM_PEAK_GBPS, 5)}}
    return blk, w, dt


def reference_block(V, args, device, seed0, ring):
    """The three candidate modes in VORS_ARITH_REFERENCE — the boundary's default arithmetic (vors_config.arithmetic = 0), the one whose
    poses are bit-identical to the oracle's (`parity_reference`) — at the headline batch size: throughput, stage times, the LM stage's
    algorithmic-byte rate against the HBM peak, and the 512-pair step (BASELINE config 4's per-GPU share) with its step-time ratio."""
    import copy
    a = copy.copy(args)
    a.arith = "reference"
    out = {"arithmetic": "reference (VORS_ARITH_REFERENCE = 0: the reference's per-point arithmetic and summation order)"}
    for mode in ("c2f", "dso", "dense"):
        steps = args.steps if mode != "dense" else max(2, min(args.steps, 5))   # (a dense step is ~70 ms at 4096 pairs)
        blk, w, dt = measure_block(V, a, mode, device, seed0, ring, steps, min(args.warmup, 2), not args.no_pmc,
                                   "valu (dependent f32 chains + the reference's per-point expressions)")
        del w
        if a.pairs != 512:
            a5 = copy.copy(a)
            a5.pairs = 512
            w5 = Workload(V, a5, mode, device, seed0)
            w5.batch.enable_kernel_timing(ring)
            dt5 = timed_run(w5, steps, min(args.warmup, 2), 1, None, None)
            blk["batch_512"] = {"value": round(512 * steps / dt5, 2), "ms_per_step": round(dt5 / steps * 1e3, 4),
                                "lm_kernel_ms": round(float(w5.batch.kernel_times("lm")[-steps:].mean()), 5),
                                "step_time_ratio_vs_headline_batch": round((dt / steps) / (dt5 / steps), 3)}
            del w5
            # throughput mode (ring of 3 handles, see `pipelined` of the headline): the default arithmetic gains most — the one-wavefront kernel's
            # straggler tail and the dependent chains of a small batch are what the next step fills
            blk["pipelined"] = ring_block(V, a, mode, device, seed0, a.pairs, steps, dt / steps * 1e3)
            blk["batch_512"]["pipelined"] = ring_block(V, a5, mode, device, seed0, 512, steps, dt5 / steps * 1e3)
            blk["batch_512"]["step_time_ratio_pipelined_vs_pipelined_headline_batch"] = round(
                blk["pipelined"]["ms_per_step"] / blk["batch_512"]["pipelined"]["ms_per_step"], 3)
        out[mode] = blk
    return out


def config5_block(V, args, device, ring):
    """BASELINE configs[4] ("config 5") on ONE GPU's share: synthetic 1280x960 RGB-D, 7-level pyramid, Huber weighting (delta = 10 grey
    levels, the oracle-defined extension), dense candidates, 512 pairs (i = 0..511) — in FUSED (what `value` times at 640x480)
    and in REFERENCE (the default arithmetic): throughput, stage times, the LM stage's roofline with live PMC traffic, and the oracle on one
    pinned host core on a sample of the same pairs."""
    import copy
    a = copy.copy(args)
    a.rows, a.cols, a.levels, a.huber, a.pairs = 960, 1280, 7, 10.0, 512
    out = {"workload": "BASELINE configs[4]: synthetic 1280x960 RGB-D, 7-level pyramid, Huber delta 10, dense candidates, 512 pairs on one GPU"}
    for arith in ("fused", "reference"):
        a.arith = arith
        steps = max(2, min(args.steps, 10 if arith == "fused" else 4))
        blk, w, dt = measure_block(V, a, "dense", device, 0x5EED0000, ring, steps, 1, not args.no_pmc,
                                   "valu" if arith == "fused" else "valu (dependent f32 chains + the reference's per-point expressions)")
        if arith == "fused" and args.cpu_pairs != 0:
            blk["cpu_baseline"] = cpu_baseline(a, w, blk["value"], n_override=8 if args.cpu_pairs < 0 else min(args.cpu_pairs, 8))
        if arith == "reference" and args.parity_pairs != 0:   # the default arithmetic at this shape: equality of bits on a sample
            from oracle import oracle as O
            n = 16
            kg, kd, cg = host_pairs(w, n)
            ref = O.track_pairs(O.make_config(a.levels, w.intr, candidates_mode=w.mode_id, huber_delta=a.huber), kg, kd, cg, n_threads=min(os.cpu_count() or 1, n))
            poses = w.poses[:n].cpu().numpy()
            blk["parity_sample"] = {"sample_pairs": n, "n_poses_bit_identical": int((poses.view(np.uint32) == ref["poses"].view(np.uint32)).all(axis=1).sum()),
                                    "status_equal": bool((w.status[:n].cpu().numpy() == ref["status"]).all())}
        del w
        out[arith] = blk
    return out


def single_tracker_bench(V, args, device, n_frames=40):
    """configs[0] / configs[2] as written are ONE sequence: vors_tracker_* (Config::init / Tracker::track / current_frame through the C ABI,
    HOST buffers in, pose out — upload, pyramid, LM, keyframe test and promotion per call), milliseconds per frame for the three candidate
    modes in both arithmetics, beside the oracle's Tracker on one pinned host core over the same frames."""
    from oracle import oracle as O
    rows, cols, L = args.rows, args.cols, args.levels
    intr = V.scaled_intrinsics(rows, cols)
    step = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
    out = {}
    for mode in ("c2f", "dso", "dense"):
        blocky = (1 << 63) if mode == "dso" else 0
        g, d = V.synth_render_frames([blocky | 31337] * n_frames, list(range(n_frames)), [step * k for k in range(n_frames)], rows, cols, intr, device=device)
        gh, dh = g.cpu().numpy(), d.cpu().numpy().view(np.uint16)
        mode_id = {"c2f": V.CANDIDATES_COARSE_TO_FINE, "dso": V.CANDIDATES_DSO, "dense": V.CANDIDATES_DENSE}[mode]
        blk = {}
        last = {}
        for arith in ("reference", "fused"):
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode_id, arithmetic=arith_id(V, arith))
            for rep in range(2):   # the first pass warms up
                vt = cfg.init(0.0, dh[0], 0.0, gh[0])
                ts, traj = [], []
                for k in range(1, n_frames):
                    t0 = time.perf_counter()
                    vt.track(float(k), dh[k], float(k), gh[k])
                    ts.append(time.perf_counter() - t0)
                    traj.append(vt.current_frame()[1])
                del vt
            ts = np.array(ts) * 1e3
            last[arith] = np.array(traj, np.float32)
            blk[arith] = {"ms_per_frame_mean": round(float(ts.mean()), 4), "ms_per_frame_median": round(float(np.median(ts)), 4),
                          "ms_per_frame_p90": round(float(np.quantile(ts, 0.9)), 4)}
        ocfg = O.make_config(L, intr, candidates_mode=mode_id)
        try:
            allowed = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, {allowed[len(allowed) // 2]})
        except (AttributeError, OSError):
            allowed = None
        try:
            O.track_sequences(ocfg, gh[:3, None], dh[:3, None], n_threads=1, variant="native")  # warm-up
            t0 = time.perf_counter()
            O.track_sequences(ocfg, gh[:, None], dh[:, None], n_threads=1, variant="native")
            t_cpu = time.perf_counter() - t0
        finally:
            if allowed is not None:
                os.sched_setaffinity(0, set(allowed))
        ref = O.track_sequences(ocfg, gh[:, None], dh[:, None], n_threads=1)
        blk["cpu_oracle_tracker_ms_per_frame_1core"] = round(t_cpu / (n_frames - 1) * 1e3, 4)
        blk["reference_trajectory_bit_identical_to_oracle_tracker"] = bool((last["reference"].view(np.uint32) == ref["poses"][0].view(np.uint32)).all())
        blk["fused_max_pose_diff_vs_oracle_tracker"] = float(np.abs(last["fused"] - ref["poses"][0]).max())
        out[mode] = blk
    out["note"] = (f"{cols}x{rows}, {L} levels, {n_frames - 1} tracked frames of one synthetic sequence; a frame = one vors_tracker_track call with HOST buffers "
                   "(upload + pyramid + LM + keyframe test + read-back); the oracle figure is its C++ Tracker on one pinned core, kind \"port\"")
    return out


def parity_sample_sizes(args):
    """Pairs compared per candidates mode: 1024 dense / 4096 sparse at 640x480 on a many-core host (the oracle takes ~15 s / ~3 s on 256
    threads), scaled down with the host's core count and up-sized images so that the default run stays within minutes."""
    if args.parity_pairs >= 0:
        return {"dense": args.parity_pairs, "c2f": args.parity_pairs, "dso": args.parity_pairs}
    ncores = os.cpu_count() or 1
    scale = (args.rows * args.cols) / (480.0 * 640.0)
    return {"dense": int(max(48, min(1024, 4 * ncores)) / scale), "c2f": int(max(192, min(4096, 16 * ncores)) / scale),
            "dso": int(max(192, min(4096, 16 * ncores)) / scale)}


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

    pairs_weak = args.pairs
    if args.scaling == "strong":  # BASELINE configs[3] as written: a fixed batch sharded over the ranks (equal blocks: the gather wants them equal)
        args.pairs = -(-args.total_pairs // world)
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
    total_pairs = (args.total_pairs if args.scaling == "strong" else world * args.pairs) * args.steps
    value = total_pairs / dt

    stats = V.decode_stats(main_w.stats)
    dense = args.candidates == "dense"
    b_io, b_lm, b_lm_flat, evals_per_pair, ptevals_per_pair = byte_model(stats, args.levels, args.rows, args.cols, dense)
    lm_avg_s = float(lm_ms.mean()) * 1e-3
    lm_bytes = b_lm + 32 * args.pairs  # algorithmic bytes of the LM stage of ONE step (this rank's batch)
    achieved = lm_bytes / lm_avg_s / 1e9
    job_gbps = (b_io + b_lm) * world * args.steps / dt / 1e9
    gt_err = np.abs(stats["lm_model"] - main_w.gt.cpu().numpy()).max(axis=1)
    cnt = {}
    if rank == 0 and world == 1 and not args.no_pmc:
        cnt = live_counters(args)
    if "traffic_bytes" not in cnt:
        cnt = lm_counters(args) or cnt
    clock_hz = V.device_info(dev_index)["clock_khz"] * 1e3  # hipDeviceAttributeClockRate: the peak shader clock
    base_shape = (args.rows, args.cols, args.levels) == (480, 640, 6)

    traffic = cnt.get("traffic_bytes")
    roofline = {
        # `bound` names the roofline `achieved` / `peak` / `frac` are taken against (the bench contract knows "hbm" and "mfma"): SURVEY
        # §8(d)'s algorithmic bytes over the stage's measured time vs the 8 TB/s HBM peak. What LIMITS the stage is `limited_by`: VALU
        # issue — its memory side is at `memory_side_frac` of the peak.
        "bound": "hbm",
        "limited_by": "valu",
        "model": "hbm: SURVEY §8(d) algorithmic bytes / measured stage time vs 8 TB/s",
        # dense mode: the LM stage is a short sequence of launches (coarse levels per pair, then one launch per energy evaluation round on
        # the finest levels + a per-pair step launch, then the per-pair epilogue); it is timed as a whole with HIP events on its stream,
        # and its algorithmic bytes are those of all its evaluations
        "kernel": ("LM stage: lm_track_kernel (coarse levels) + lm_split_eval_kernel / lm_split_step_kernel per evaluation round "
                   "+ lm_track_kernel (epilogue)") if dense else "lm_track_kernel",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 5),
        # HBM bytes of the same stage per step from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; separate rocprofv3 --pmc passes)
        "traffic": traffic,
        "traffic_source": cnt.get("source"),
        "traffic_over_algorithmic": (round(traffic / lm_bytes, 3) if traffic else None),
        # the memory side of the same stage: counter bytes / stage time / peak
        "memory_side_frac": (round(traffic / lm_avg_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
        "algorithmic_bytes_per_launch": lm_bytes,
        "algorithmic_bytes_flat_model": b_lm_flat + 32 * args.pairs,
        "frac_flat_model": round((b_lm_flat + 32 * args.pairs) / lm_avg_s / 1e9 / HBM_PEAK_GBPS, 5),
        "kernel_ms_avg": round(lm_avg_s * 1e3, 5),
        "whole_job_GBps": round(job_gbps, 2),
        "whole_job_frac": round(job_gbps / (HBM_PEAK_GBPS * world), 5),
        "io_only_GBps": round(b_io * world * args.steps / dt / 1e9, 2),
        "binding_resource": "VALU issue",
        # wave64 VALU instructions issued (SQ_INSTS_VALU) x 2 cycles (the best-case issue interval, tools/ubench) over the SIMD-cycles of
        # the stage at the device's PEAK clock (hipDeviceAttributeClockRate; a throttled clock makes the true fraction larger)
        "shader_clock_hz_peak": clock_hz,
        "valu_issue_frac": (round(cnt["sq_insts_valu"] * 2.0 / (N_SIMD * lm_avg_s * clock_hz), 4) if "sq_insts_valu" in cnt else None),
        # the same with the issue interval MEASURED on this chip for the fastest class (v_fma / v_mul / v_add: one wave64 instruction per
        # 2.8 cycles per SIMD, tools/ubench/valu_ops, pk_ops; v_pk_fma_f32 5.06 per pair; conversions / compares / selects / DPP 4.4)
        "valu_issue_frac_at_measured_2p8_cycles": (round(cnt["sq_insts_valu"] * 2.8 / (N_SIMD * lm_avg_s * clock_hz), 4)
                                                   if "sq_insts_valu" in cnt else None),
        # share of the wavefronts' resident time in which a VALU instruction of theirs is executing (both counters are per-wave quad-cycle
        # sums: a ratio of like units, <= 1 by construction — the r02 `valu_busy_frac` divided a per-wave sum by SIMD-cycles and could exceed 1)
        "valu_active_over_wave_cycles": (round(cnt["sq_active_inst_valu"] / cnt["sq_wave_cycles"], 4)
                                         if cnt.get("sq_wave_cycles") and "sq_active_inst_valu" in cnt else None),
    }
    if roofline["valu_active_over_wave_cycles"] is not None:
        assert roofline["valu_active_over_wave_cycles"] <= 1.0, roofline
    out = {
        "metric": "frame-pairs/sec (640x480, 6 pyramid levels)" if base_shape else f"frame-pairs/sec ({args.cols}x{args.rows}, {args.levels} pyramid levels)",
        "value": round(value, 2),
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[3]: {args.total_pairs} pairs sharded over {world} ranks (synthetic 640x480 RGB-D, 6-level pyramid, {args.candidates} candidates)"
                         if world > 1 and args.scaling == "strong" and base_shape and args.huber == 0 else
                         "BASELINE configs[1]: synthetic 640x480 RGB-D (gray u8 + depth u16), 6-level pyramid, dense candidates"
                         if dense and base_shape and args.huber == 0 else
                         ("BASELINE configs[4] shape: synthetic 1280x960 RGB-D, 7-level pyramid, Huber weighting, dense candidates"
                          if dense and (args.rows, args.cols, args.levels) == (960, 1280, 7) and args.huber > 0 else
                          f"synthetic {args.cols}x{args.rows} gray u8 + depth u16, {args.levels}-level pyramid, {args.candidates} candidates")),
            "pairs_per_gpu": args.pairs,
            "total_pairs": (args.total_pairs if args.scaling == "strong" else world * args.pairs),
            "candidates": {"dense": "dense (all-true level-0 mask, extension)", "c2f": "coarse_to_fine (reference selection)",
                           "dso": "DSO-style selection (dso.rs, examples/candidates_dso.rs parameters)"}[args.candidates],
            "huber_delta": args.huber,
            "arithmetic": args.arith,
            "parallelism": (f"pairs sharded over {world} ranks, one all-gather of pose+status per step "
                            f"({'RCCL over xGMI, one GPU per rank' if args.backend == 'nccl' else 'gloo TEST backend, ranks share ' + str(n_dev) + ' GPU(s)'})"
                            if world > 1 else "1 GPU"),
            "launch": "hipGraph replay" if args.graph else "eager",
            "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
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

    parity_secondary = None
    if world > 1:
        out["self_check"] = multi_gpu_self_check(args, main_w, packed, gathered, rank, world)
        # ---- N > 1: BOTH readings of "throughput at N GPUs" in the one line. `weak` = --pairs per GPU whatever N (the
        # driver's default contract: per-GPU work fixed); `strong` = BASELINE configs[3] AS WRITTEN: the fixed batch of --total-pairs (4096)
        # sharded over the N ranks, 4096 / N each. `value` is the one --scaling names (default weak); the other one is timed right after
        # it with the same steps / warm-up / barrier + MAX-over-ranks rule and its own gather self-check.
        def scaling_block(kind, work, dt_k, check):
            per = work.n
            total = args.total_pairs if kind == "strong" else world * per
            return {"workload": (f"BASELINE configs[3]: {args.total_pairs} pairs sharded over {world} ranks ({per} per GPU)" if kind == "strong" else
                                 f"{per} pairs per GPU on {world} ranks ({total} pairs per step)"),
                    "scaling": kind, "value": round(total * args.steps / dt_k, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt_k / args.steps * 1e3, 4),
                    "pairs_per_gpu": per, "total_pairs": total, "failed_pairs_rank0": int((work.status != 0).sum().item()), "self_check": check}
        out[args.scaling] = scaling_block(args.scaling, main_w, dt, out["self_check"])

        def pipelined_scaling(a_k, kind):
            # the same per-rank share as a continuous feed through a ring of 3 handles (vors_pipeline_*), the all-gather of a step's 8 f32 per
            # pair issued once that step has completed in stream order (two steps behind the submissions); barrier + MAX over ranks as above
            import torch.distributed as dist
            from vors_amd.distributed import gather_packed
            # (an extra measurement must never take the run down: the ring and its buffers are set up under a guard, and the ranks agree —
            # one MIN all-reduce — that everybody has them before anyone enters the collective loop)
            r = pk = ga = None
            err = None
            try:
                r = Ring(V, a_k, args.candidates, device, 0x5EED0000 + rank * a_k.pairs, a_k.pairs, 3)
                pk = torch.zeros((a_k.pairs, 8), dtype=torch.float32, device=device)
                ga = torch.zeros((world * a_k.pairs, 8), dtype=torch.float32, device=device)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                del r, pk, ga
                return {"ring": 3, "value": None, "error": err or "another rank could not set up its ring"}

            def after(i, w):
                pk[:, :7] = w.poses
                pk[:, 7] = w.status
                gather_packed(pk, out=ga)
            dt_p = r.run(args.steps, max(args.warmup, 3), after, barrier=dist.barrier)
            t = torch.tensor([dt_p], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_p = float(t.item())
            total = args.total_pairs if kind == "strong" else world * a_k.pairs
            del r
            return {"ring": 3, "value": round(total * args.steps / dt_p, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt_p / args.steps * 1e3, 4),
                    "note": "vors_pipeline_* ring of 3 handles per rank + the per-step all-gather; not `value`"}
        out[args.scaling]["pipelined"] = pipelined_scaling(args, args.scaling)
        other = "strong" if args.scaling == "weak" else "weak"
        import copy
        a2 = copy.copy(args)
        a2.pairs = -(-args.total_pairs // world) if other == "strong" else pairs_weak
        w2 = Workload(V, a2, args.candidates, device, 0x5EED0000 + rank * a2.pairs)
        packed2 = torch.zeros((a2.pairs, 8), dtype=torch.float32, device=device)
        gathered2 = torch.zeros((world * a2.pairs, 8), dtype=torch.float32, device=device)
        dt_o = timed_run(w2, args.steps, args.warmup, world, packed2, gathered2)
        out[other] = scaling_block(other, w2, dt_o, multi_gpu_self_check(a2, w2, packed2, gathered2, rank, world))
        del w2
        w2 = None
        out[other]["pipelined"] = pipelined_scaling(a2, other)
        out["config"]["value_is"] = f"`{args.scaling}` (see the `weak` and `strong` blocks; `strong` is BASELINE configs[3] as written)"
        del packed2, gathered2
    def guarded(key, fn):
        # Everything after the headline is an EXTRA measurement: one that fails (a box with less memory, a missing tool) is recorded in
        # `errors` — loudly, in the line — instead of taking the line, and with it `value`, `roofline` and the other blocks, down.
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001
            out.setdefault("errors", {})[key] = f"{type(e).__name__}: {e}"
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass

    if rank == 0 and world == 1:
        try:
            if not args.no_secondary:
                # ---- SURVEY §8d batch size: 256 pairs (one per CU) of the same workload, and BASELINE config 4's per-GPU share on 8 GPUs (512)
                for nb in (256, 512):
                    if args.pairs != nb:
                        ws = Workload(V, args, args.candidates, device, seed0, pairs=nb)
                        ws.batch.enable_kernel_timing(ring)
                        dts = timed_run(ws, args.steps, args.warmup, 1, None, None)
                        out[f"batch_{nb}"] = {"value": round(nb * args.steps / dts, 2), "unit": "frame-pairs/s", "pairs_per_gpu": nb,
                                              "ms_per_step": round(dts / args.steps * 1e3, 4),
                                              "lm_kernel_ms": round(float(ws.batch.kernel_times("lm")[-args.steps:].mean()), 5),
                                              "step_time_ratio_vs_headline_batch": round((dt / args.steps) / (dts / args.steps), 3)}
                        del ws
                # ---- the other candidate modes, each with its own stage times, roofline block (I/O-only = SURVEY's conservative sparse headline,
                # whole job, LM stage; live PMC traffic), CPU baseline (1 pinned core on a sample of the same pairs) and full-batch parity:
                # `secondary` = the reference's own selection (coarse-to-fine) when the headline is dense, `dso` = config 3's selector
                for key, other in (("secondary", "c2f" if dense else "dense"), ("dso", "dso")):
                    if other == args.candidates:
                        continue
                    w2 = Workload(V, args, other, device, seed0)
                    w2.batch.enable_kernel_timing(ring)
                    dt2 = timed_run(w2, args.steps, args.warmup, 1, None, None)
                    st2 = V.decode_stats(w2.stats)
                    io2, lmb2, lmflat2, ev2, _ = byte_model(st2, args.levels, args.rows, args.cols, other == "dense")
                    lm2 = float(w2.batch.kernel_times("lm")[-args.steps:].mean())
                    kf2 = float(w2.batch.kernel_times("keyframe")[-args.steps:].mean())
                    py2 = float((w2.batch.kernel_times("pyramid_keyframe")[-args.steps:] + w2.batch.kernel_times("pyramid_current")[-args.steps:]).mean())
                    val2 = args.pairs * args.steps / dt2
                    cnt2 = live_counters(args, other) if not args.no_pmc else {}
                    tr2 = cnt2.get("traffic_bytes")
                    lm_bytes2 = lmb2 + 32 * args.pairs
                    out[key] = {
                        "candidates": {"c2f": "coarse_to_fine (reference selection)", "dense": "dense", "dso": "DSO-style selection (config 3)"}[other],
                        "value": round(val2, 2), "unit": "frame-pairs/s",
                        "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                        "stages_ms": {"pyramids": round(py2, 5), "keyframe": round(kf2, 5), "lm": round(lm2, 5)},
                        "lm_kernel_ms": round(lm2, 5),
                        "lm_evals_per_pair": round(ev2, 2),
                        "roofline": {
                            "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "io_only_GBps": round(io2 * args.steps / dt2 / 1e9, 2), "io_only_frac": round(io2 * args.steps / dt2 / 1e9 / HBM_PEAK_GBPS, 5),
                            "whole_job_GBps": round((io2 + lmb2) * args.steps / dt2 / 1e9, 2),
                            "whole_job_frac": round((io2 + lmb2) * args.steps / dt2 / 1e9 / HBM_PEAK_GBPS, 5),
                            "lm_stage_achieved": round(lm_bytes2 / (lm2 * 1e-3) / 1e9, 2), "lm_stage_frac": round(lm_bytes2 / (lm2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                            "lm_stage_algorithmic_bytes": lm_bytes2, "traffic": tr2, "traffic_source": cnt2.get("source"),
                            "traffic_over_algorithmic": (round(tr2 / lm_bytes2, 3) if tr2 else None),
                            "note": "sparse modes: SURVEY §8(d) takes the I/O-only fraction as the conservative headline (the candidate lists are cache-sized)",
                        },
                    }
                    if args.cpu_pairs != 0:
                        out[key]["cpu_baseline"] = cpu_baseline(args, w2, val2, n_override=(48 if other == "dense" else 256) if args.cpu_pairs < 0 else args.cpu_pairs)
                    n_par = parity_sample_sizes(args)[other]
                    if n_par > 0:
                        pb = parity_block(args, w2, n_par)
                        if key == "secondary":
                            parity_secondary = pb
                        else:
                            out["parity_dso"] = pb
                    del w2
                # ---- THROUGHPUT MODE (vors_pipeline_*, the C ABI's ring of batch handles on internal streams): the same steps as a continuous feed,
                # consecutive steps on consecutive handles — each step is still one full pass over its own batch; step k + 1's pyramids and keyframe
                # stage run under step k's LM stage and its straggler tail. `value` above stays the single-stream figure (its stage times and
                # roofline are self-consistent); this is what a deployment that feeds batches back to back gets, and what BASELINE config 4's
                # per-GPU share (512 pairs) needs: a 512-pair step alone leaves most of the chip idle.
                single_ms = dt / args.steps * 1e3
                out["pipelined_two_streams"] = ring_block(V, args, args.candidates, device, seed0, args.pairs, args.steps, single_ms, depth=2)
                out["pipelined"] = ring_block(V, args, args.candidates, device, seed0, args.pairs, args.steps, single_ms, depth=3)
                out["pipelined"]["note"] = ("vors_pipeline_* with a ring of 3 handles (pipelined_two_streams: ring of 2); a step = one full pass over its "
                                            "own batch; not the headline")
                if "batch_512" in out and args.pairs != 512:
                    out["batch_512"]["pipelined"] = ring_block(V, args, args.candidates, device, seed0, 512, args.steps, out["batch_512"]["ms_per_step"])
                    out["batch_512"]["step_time_ratio_pipelined_vs_pipelined_headline_batch"] = round(
                        out["pipelined"]["ms_per_step"] / out["batch_512"]["pipelined"]["ms_per_step"], 3)
                for key, other in (("secondary", "c2f" if dense else "dense"), ("dso", "dso")):
                    if key in out and args.pairs != 512:   # the other candidate modes: the 512-pair step, single stream and ring, and both ratios
                        ms4096 = out[key]["ms_per_step"]
                        w5 = Workload(V, args, other, device, seed0, pairs=512)
                        dt5 = timed_run(w5, args.steps, args.warmup, 1, None, None)
                        del w5
                        p4096 = ring_block(V, args, other, device, seed0, args.pairs, args.steps, ms4096)
                        p512 = ring_block(V, args, other, device, seed0, 512, args.steps, dt5 / args.steps * 1e3)
                        out[key]["pipelined"] = p4096
                        out[key]["batch_512"] = {"value": round(512 * args.steps / dt5, 2), "ms_per_step": round(dt5 / args.steps * 1e3, 4),
                                                 "step_time_ratio_vs_headline_batch": round(ms4096 / (dt5 / args.steps * 1e3), 3),
                                                 "pipelined": p512,
                                                 "step_time_ratio_pipelined_vs_pipelined_headline_batch": round(p4096["ms_per_step"] / p512["ms_per_step"], 3)}
        except Exception as e:  # noqa: BLE001
            out.setdefault("errors", {})["secondary_legs"] = f"{type(e).__name__}: {e}"
        if not args.no_secondary and args.arith != "reference":
            guarded("reference", lambda: reference_block(V, args, device, seed0, ring))
        if not args.no_sequences and not args.no_secondary:
            guarded("sequences_64", lambda: sequences_bench(V, args, device))
            guarded("single_tracker", lambda: single_tracker_bench(V, args, device))
        if not args.no_secondary and base_shape and args.huber == 0 and not args.no_config5:
            guarded("config5", lambda: config5_block(V, args, device, ring))
        if not args.no_secondary and args.parity_pairs != 0 and args.arith != "reference":
            sizes = parity_sample_sizes(args)
            guarded("parity_reference", lambda: reference_parity(V, args, device, seed0, sizes))
        if args.cpu_pairs != 0:
            guarded("cpu_baseline", lambda: cpu_baseline(args, main_w, value))
        n_par = parity_sample_sizes(args)[args.candidates]
        if n_par > 0:
            # the headline arithmetic over a FULL-SIZE sample of the batch the timed steps ran on (+ the same for the secondary workload)
            guarded("parity", lambda: parity_block(args, main_w, n_par))
            if parity_secondary is not None:
                out["parity_secondary"] = parity_secondary
    if rank == 0:
        # "bit-identical to the oracle" is NOT "bit-identical to vors" until a document produced by the Rust reference says so
        from oracle.rust_pin import parity_pinned
        pinned, detail = parity_pinned()
        out["parity_pinned"] = pinned
        out["parity_pinned_detail"] = detail
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
