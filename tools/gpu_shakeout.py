"""Quick end-to-end comparison of the HIP path with the oracle (development aid; the real checks live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np
import torch
import vors_amd as V
from oracle import oracle as O

def run(rows, cols, L, n, mode=0, seed0=0x5EED0000):
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_batch(n, rows, cols, seed0=seed0, intr=intr)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode)
    ocfg = O.make_config(L, intr, candidates_mode=mode)
    dev = torch.device("cuda:0")
    b = V.Batch(cfg, n, rows, cols)
    t_kg = torch.from_numpy(kg).to(dev); t_kd = torch.from_numpy(kd.view(np.int16)).to(dev); t_cg = torch.from_numpy(cg).to(dev)
    poses = torch.zeros((n, 7), dtype=torch.float32, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
    stats = V.stats_tensor(n)
    b.track_pairs(t_kg, t_kd, t_cg, poses, status, stats)
    torch.cuda.synchronize()
    t0 = time.time(); ref = O.track_pairs(ocfg, kg, kd, cg); tcpu = time.time() - t0
    st = V.decode_stats(stats)
    # stage checks on pair 0
    tr = O.Tracker(ocfg, 0.0, kd[0], 0.0, kg[0])
    for l in range(L):
        img = b.keyframe_image(0, l)
        assert (img == tr.image(l)).all(), f"pyramid level {l} differs"
        xy, iz, jac, tm = b.points(0, l)
        oxy, oiz, ojac = tr.points(l)
        o1 = np.lexsort((xy[:, 1], xy[:, 0])); o2 = np.lexsort((oxy[:, 1], oxy[:, 0]))
        ok_xy = xy.shape == oxy.shape and (xy[o1] == oxy[o2]).all()
        ok_iz = ok_xy and (iz[o1].view(np.uint32) == oiz[o2].view(np.uint32)).all()
        ok_j = ok_xy and (jac[o1].view(np.uint32) == ojac[o2].view(np.uint32)).all()
        print(f"  level {l}: n={len(iz)} oracle n={len(oiz)} xy={ok_xy} idepth_bits={ok_iz} jac_bits={ok_j}")
        if ok_xy and not ok_j:
            d = np.abs(jac[o1] - ojac[o2]); print("    jac max abs diff", d.max(), "rel", (d / (np.abs(ojac[o2]) + 1e-6)).max())
    p = poses.cpu().numpy()
    err = np.abs(p - ref["poses"]).max(axis=1)
    it_same = (st["nb_iter"][:, :L] == ref["nb_iter"]).all(axis=1)
    print(f"{cols}x{rows} L={L} n={n} mode={mode}: max pose err {err.max():.3g}  pairs>1e-4: {(err > 1e-4).sum()}  iter-identical pairs {it_same.sum()}/{n}"
          f"  status eq {(status.cpu().numpy() == ref['status']).all()}  flow maxdiff {np.abs(st['optical_flow'] - ref['flow']).max():.3g}  cpu {n / tcpu:.1f} pairs/s")
    gterr = np.abs(st["lm_model"] - gt).max(axis=1)
    print(f"   vs ground truth: median {np.median(gterr):.3g} max {gterr.max():.3g}; n_points {st['n_points'][0][:L]} nb_iter {st['nb_iter'][0][:L]} oracle {ref['nb_iter'][0]}")
    return b

if __name__ == "__main__":
    print("devices", V.device_count(), torch.cuda.get_device_name(0))
    run(120, 160, 4, 8)
    run(480, 640, 6, 8)
    run(123, 167, 3, 4)
    run(120, 160, 4, 4, mode=1)
    run(101, 135, 3, 2, mode=1)
    # timing
    rows, cols, L, n = 480, 640, 6, 256
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000, n, rows, cols, intr)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]))
    b = V.Batch(cfg, n, rows, cols); b.enable_kernel_timing(True)
    poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
    stats = V.stats_tensor(n)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"sparse 256 pairs: {dt*1e3:.2f} ms -> {n/dt:.0f} pairs/s", b.last_kernel_ms())
    st = V.decode_stats(stats)
    e = np.abs(st["lm_model"] - gt.cpu().numpy()).max(axis=1)
    print("gt err median", np.median(e), "max", e.max(), "status", status.sum().item(), "evals/pair", (st["nb_iter"][:, :L] + 1).sum(1).mean(), "points", st["n_points"][:, :L].mean(0))
    cfgd = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1)
    n2 = 64
    bd = V.Batch(cfgd, n2, rows, cols); bd.enable_kernel_timing(True)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        bd.track_pairs(kg[:n2], kd[:n2], cg[:n2], poses[:n2], status[:n2], stats)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"dense {n2} pairs: {dt*1e3:.2f} ms -> {n2/dt:.0f} pairs/s", bd.last_kernel_ms())
    st = V.decode_stats(stats)[:n2]
    e = np.abs(st["lm_model"] - gt.cpu().numpy()[:n2]).max(axis=1)
    print("dense gt err median", np.median(e), "max", e.max(), "evals/pair", (st["nb_iter"][:, :L] + 1).sum(1).mean())
