"""Development aid (run through gpurun): are the poses of two builds of libvors_hip identical bit for bit?  Runs the same synthetic batch
through each library in its own interpreter (VORS_HIP_LIB) and compares poses, statuses and per-level iteration counts.
usage: python tools/ab_bits.py LIB_A LIB_B [mode=1] [pairs=256] [arith=fused] [rows cols levels]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = r"""
import sys, os
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L, n, mode = {rows}, {cols}, {L}, {n}, {mode}
intr = V.scaled_intrinsics(rows, cols)
kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
arith = dict(fused=V.ARITH_FUSED, exact=V.ARITH_EXACT, reference=V.ARITH_REFERENCE)[{arith!r}]
b = V.Batch(V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith), n, rows, cols)
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
b.track_pairs(kg, kd, cg, poses, status, stats); torch.cuda.synchronize()
np.savez({out!r}, poses=poses.cpu().numpy(), status=status.cpu().numpy(), raw=stats.cpu().numpy())
"""
la, lb = sys.argv[1], sys.argv[2]
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
arith = sys.argv[5] if len(sys.argv) > 5 else "fused"
rows, cols, L = (int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (480, 640, 6)
res = []
for lib in (la, lb):
    out = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ)
    env["VORS_HIP_LIB"] = lib if os.path.isabs(lib) else os.path.join(ROOT, "visual-odometry-rs_amd", "vors_amd", lib)
    subprocess.run([sys.executable, "-c", DUMP.format(root=ROOT, rows=rows, cols=cols, L=L, n=n, mode=mode, arith=arith, out=out)], check=True, env=env)
    res.append(np.load(out))
a, b = res
same = (a["poses"].view(np.uint32) == b["poses"].view(np.uint32)).all(axis=1)
print(f"mode {mode} {arith} {cols}x{rows} L{L} {n} pairs: poses bit-identical for {int(same.sum())} of {n}; max |diff| {np.abs(a['poses'] - b['poses']).max():.3g}; "
      f"status equal {bool((a['status'] == b['status']).all())}; stats bytes equal {bool((a['raw'].view(np.uint8) == b['raw'].view(np.uint8)).all())}")
