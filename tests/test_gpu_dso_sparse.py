"""The sparse form of the DSO / generic-mask keyframe path (dso_kernels.hip: Morton-sorted pick list, segmented fusion per level) against
the plane path it replaces (per-level inverse-depth planes, count / compact / records), which VORS_DSO_PLANES=1 still selects: the usable
candidates of every level must be the same set with bit-identical inverse depths, templates and gradients. The environment switch is
read once per process, so each path runs in its own interpreter. GPU only.

Shapes: the bench shape; an odd shape (trailing row / column without parents); a small image where the selector's recursion ends with a
mask far denser than its target (the pick list does not fit the pair's scratch in one go: groups of bands)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = r"""
import sys, os
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L, n, out = {rows}, {cols}, {L}, 3, {out!r}
intr = V.scaled_intrinsics(rows, cols)
kg, kd, cg, _, gt = V.synth_render_pairs((1 << 63) | 0x5EEDD500, n, rows, cols, intr)
b = V.Batch(V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=2), n, rows, cols)
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
b.track_pairs(kg, kd, cg, poses, status); torch.cuda.synchronize()
res = dict(poses=poses.cpu().numpy(), status=status.cpu().numpy())
for p in range(n):
    for l in range(L):
        xy, iz, jac, tm = b.points(p, l)
        o = np.lexsort((xy[:, 0], xy[:, 1]))
        res[f"xy_{{p}}_{{l}}"] = xy[o]; res[f"iz_{{p}}_{{l}}"] = iz[o].view(np.uint32); res[f"jac_{{p}}_{{l}}"] = jac[o].view(np.uint32); res[f"tm_{{p}}_{{l}}"] = tm[o]
np.savez(out, **res)
"""


def run(tmp_path, tag, planes, rows, cols, L, scan=False, threads=None, sort=None, extra_env=None):
    out = str(tmp_path / f"{tag}.npz")
    env = dict(os.environ)
    for k in ("VORS_DSO_ROUNDS_LDS", "VORS_DSO_STAMPS", "VORS_DSO_FIRST_MAXIMA"):
        env.pop(k, None)
    env.update(extra_env or {})
    env["VORS_DSO_PLANES"] = "1" if planes else "0"
    env["VORS_DSO_SCAN"] = "1" if scan else "0"
    env.pop("VORS_DSO_SORT", None)
    if sort:
        env["VORS_DSO_SORT"] = sort
    if threads:
        env["VORS_DSO_ROUNDS_THREADS"], env["VORS_DSO_RECORDS_THREADS"] = str(threads[0]), str(threads[1])
    subprocess.run([sys.executable, "-c", DUMP.format(root=ROOT, rows=rows, cols=cols, L=L, out=out)], check=True, env=env, timeout=300)
    return np.load(out)


@pytest.mark.parametrize("rows,cols,L", [(480, 640, 6), (121, 163, 4), (96, 128, 3), (64, 96, 2)])
def test_sparse_form_equals_plane_path(tmp_path, rows, cols, L):
    a, b = run(tmp_path, "sparse", False, rows, cols, L), run(tmp_path, "planes", True, rows, cols, L)
    assert (a["status"] == b["status"]).all()
    n_lvl0 = []
    for key in a.files:
        if key in ("poses", "status"):
            continue
        assert a[key].shape == b[key].shape and (a[key] == b[key]).all(), key
        if key.startswith("xy_") and key.endswith("_0"):
            n_lvl0.append(len(a[key]))
    print(f"[{cols}x{rows} L{L}] level-0 candidates per pair {n_lvl0}; max pose difference between the two list orders "
          f"{np.abs(a['poses'] - b['poses']).max():.2e}")
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-4  # same candidates, different order of summation


@pytest.mark.parametrize("rows,cols,L", [(480, 640, 6), (121, 163, 4), (96, 128, 3), (64, 96, 2)])
def test_pick_list_of_the_selection_rounds_equals_the_scan_of_the_stamp_plane(tmp_path, rows, cols, L):
    """Round 3: dso_rounds_kernel hands the picks of its final round over as a list (no pass over the 1-byte-per-pixel stamp plane);
    VORS_DSO_SCAN=1 still extracts them with mask_sparse_scan_kernel. Both feed the same sort: identical lists, hence identical POSES
    bit for bit (incl. the shape whose list overflows and goes through in groups of bands)."""
    a, b = run(tmp_path, "list", False, rows, cols, L), run(tmp_path, "scan", False, rows, cols, L, scan=True)
    for key in a.files:
        assert a[key].shape == b[key].shape and (a[key].view(np.uint8) == b[key].view(np.uint8)).all(), key


@pytest.mark.parametrize("rows,cols,L", [(480, 640, 6), (121, 163, 4), (64, 96, 2)])
def test_threads_per_pair_of_the_selector_and_records_kernels_do_not_change_the_lists(tmp_path, rows, cols, L):
    """Large batches run the selection rounds and the records kernel with 512 threads per pair, small ones with 1024 (scheduling only):
    same candidates, same values, same order, hence the same poses bit for bit."""
    a = run(tmp_path, "t1024", False, rows, cols, L, threads=(1024, 1024))
    for tag, threads in (("t512", (512, 512)), ("t768", (768, 512))):  # (768: round 4's choice for the selection rounds from 512 pairs on)
        b = run(tmp_path, tag, False, rows, cols, L, threads=threads)
        for key in a.files:
            assert a[key].shape == b[key].shape and (a[key].view(np.uint8) == b[key].view(np.uint8)).all(), (tag, key)


@pytest.mark.parametrize("rows,cols,L,threads", [(480, 640, 6, (512, 512)), (480, 640, 6, (1024, 1024)), (121, 163, 4, None), (960, 1280, 7, None), (64, 96, 2, None)])
def test_bucket_sort_of_the_pick_list_equals_the_bitonic_network(tmp_path, rows, cols, L, threads):
    """Round 4: the records kernel orders a pair's picks (distinct Morton codes) with a two-step bucket sort in LDS — a counter per image tile,
    then an element's place among its few tile mates — instead of round 3's 66-stage bitonic network (VORS_DSO_SORT=bitonic keeps it).
    Same order, hence the same lists, values and poses bit for bit; 1280x960 has 22-bit codes (64x64-pixel buckets), 64x96 overflows into
    groups of bands."""
    a = run(tmp_path, "bucket", False, rows, cols, L, threads=threads)
    b = run(tmp_path, "bitonic", False, rows, cols, L, threads=threads, sort="bitonic")
    for key in a.files:
        assert a[key].shape == b[key].shape and (a[key].view(np.uint8) == b[key].view(np.uint8)).all(), key


@pytest.mark.parametrize("rows,cols,L", [(480, 640, 6), (96, 128, 3), (64, 96, 2), (960, 1280, 7)])
def test_selection_round_forms_give_the_same_lists(tmp_path, rows, cols, L):
    """Round 5: the first selection round runs on the first pass's 4 x 4 block maxima with the two upper block levels in LDS, and in the list
    form it writes its pick stamps only when something will read them (a pick list or a sort buffer that overflows: 64 x 96 goes through
    in groups of bands, from the stamps). Against the stamps always written (VORS_DSO_STAMPS=1), the generic rounds on the planes in global
    memory (VORS_DSO_ROUNDS_LDS=0; 1280 x 960 takes them anyway: its upper levels do not fit) and the block maxima recomputed by the rounds
    kernel (VORS_DSO_FIRST_MAXIMA=0): identical lists, values and poses, bit for bit."""
    a = run(tmp_path, "default", False, rows, cols, L)
    for tag, env in (("stamps", {"VORS_DSO_STAMPS": "1"}), ("global", {"VORS_DSO_ROUNDS_LDS": "0"}), ("nomax", {"VORS_DSO_FIRST_MAXIMA": "0"})):
        b = run(tmp_path, tag, False, rows, cols, L, extra_env=env)
        for key in a.files:
            assert a[key].shape == b[key].shape and (a[key].view(np.uint8) == b[key].view(np.uint8)).all(), (tag, key)

