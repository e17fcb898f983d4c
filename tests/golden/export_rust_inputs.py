"""Writes the RAW INPUTS of tests/golden/*.npz as flat little-endian .bin files + a line-based manifest under tests/golden/rust_inputs/<case>/,
for tools/rust_golden/dump_golden.rs — the program a maintainer with `cargo` runs inside the reference crate to produce
tests/golden/rust/<case>.json, which tests/test_golden_rust.py then compares with the oracle BIT FOR BIT.
Everything here is data: images, depth maps, the configuration, and the per-level observation lists (coordinates, inverse depths, Jacobians
of pair 0) the operator-level part of the dump feeds to LMOptimizerState::iterative_solve.     Run from the repo root:
    python tests/golden/export_rust_inputs.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def f32_hex(x):
    return f"{np.float32(x).view(np.uint32):08x}"


def export(name):
    d = np.load(os.path.join(HERE, name + ".npz"))
    out = os.path.join(HERE, "rust_inputs", name)
    os.makedirs(out, exist_ok=True)
    rows, cols, L, mode = int(d["rows"]), int(d["cols"]), int(d["L"]), int(d["mode"])
    n = d["kf_gray"].shape[0]
    d["kf_gray"].astype(np.uint8).tofile(os.path.join(out, "kf_gray.bin"))       # [pairs][rows][cols] row-major
    d["cur_gray"].astype(np.uint8).tofile(os.path.join(out, "cur_gray.bin"))
    d["kf_depth"].astype("<u2").tofile(os.path.join(out, "kf_depth.bin"))
    d["cur_depth"].astype("<u2").tofile(os.path.join(out, "cur_depth.bin"))
    lines = [f"case {name}", f"rows {rows}", f"cols {cols}", f"levels {L}", f"mode {mode}", f"thresh {int(d['thresh'])}", f"pairs {n}",
             "depth_scale_f32 " + f32_hex(5000.0), "idepth_variance_f32 " + f32_hex(1e-4),
             "intrinsics_f32 " + " ".join(f32_hex(v) for v in d["intr"])]   # cu cv fu fv skew
    for l in range(L):
        xy, iz, jac = d[f"xy{l}"], d[f"iz{l}"], d[f"jac{l}"]
        xy.astype("<i4").tofile(os.path.join(out, f"xy{l}.bin"))               # [n][2] = (x, y)
        iz.astype("<f4").tofile(os.path.join(out, f"iz{l}.bin"))
        jac.astype("<f4").tofile(os.path.join(out, f"jac{l}.bin"))             # [n][6]
        lines.append(f"level {l} n {len(iz)}")
    open(os.path.join(out, "manifest.txt"), "w").write("\n".join(lines) + "\n")
    print(name, "->", out, sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)), "bytes")


if __name__ == "__main__":
    for name in ("sparse_128x96_L4", "sparse_odd_167x123_L3", "dense_80x60_L3"):
        export(name)
