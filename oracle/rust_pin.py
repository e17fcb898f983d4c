"""Test infrastructure (like everything under oracle/): the comparator that pins the ORACLE to the Rust reference, bit for bit.

tests/golden/rust/<case>.json are what tools/rust_golden/dump_golden.rs prints when it runs inside the reference crate on the raw inputs of
the golden fixtures (tests/golden/rust_inputs/). No Rust toolchain exists in the image this repository is built in, so the documents cannot
be produced here (INTEGRATION.md §5 is the one-command recipe). `parity_pinned()` is what bench.py's line and __graft_entry__.smoke() report:
True iff every case's document exists AND equals what the oracle predicts — until then every "bit-identical" in this repository means
"to the oracle", not "to vors". Used by tests/test_golden_rust.py, bench.py (after the timed region) and smoke() only."""
import json
import os

import numpy as np

from . import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["sparse_128x96_L4", "sparse_odd_167x123_L3", "dense_80x60_L3"]
FNV_OFFSET, FNV_PRIME, MASK64 = 0xcbf29ce484222325, 0x100000001b3, (1 << 64) - 1


def fnv1a(data, h=FNV_OFFSET):
    for b in bytes(data):
        h = ((h ^ b) * FNV_PRIME) & MASK64
    return h


def f32_hex(a):
    return [f"{v:08x}" for v in np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()]


def expected_document(case):
    """The document dump_golden.rs must print for `case` if the oracle equals the reference — same keys, same encodings."""
    d = np.load(os.path.join(GOLDEN, case + ".npz"))
    rows, cols, L, mode = int(d["rows"]), int(d["cols"]), int(d["L"]), int(d["mode"])
    doc = {"case": case}
    if mode == 0:
        doc["poses"] = [f32_hex(p) for p in d["poses"]]
        doc["mask0"] = "".join("1" if v else "0" for v in d["mask0"].ravel())
        doc["idepth"] = [{"n": int(len(d[f"iz{l}"])), "fnv": f"{fnv1a(np.ascontiguousarray(d[f'iz{l}'], '<f4').tobytes()):016x}"} for l in range(L)]
    pyr = O.mean_pyramid(d["kf_gray"][0], L)
    cur = O.mean_pyramid(d["cur_gray"][0], L)
    doc["pyramid"] = [f"{fnv1a(np.ascontiguousarray(img).tobytes()):016x}" for img in pyr]
    lm, model = [], np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    for l in range(L - 1, -1, -1):
        st, out, it, e, lam = O.lm_solve(d[f"k{l}"], pyr[l], cur[l], d[f"xy{l}"], d[f"iz{l}"], d[f"jac{l}"], model)
        if st != 0:
            lm.append({"level": l, "error": "Error at Cholesky decomposition of hessian"})
            break
        model = out
        lm.append({"level": l, "nb_iter": int(it), "model": f32_hex(out), "energy": f32_hex([e])[0], "lm_coef": f32_hex([lam])[0]})
    doc["lm"] = lm
    return doc


def compare(doc, exp):
    """-> list of human-readable differences (empty = bit-identical)."""
    diffs = []
    for key in ("poses", "mask0", "idepth", "pyramid"):
        if key in exp:
            if key not in doc:
                diffs.append(f"{key}: missing")
            elif doc[key] != exp[key]:
                if key == "poses":
                    bad = [i for i, (a, b) in enumerate(zip(doc[key], exp[key])) if a != b]
                    diffs.append(f"poses: pairs {bad} differ (first: rust {doc[key][bad[0]]} oracle {exp[key][bad[0]]})" if bad else "poses: length")
                elif key == "mask0":
                    n = sum(a != b for a, b in zip(doc[key], exp[key])) if len(doc[key]) == len(exp[key]) else -1
                    diffs.append(f"mask0: {n} pixels differ")
                else:
                    diffs.append(f"{key}: rust {doc[key]} oracle {exp[key]}")
    if len(doc.get("lm", [])) != len(exp["lm"]):
        diffs.append(f"lm: {len(doc.get('lm', []))} levels vs {len(exp['lm'])}")
    for a, b in zip(doc.get("lm", []), exp["lm"]):
        for k in b:
            if a.get(k) != b[k]:
                diffs.append(f"lm level {b['level']} {k}: rust {a.get(k)} oracle {b[k]}")
    return diffs




def parity_pinned():
    """-> (pinned: bool, detail: str). pinned = every tests/golden/rust/<case>.json exists and compares equal to the oracle's prediction."""
    missing, differing = [], []
    for case in CASES:
        path = os.path.join(GOLDEN, "rust", case + ".json")
        if not os.path.exists(path):
            missing.append(case)
            continue
        try:
            diffs = compare(json.load(open(path)), expected_document(case))
        except Exception as e:  # a malformed document pins nothing
            diffs = [f"unreadable: {e}"]
        if diffs:
            differing.append(f"{case}: {diffs[0]}")
    if differing:
        return False, "Rust documents DIFFER from the oracle: " + "; ".join(differing)
    if missing:
        return False, (f"tests/golden/rust/ lacks {len(missing)} of {len(CASES)} documents (no cargo / rustc in the build image): the oracle is a restatement, "
                       "unpinned by the Rust reference; recipe: INTEGRATION.md §5 (tools/rust_golden/dump_golden.rs)")
    return True, f"all {len(CASES)} documents produced by the Rust reference equal the oracle bit for bit"
