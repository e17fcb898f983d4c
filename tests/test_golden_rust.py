"""Pins the ORACLE to the Rust reference, bit for bit — when tests/golden/rust/<case>.json exist.

Those documents are what tools/rust_golden/dump_golden.rs prints when it runs, INSIDE THE REFERENCE CRATE, on the raw inputs of the golden
fixtures (tests/golden/rust_inputs/, written by tests/golden/export_rust_inputs.py): the reference's own Tracker, candidate selection,
inverse-depth pyramid and LM loop. No Rust toolchain exists in the image this repository is built in, so the documents cannot be produced
here; anyone with `cargo` can (one command per case), and this test then turns `parity unpinned` into a
checked fact. Without the documents the comparison is SKIPPED — but the comparator itself is exercised on the document the oracle
predicts, and the exported inputs are checked against the fixtures, so the recipe cannot rot. CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
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


@pytest.mark.parametrize("case", CASES)
def test_the_oracle_equals_the_rust_reference_bit_for_bit(case):
    path = os.path.join(GOLDEN, "rust", case + ".json")
    if not os.path.exists(path):
        pytest.skip(f"{path} not present: produce it with tools/rust_golden/dump_golden.rs inside the reference crate (INTEGRATION.md §5)")
    doc = json.load(open(path))
    diffs = compare(doc, expected_document(case))
    assert not diffs, "\n".join(diffs)


@pytest.mark.parametrize("case", CASES)
def test_the_comparator_accepts_the_oracles_own_document_and_sees_one_flipped_bit(case):
    exp = expected_document(case)
    doc = json.loads(json.dumps(exp))  # what a faithful dump would parse to
    assert compare(doc, exp) == []
    lvl = doc["lm"][-1]
    lvl["model"][0] = f"{int(lvl['model'][0], 16) ^ 1:08x}"
    assert any("model" in s for s in compare(doc, exp))
    assert len(exp["lm"]) == int(np.load(os.path.join(GOLDEN, case + ".npz"))["L"])
    # the per-level chain of lm_solve calls on the stored lists ends where the stored tracker run ended
    d = np.load(os.path.join(GOLDEN, case + ".npz"))
    assert exp["lm"][-1]["model"] == f32_hex(d["models"][0])
    assert [e["nb_iter"] for e in exp["lm"]] == [int(v) for v in d["nb_iter"][0][::-1]]


@pytest.mark.parametrize("case", CASES)
def test_the_exported_inputs_are_the_fixtures(case):
    d = np.load(os.path.join(GOLDEN, case + ".npz"))
    root = os.path.join(GOLDEN, "rust_inputs", case)
    man = dict(line.split(None, 1) for line in open(os.path.join(root, "manifest.txt")).read().splitlines() if not line.startswith("level "))
    assert int(man["rows"]) == int(d["rows"]) and int(man["cols"]) == int(d["cols"]) and int(man["levels"]) == int(d["L"])
    assert man["intrinsics_f32"].split() == f32_hex(d["intr"])
    assert (np.fromfile(os.path.join(root, "kf_gray.bin"), np.uint8) == d["kf_gray"].ravel()).all()
    assert (np.fromfile(os.path.join(root, "cur_gray.bin"), np.uint8) == d["cur_gray"].ravel()).all()
    assert (np.fromfile(os.path.join(root, "kf_depth.bin"), "<u2") == d["kf_depth"].ravel()).all()
    for l in range(int(d["L"])):
        assert (np.fromfile(os.path.join(root, f"xy{l}.bin"), "<i4") == d[f"xy{l}"].ravel()).all()
        assert (np.fromfile(os.path.join(root, f"jac{l}.bin"), "<f4").view(np.uint32) == d[f"jac{l}"].astype("<f4").ravel().view(np.uint32)).all()
