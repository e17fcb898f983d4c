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

from oracle.rust_pin import CASES, GOLDEN, compare, expected_document, f32_hex, fnv1a, parity_pinned  # noqa: F401  (the comparator is shared with bench.py / smoke())


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


def test_parity_pinned_reports_the_absence_of_the_rust_documents_loudly():
    """bench.py's `parity_pinned` and smoke()'s last line: False with a reason unless every document exists and compares equal."""
    pinned, detail = parity_pinned()
    have = [c for c in CASES if os.path.exists(os.path.join(GOLDEN, "rust", c + ".json"))]
    if len(have) < len(CASES):
        assert pinned is False and "INTEGRATION.md" in detail
    else:
        assert isinstance(pinned, bool) and detail
