"""The oracle's batched sequence entry (vo_track_sequences: one Tracker per sequence, frame-major storage, threads over sequences) is the
per-frame Tracker loop of src/bin/vors_track.rs:46-62 and nothing else. CPU only."""
import numpy as np

from oracle import oracle as O


def test_track_sequences_equals_the_tracker_loop():
    rows, cols, L, F, n = 60, 80, 3, 6, 3
    intr = O.scaled_intrinsics(rows, cols)
    step = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    g = np.zeros((F, n, rows, cols), np.uint8)
    d = np.zeros((F, n, rows, cols), np.uint16)
    for k in range(F):
        for s in range(n):
            g[k, s], d[k, s] = O.synth_frame(100 + s, step * k * (1 + s), rows, cols, intr, frame_salt=k)
    cfg = O.make_config(L, intr)
    r = O.track_sequences(cfg, g, d, n_threads=2)
    for s in range(n):
        t = O.Tracker(cfg, 0.0, d[0, s], 0.0, g[0, s])
        for k in range(1, F):
            assert t.track(float(k), d[k, s], float(k), g[k, s]) == r["status"][s, k - 1]
            assert (t.current_frame()[1] == r["poses"][s, k - 1]).all()
            assert t.last()["changed_keyframe"] == bool(r["changed_keyframe"][s, k - 1])
    assert r["changed_keyframe"].sum() >= 1
    assert (O.track_sequences(cfg, g, d, n_threads=1)["poses"] == r["poses"]).all()  # threads only partition the sequences
