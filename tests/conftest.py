import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure): compiled on demand with gcc."""
    from oracle import oracle as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def f32_dist_sq(points, q):
    """metric.rs:21-26 in numpy float32 with the reference's op order: ((0+dx^2)+dy^2)+dz^2."""
    p = np.asarray(points, np.float32)
    q = np.asarray(q, np.float32)
    dx = p[..., 0] - q[..., 0]
    dy = p[..., 1] - q[..., 1]
    dz = p[..., 2] - q[..., 2]
    s = np.float32(0.0) + dx * dx
    s = s + dy * dy
    s = s + dz * dz
    return s.astype(np.float32)


def check_knn_contract(points, queries, got_entries, got_counts, ref_entries, ref_counts):
    """SURVEY.md H2 contract between an implementation and the reference restatement:
    counts identical, distances bit-identical, and every index either identical or a
    legitimate member of the same distance tie group (its recomputed distance is bit-equal
    to the reference entry's), with no duplicate indices in a row."""
    points = np.asarray(points, np.float32)
    queries = np.asarray(queries, np.float32).reshape(-1, 3)
    assert np.array_equal(np.asarray(got_counts, np.int64), np.asarray(ref_counts, np.int64)), "counts differ"
    k = got_entries.shape[1]
    valid = np.arange(k)[None, :] < np.asarray(ref_counts)[:, None]
    gd = got_entries["distance"].view(np.uint32)
    rd = ref_entries["distance"].view(np.uint32)
    assert np.array_equal(gd[valid], rd[valid]), "distances are not bit-identical"
    gi = got_entries["index"].astype(np.int64)
    ri = ref_entries["index"].astype(np.int64)
    diff = valid & (gi != ri)
    rows, cols = np.nonzero(diff)
    if len(rows):
        d_got = f32_dist_sq(points[gi[rows, cols]], queries[rows]).view(np.uint32)
        assert np.array_equal(d_got, rd[rows, cols]), "an index outside its distance tie group was returned"
    for r in np.unique(rows):
        row = gi[r][valid[r]]
        assert len(np.unique(row)) == len(row), "duplicate index in a result row"
    return int(diff.sum())
