import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    from far3d_amd import lib
    return lib.load()


def assert_detections_match(got, want, tag="", score_tol=1e-3, box_tol=2e-2):
    """Top-k decode comparison that is robust to the ranking of (near-)tied scores (decided by 1e-7-level summation-order
    noise): every reference detection clearing the k-th score by more than the logit tolerance must be present with the
    same label / score / box, and nothing else may appear above that bar.  got/want: (labels, boxes, scores) numpy arrays."""
    import numpy as np
    gl, gb, gs = got
    wl, wb, ws = want
    assert gl.shape == wl.shape, (tag, gl.shape, wl.shape)
    bar = ws.min() + score_tol
    used = np.zeros(len(gl), bool)
    for i in np.nonzero(ws > bar)[0]:
        cand = np.nonzero((~used) & (gl == wl[i]) & (np.abs(gs - ws[i]) < score_tol) & (np.abs(gb - wb[i]).max(axis=1) < box_tol))[0]
        assert len(cand) > 0, "%s: reference detection %d (label %d score %.4f) missing" % (tag, i, wl[i], ws[i])
        used[cand[0]] = True
    assert (gs[~used] <= bar + score_tol).all(), "%s: unexpected detections above the top-k bar" % tag
