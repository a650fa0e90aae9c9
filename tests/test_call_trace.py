"""tools/call_trace.py + tools/replay_call_trace.cpp: a recorded sequence of C-ABI calls replayed from C++ must reproduce the recorded run's
ciphertext words (the tool that measures the unchanged per-call pattern of the LoLa networks through the boundary)."""
import os
import sys

import numpy as np
import pytest

from conftest import get_gpu, get_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_trace_library_builds():
    import ctypes
    import call_trace
    L = ctypes.CDLL(call_trace.build())
    assert hasattr(L, "ct_replay")


@pytest.mark.gpu
def test_recorded_calls_replay_to_the_same_words(rng):
    import call_trace
    o, g = get_oracle("tiny"), get_gpu("tiny")
    cts = np.stack([o.encrypt(o.encode(rng.integers(0, 30, size=o.n, dtype=np.uint64))) for _ in range(4)])
    src = g.ct_alloc(4)
    g.ct_upload(src, 0, cts)
    pt = g.pt_alloc(2)
    g.encode_batch(rng.integers(1, 9, size=(2, o.n), dtype=np.uint64), pt, 0)

    def program():
        a, b, c = g.ct_alloc(1), g.ct_alloc(2), g.ct_alloc(1)
        g.add(src, 0, src, 1, a, 0)
        g.mul_plain(src, 2, pt, 0, b, 0, 2, pt_stride=1)
        g.rotate_rows(b, 0, -3, b, 0, 2)
        g.add_many(b, [0, 1], c, 0)
        g.mul_scalar(c, 0, [5], c, 0)
        g.sub(c, 0, a, 0, c, 0)
        m = g.pt_alloc(1)
        g.encode(np.arange(7, dtype=np.uint64), m, 0)
        g.add_plain(c, 0, m, 0, c, 0, subtract=True)
        g.mul_relin(c, 0, src, 3, a, 0)
        g.rotate_rows_add(a, 0, 2, c, 0, c, 0)
        g.rotate_columns(c, 0, c, 0)
        g.sum_slots(c, 0, 1, 8)
        out = g.ct_alloc(1)
        g.scalar_dot([a, 0, c], np.zeros(3, dtype=np.uint32), np.array([3, 1, g.t - 2], dtype=np.uint64), out, 0)
        g.free(a), g.free(b), g.free(m)
        return out, c                                        # c is left allocated: the replay must release it itself
    live0 = g.live_handles()
    rec = call_trace.Recorder(g).start()
    try:
        out, left = program()
    finally:
        rec.stop()
    want = g.ct_download(out, 0, 1)[0]
    rid = rec.ids[int(out)]
    g.free(out), g.free(left)
    assert g.live_handles() == live0
    for mode in (0, 1, 2):
        ms, handles = call_trace.replay([rec], 3, mode, [rid], warmup=1)
        assert ms > 0
        assert np.array_equal(g.ct_download(handles[0], 0, 1)[0], want), mode
        g.free(handles[0])
        assert g.live_handles() == live0                     # nothing the replays allocated survives
    g.free(src), g.free(pt)
