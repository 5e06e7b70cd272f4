#!/usr/bin/env python3
"""Stress of the one-launch bound (the workgroup that draws the last ticket adds the terms up): the same state, the bound
many times over for several K -- every result must have the first one's bits."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_vb_state import _data, _fit
for K, D in ((3, 2), (64, 20), (200, 8), (1000, 3), (128, 40)):
    x = _data(max(4 * K, 2000), D, 4, K + D)
    vb = _fit(x, K, True)
    vb.update()
    st = vb._state
    first = st.step(None, bound=True)["bound"].copy()
    bad = 0
    for i in range(3000):
        b = st.step(None, bound=True)["bound"]
        bad += not np.array_equal(b, first)
    # interleaved with E-steps (other kernels between two bounds)
    for i in range(200):
        vb.E_step()
        b = st.step(None, bound=True)["bound"]
    print("K=%4d D=%2d: 3000 bounds, %d different from the first (%.15g)" % (K, D, bad, first[0]), flush=True)
    assert bad == 0
print("ok")
