#!/usr/bin/env python3
"""Development aid: field by field, the device-resident VB update against the host path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_vb_state import _data, _fit, FIELDS
K, D, N = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 1, 500))]
x = _data(N, D, K, 11 * K + D)
dev, host = _fit(x, K, True), _fit(x, K, False)
for it in range(3):
    print("bound", dev.likelihood_bound(), host.likelihood_bound())
    for a in ("_expectation_log_p_X", "_expectation_log_p_Z", "_expectation_log_p_pi", "_expectation_log_p_mu_lambda",
              "_expectation_log_q_Z", "_expectation_log_q_pi", "_expectation_log_q_mu_lambda"):
        print("   ", a, getattr(dev, a), getattr(host, a))
    for name in FIELDS:
        a, b = np.asarray(dev._peek(name)), np.asarray(host._peek(name))
        print("  %-28s max rel %.3e" % (name, np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)))
    dev.update(); host.update()
