import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_vb_state import _data, _fit
K, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 20)
x = _data(20000, D, 8, 1)
vb = _fit(x, K, True)
for _ in range(200):
    vb.M_step()
vb._state.get("alpha")
