import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pypmc_amd.tools._linalg import chol_inv_det_batch
for K, D in ((128, 40), (64, 20), (32, 20)):
    rs = np.random.RandomState(0)
    A = rs.normal(size=(K, D, D)); m = A @ A.transpose(0, 2, 1) / D + 0.5 * np.eye(D)
    chol_inv_det_batch(m, check_symmetric=False)
    t0 = time.perf_counter()
    for _ in range(50): chol_inv_det_batch(m, check_symmetric=False)
    t = (time.perf_counter() - t0) / 50
    # copies of the same volume
    d = torch.empty(K * D * D * 2, dtype=torch.float64, device="cuda")
    h_in = torch.from_numpy(m.reshape(-1).copy()); h_out = torch.empty(K * D * D * 2, dtype=torch.float64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        d[:K * D * D].copy_(h_in); h_out.copy_(d); torch.cuda.synchronize()
    tc = (time.perf_counter() - t0) / 50
    print("K=%d D=%d chol_inv_det_batch %.3f ms; pageable up %d KB + down %d KB: %.3f ms" % (K, D, t * 1e3, K*D*D*8//1024, K*D*D*16//1024, tc * 1e3))
