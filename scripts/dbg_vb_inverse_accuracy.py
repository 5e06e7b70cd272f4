#!/usr/bin/env python3
"""Development aid: how far the device M-step's W (Cholesky + inverse of the factor + Gram matrix in fp64, one wavefront per
component) and the host's (LAPACK potrf / potri) are from the inverse taken in extended precision, on the golden VB cases."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import frontend_cases as fc
from pypmc_amd.backend import get_backend
from pypmc_amd.mix_adapt.variational import GaussianInference
be = get_backend(None)


def inv_ld(a):
    a = a.astype(np.longdouble)
    D = len(a)
    L = np.zeros((D, D), dtype=np.longdouble)
    for j in range(D):
        s = a[j, j] - (L[j, :j] ** 2).sum()
        L[j, j] = np.sqrt(s)
        for i in range(j + 1, D):
            L[i, j] = (a[i, j] - (L[i, :j] * L[j, :j]).sum()) / L[j, j]
    X = np.zeros((D, D), dtype=np.longdouble)
    for j in range(D):
        X[j, j] = 1 / L[j, j]
        for i in range(j + 1, D):
            X[i, j] = -(L[i, j:i] * X[j:i, j]).sum() / L[i, i]
    return X.T @ X


for tag, weighted in (("d2k3", False), ("d5k4w", True), ("d20k8", False), ("d3k5first", True)):
    g = fc.load_golden("vb_" + tag)
    out = {}
    for device in (True, False):
        GaussianInference.device_update = device
        vb = fc._vb_from_golden(g, be, weighted)
        vb.M_step()
        out[device] = np.array(vb.W)
        if not device:
            dx = vb.x_mean_comp - vb.m0
            inv_w = np.einsum('ki,kj->kij', dx, dx) * (vb.beta0 / (vb.beta0 + vb.N_comp))[:, None, None]
            inv_w += vb.S
            inv_w *= vb.N_comp[:, None, None]
            inv_w += vb.inv_W0
    GaussianInference.device_update = True
    true = np.array([inv_ld(m) for m in inv_w])
    n = lambda a: float(np.abs(a).max(axis=(1, 2)).max())
    rel = lambda a, b: float((np.abs(a - b).max(axis=(1, 2)) / np.abs(b).max(axis=(1, 2)).astype(float)).max())
    print("%-10s cond %.2e   device vs true %.2e   host vs true %.2e   device vs host %.2e   golden W vs true %.2e" % (
        tag, max(np.linalg.cond(m) for m in inv_w), rel(out[True], true), rel(out[False], true), rel(out[True], out[False]),
        rel(g["u1_W"], true) if g["u1_W"].shape == true.shape else -1))
