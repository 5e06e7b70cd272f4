#!/usr/bin/env python3
"""Where the time of MixtureDensity.multi_evaluate(host array) goes at a small batch (K = 32, D = 20, N = 1e4: 1.6 MB in, 80 KB out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pypmc_amd.backend import get_backend
from pypmc_amd.density.mixture import create_gaussian_mixture, component_set
from test_gpu_kernels import mk
be = get_backend()
def us(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for K, D, N in ((32, 20, 10000), (32, 20, 1000), (128, 40, 4096)):
    mix = create_gaussian_mixture(*mk(K, D, 1)); np.random.seed(1); x = mix.propose(N)
    cs = component_set(mix.components, mix.weights); xd = be.asdevice(x); out = be.empty(N)
    pin = torch.empty((N, D), dtype=torch.float64).pin_memory(); pin.numpy()[:] = x
    xd2 = torch.empty((N, D), dtype=torch.float64, device=be.device)
    print("K=%d D=%d N=%d" % (K, D, N))
    print("  multi_evaluate (host in, host out)      %7.1f us" % us(lambda: mix.multi_evaluate(x)))
    print("  component_set lookup                    %7.1f us" % us(lambda: component_set(mix.components, mix.weights)))
    print("  asdevice(x)  [from_numpy().to(device)]  %7.1f us" % us(lambda: be.asdevice(x)))
    print("  pinned copy: memcpy into pinned + H2D   %7.1f us" % us(lambda: (pin.numpy().__setitem__(slice(None), x), xd2.copy_(pin, non_blocking=True), torch.cuda.current_stream().synchronize())))
    print("  H2D from pinned only                    %7.1f us" % us(lambda: (xd2.copy_(pin, non_blocking=True), torch.cuda.current_stream().synchronize())))
    print("  be.logpdf(device x), no sync            %7.1f us" % us(lambda: be.logpdf(xd, cs)))
    print("  be.logpdf(device x) + tohost(out)       %7.1f us" % us(lambda: be.tohost(be.logpdf(xd, cs)["out"])))
    print("  tohost(out) alone                       %7.1f us" % us(lambda: be.tohost(out)))
