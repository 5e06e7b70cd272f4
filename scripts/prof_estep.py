import sys, os, cProfile, pstats, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import mk
from pypmc_amd.density.mixture import create_gaussian_mixture
from pypmc_amd.mix_adapt.variational import GaussianInference
D, K, N = 20, 64, 1_250_000
mix = create_gaussian_mixture(*mk(K, D, 3))
np.random.seed(9)
x = mix.propose(N, device=True)
vb = GaussianInference(x, initial_guess=mix)
for _ in range(5): vb.E_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): vb.E_step()
torch.cuda.synchronize()
print("E_step %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50): vb.update()
torch.cuda.synchronize()
print("update (M+E) %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): vb.E_step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
