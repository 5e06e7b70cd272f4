#!/usr/bin/env python3
"""BASELINE.json's configurations 2-5 through the public front-end on ONE MI355X (one GPU's share
where the configuration is quoted on 8): wall time per call with the samples resident on the device.
Writes one JSON object (gpurun_out/configs.json when run on the GPU box).

    python scripts/configs_bench.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mk(K, D, seed, spread=3.0):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, spread, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    return mu, cov, w / w.sum()


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best.append(time.perf_counter() - t0)
    return float(np.median(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import pypmc_amd as pypmc
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture
    from pypmc_amd.sampler.importance_sampling import ImportanceSampler
    from pypmc_amd.mix_adapt.variational import GaussianInference
    from pypmc_amd.mix_adapt.pmc import gaussian_pmc
    from pypmc_amd.tools.convergence import perp_from_sums
    out = {"device": torch.cuda.get_device_name(0)}

    # -- config 2: MixtureDensity.multi_evaluate, D=20, K=16 Gaussian, N=1e6
    D, K, N = 20, 16, 1_000_000
    mix = create_gaussian_mixture(*mk(K, D, 1))
    np.random.seed(7)
    x = mix.propose(N, device=True)
    xh = x.cpu().numpy()
    be = pypmc.backend.get_backend()
    from pypmc_amd.density.mixture import component_set
    cs = component_set(mix.components, mix.weights)
    t_dev = timed(lambda: be.logpdf(x, cs), args.reps)
    t_host = timed(lambda: mix.multi_evaluate(xh), args.reps)
    out["cfg2_multi_evaluate_D20_K16_N1e6"] = {
        "device_resident_s": t_dev, "evals_per_s": N / t_dev,
        "numpy_in_numpy_out_s": t_host, "numpy_evals_per_s": N / t_host}

    # -- config 3: Student-t mixture (nu=8), D=30, K=32, N=1e7: importance weights + perplexity against the
    #    SURVEY target (a K_t=4 Gaussian mixture, mk(4, D, 11)) -- proposal and target families differ: one pass
    D, K, N = 30, 32, 10_000_000
    mu, cov, w = mk(K, D, 2)
    prop = create_t_mixture(mu, cov, np.full(K, 8.), w)
    target = create_gaussian_mixture(*mk(4, D, 11))
    np.random.seed(8)
    sampler = ImportanceSampler(target.evaluate, prop)
    res = {}

    def cfg3():
        res["r"] = sampler.run_device(N)
    t = timed(cfg3, max(2, args.reps // 2))
    x3 = res["r"]["samples"]
    pcs, tcs = component_set(prop.components, prop.weights), component_set(target.components, target.weights)
    t_w = timed(lambda: be.importance_weights(x3, pcs, tcs), args.reps)
    S, L, Q = res["r"]["weight_sums"]
    flops = N * (36 * (D * D + 4 * D) + 32 * 80 + 4 * 40)     # c_tr = 40, +40 for Student-t's log
    out["cfg3_student_t_IS_D30_K32_Kt4_N1e7"] = {
        "propose_plus_weights_s": t, "samples_per_s": N / t, "weights_only_s": t_w,
        "weights_only_samples_per_s": N / t_w, "weights_only_tflops": flops / t_w * 1e-12,
        "fraction_of_fp64_bound_2.0e9_per_s": N / t_w / (78.6e12 / (flops / N)),
        "perplexity": perp_from_sums(S, L, N)}
    # the same with a target close to the proposal (healthy weights; K_t = 32)
    target2 = create_gaussian_mixture(mu + 0.05, cov, w)
    tcs2 = component_set(target2.components, target2.weights)
    t_w2 = timed(lambda: be.importance_weights(x3, pcs, tcs2), args.reps)
    out["cfg3_student_t_IS_D30_K32_Kt32_N1e7"] = {"weights_only_s": t_w2, "weights_only_samples_per_s": N / t_w2}
    del sampler, res, x3

    # -- config 4: GaussianInference E-step, D=20, K=64; N=1e7 on one GPU and one GPU's share of 8
    D, K = 20, 64
    mu, cov, w = mk(K, D, 3)
    mix = create_gaussian_mixture(mu, cov, w)
    for label, N in (("N1e7_one_gpu", 10_000_000), ("N1.25e6_share_of_8", 1_250_000)):
        np.random.seed(9)
        x = mix.propose(N, device=True)
        vb = GaussianInference(x, initial_guess=mix)
        t = timed(vb.E_step, args.reps)
        out["cfg4_vb_estep_D20_K64_" + label] = {"estep_s": t, "samples_per_s": N / t}
        del vb, x

    # -- config 5: PMC adapt loop, D=40, K=128, N=1e8 over 8 GPUs -> one GPU's share: 1.25e7 samples per iteration
    D, K, KT, N = 40, 128, 4, 12_500_000
    tmu, tcov, tw = mk(KT, D, 11, spread=1.0)
    target = create_gaussian_mixture(tmu, tcov, tw)
    rs = np.random.RandomState(5)
    which = np.arange(K) % KT
    proposal = create_gaussian_mixture(tmu[which] + rs.normal(0, 0.15, (K, D)), 1.5 * tcov[which])
    np.random.seed(100)
    sampler = ImportanceSampler(target.evaluate, proposal)
    info = {}

    def iteration():
        t0 = time.perf_counter()
        run = sampler.run_device(N, trace_sort=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"], mincount=0, rb=True,
                     copy=False)
        torch.cuda.synchronize()
        info["propose_weight_s"], info["update_s"] = t1 - t0, time.perf_counter() - t1
        info["perplexity"] = perp_from_sums(run["weight_sums"][0], run["weight_sums"][1], N)
    t = timed(iteration, 3)
    out["cfg5_pmc_loop_D40_K128_N1.25e7_per_iteration"] = dict(iteration_s=t, samples_per_s=N / t, **info)

    # the same loop with the update reusing the component log-densities the weighting pass kept
    def iteration_reuse():
        t0 = time.perf_counter()
        run = sampler.run_device(N, trace_sort=True, keep_mahalanobis=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gaussian_pmc(run["samples"], sampler.proposal, run["weights"], run["origin"], mincount=0, rb=True,
                     copy=False, mahalanobis=run["mahalanobis"])
        torch.cuda.synchronize()
        info["propose_weight_s"], info["update_s"] = t1 - t0, time.perf_counter() - t1
        info["perplexity"] = perp_from_sums(run["weight_sums"][0], run["weight_sums"][1], N)
    t = timed(iteration_reuse, 3)
    out["cfg5_pmc_loop_D40_K128_N1.25e7_per_iteration_reusing_mahalanobis"] = \
        dict(iteration_s=t, samples_per_s=N / t, **info)

    text = json.dumps(out, indent=1)
    print(text)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    open(os.path.join(dst, "configs.json"), "w").write(text)


if __name__ == "__main__":
    main()
