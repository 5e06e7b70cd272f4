#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference implementation.

Run in the BUILD container only (it needs /root/reference):

    python tests/golden/make_golden.py [--ref-build /tmp/pypmc_ref]

The script copies /root/reference to a scratch directory, builds its Cython extensions
there (``python3 setup.py build_ext --inplace``), imports that build, and stores *data* --
seeded inputs and the outputs the reference computes for them -- as small ``.npz`` files.
No reference source or bytecode is stored.  The ``.npz`` files are committed; the parity
tests (CPU oracle and, on the GPU box, the HIP path) read only them.

Synthetic mixtures follow SURVEY.md section 8(d):  mu_k ~ N(0, 3^2 I),
Sigma_k = A A^T / D + 0.5 I with A_ij ~ N(0,1), weights ~ U(0.5, 1.5) normalised.
"""
import argparse
import os
import shutil
import subprocess
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def ensure_reference(build_dir):
    marker = os.path.join(build_dir, "pypmc", "tools")
    have = os.path.isdir(marker) and any(f.startswith("_linalg.") and f.endswith(".so")
                                         for f in os.listdir(marker))
    if not have:
        if os.path.exists(build_dir):
            shutil.rmtree(build_dir)
        shutil.copytree("/root/reference", build_dir)
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                              cwd=build_dir, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    sys.path.insert(0, build_dir)


def mk(K, D, seed, dof=None):
    rs = np.random.RandomState(seed)
    mu = rs.normal(0, 3, size=(K, D))
    cov = np.empty((K, D, D))
    for k in range(K):
        A = rs.normal(0, 1, size=(D, D))
        cov[k] = A.dot(A.T) / D + 0.5 * np.eye(D)
    w = rs.uniform(0.5, 1.5, size=K)
    w /= w.sum()
    return mu, cov, w


ONLY = None


def save(name, **arrays):
    if ONLY is not None and not name.startswith(ONLY):
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.))


def comp_params(mix, student=False):
    K = len(mix)
    D = mix.dim
    inv_sigma = np.array([c.inv_sigma for c in mix.components])
    log_det = np.array([c.log_det_sigma for c in mix.components])
    if student:
        log_norm = np.array([c._local_t.log_normalization for c in mix.components])
        dof = np.array([c.dof for c in mix.components])
    else:
        log_norm = np.array([c._local_gauss.log_normalization for c in mix.components])
        dof = np.zeros(0)
    mu = np.array([c.mu for c in mix.components]).reshape(K, D)
    sigma = np.array([c.sigma for c in mix.components]).reshape(K, D, D)
    return dict(weights=np.array(mix.weights), mu=mu, sigma=sigma, inv_sigma=inv_sigma,
                log_det_sigma=log_det, log_norm=log_norm, dof=dof)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-build", default="/tmp/pypmc_ref")
    ap.add_argument("--only", default=None, help="write only the files whose name starts with this (the others are left as they are)")
    args = ap.parse_args()
    global ONLY
    ONLY = args.only
    warnings.simplefilter("ignore")
    ensure_reference(args.ref_build)

    import logging
    logging.disable(logging.CRITICAL)
    import pypmc
    from pypmc.tools._linalg import bilinear_sym
    from pypmc.tools._regularize import logsumexp, logsumexp2D
    from pypmc.tools.convergence import perp, ess
    from pypmc.density.gauss import Gauss
    from pypmc.density.student_t import StudentT, LocalStudentT
    from pypmc.density.mixture import (MixtureDensity, create_gaussian_mixture,
                                       create_t_mixture)
    from pypmc.sampler.importance_sampling import ImportanceSampler, combine_weights
    from pypmc.mix_adapt.variational import GaussianInference, VBMerge
    from pypmc.mix_adapt.pmc import gaussian_pmc, student_t_pmc, PMC

    # ------------------------------------------------------------------ known-answer inputs
    # inputs are those of the reference's unit tests; `*_pinned` are the
    # literal expected values those tests assert, `*_ref` what the reference computes here.
    vec = np.array([2., 4.3, 7.])
    mat = np.array([[3., 5., 1.9], [5., .8, 2.2], [1.9, 2.2, 4.2]])
    lse_v = np.array([1., 2., 3.])
    lse_w = np.array([.3, .4, .3])
    lse2_v = np.array([[4., 8., 3.], [.3, .1, 5.], [2.3, 5.6, 2.3]])
    lse2_w = np.array([1.3, .4, .3])
    g_sigma = np.array([[0.01, 0.003], [0.003, 0.0025]])
    g_mean = np.array([4.3, 1.1])
    g_point = np.array([4.35, 1.2])
    g = Gauss(g_mean, g_sigma)
    t_mean = np.array([1.25, 4.3])
    t_sigma = np.array([[0.0049, 0.], [0., .01]])
    t = StudentT(t_mean, t_sigma, 5.)
    t_points = np.array([[1.3, 4.4], [1.26, 4.424]])
    cauchy = LocalStudentT(sigma=1, dof=1)
    pw = np.array([0., 1., 2., 3., 4.])
    save("kat",
         bil_matrix=mat, bil_vector=vec, bil_pinned=504.23200000000003,
         bil_ref=bilinear_sym(mat, vec),
         lse_values=lse_v, lse_weights=lse_w, lse_pinned=2.28205254, lse_ref=logsumexp(lse_v, lse_w),
         lse2_values=lse2_v, lse2_weights=lse2_w,
         lse2_pinned=np.array([7.14628895, 3.844190158, 4.82132340]),
         lse2_ref=logsumexp2D(lse2_v, lse2_w),
         gauss_mean=g_mean, gauss_sigma=g_sigma, gauss_point=g_point, gauss_pinned=1.30077135,
         gauss_inv_sigma=g.inv_sigma, gauss_log_norm=g._local_gauss.log_normalization,
         gauss_ref=g.evaluate(g_point),
         t_mean=t_mean, t_sigma=t_sigma, t_dof=5., t_points=t_points,
         t_pinned=np.array([2.200202941, 2.174596526]),
         t_inv_sigma=t.inv_sigma, t_log_norm=t._local_t.log_normalization,
         t_ref=t.multi_evaluate(t_points),
         cauchy_x=np.array([3.2]), cauchy_pinned=-3.5642087303149452,
         cauchy_ref=cauchy.evaluate(np.array([3.2]), np.array([0.])),
         perp_weights=pw, perp_pinned=0.71922309332486445, perp_ref=perp(pw),
         ess_pinned=2. / 3., ess_ref=ess(pw))

    # ------------------------------------------------------------------ mixture log-pdf
    for tag, K, D, N, seed in (("d2k3", 3, 2, 257, 21), ("d5k4", 4, 5, 300, 22),
                               ("d20k16", 16, 20, 200, 1), ("d1k2", 2, 1, 70, 23),
                               ("d7k1", 1, 7, 65, 24)):
        mu, cov, w = mk(K, D, seed)
        mix = create_gaussian_mixture(mu, cov, w)
        np.random.seed(7)
        x = mix.propose(N)
        individual = np.empty((N, K))
        out = mix.multi_evaluate(x, individual=individual)
        subset = list(range(0, K, 2))
        ind_subset = np.zeros((N, K))
        mix.multi_evaluate(x, individual=ind_subset, components=subset)
        # zero-weight component takes part in the row maximum (logsumexp2D)
        w0 = w.copy()
        w0[0] = 0.0
        mix0 = create_gaussian_mixture(mu, cov, w)
        mix0.weights[:] = w0
        out_w0 = mix0.multi_evaluate(x)
        single = np.array([mix.evaluate(xi) for xi in x[:5]])
        save("logpdf_gauss_" + tag, x=x, out=out, individual=individual,
             subset=np.array(subset), individual_subset=ind_subset, weights_zero0=w0,
             out_zero0=out_w0, evaluate_first5=single, **comp_params(mix))

    for tag, K, D, N, seed, dof in (("d3k2", 2, 3, 130, 31, 4.5), ("d30k8", 8, 30, 100, 2, 8.),
                                    ("d2k3", 3, 2, 200, 32, 1.0)):
        mu, cov, w = mk(K, D, seed)
        dofs = np.full(K, dof) + np.arange(K) * 0.25
        mix = create_t_mixture(mu, cov, dofs, w)
        np.random.seed(7)
        x = mix.propose(N)
        individual = np.empty((N, K))
        out = mix.multi_evaluate(x, individual=individual)
        save("logpdf_student_" + tag, x=x, out=out, individual=individual,
             **comp_params(mix, student=True))

    # ------------------------------------------------------------------ pruned (zero-weight) components at a dimension
    # where the matrix-product form of the GPU path applies: the reference's row maximum runs over ALL components'
    # unweighted values (logsumexp2D), so samples that sit on a dead component far from every live one come out degraded
    # (the live terms exp(a - max) underflow) or as log 0 = -inf.  All components share ONE covariance (the file stays
    # small: one inverse, as the reference computed it).  Rows 0 ... 383: samples of the live mixture; 384 ... 511: samples
    # around the dead component 3, which sits 70 sigma away, approaching it from 20 sigma.
    for tag, K, D, seed in (("d40k32", 32, 40, 51), ("d24k64", 64, 24, 52)):
        mu, cov, w = mk(K, D, seed)
        cov = np.repeat(cov[:1], K, axis=0)
        w[np.arange(K) % 5 == 3] = 0.0
        w /= w.sum()
        live = w > 0
        np.random.seed(9)
        x_live = create_gaussian_mixture(mu[live], cov[live], w[live]).propose(384)
        rs = np.random.RandomState(10)
        L = np.linalg.cholesky(cov[0])
        sig = np.sqrt(np.linalg.eigvalsh(cov[0]).max())
        far_dir = np.ones(D) / np.sqrt(D)
        dist = np.linspace(20., 70., 128)
        x_far = mu[3] + dist[:, None] * sig * far_dir + rs.normal(size=(128, D)).dot(L.T)
        mu_far = mu.copy()
        mu_far[3] = mu[3] + dist[-1] * sig * far_dir
        x = np.vstack([x_live, x_far])
        mix_far = create_gaussian_mixture(mu_far, cov, np.full(K, 1. / K))
        mix_far.weights[:] = w
        individual = np.empty((len(x), K))
        out = mix_far.multi_evaluate(x, individual=individual)
        c0 = mix_far.components[0]
        assert all(np.array_equal(c.inv_sigma, c0.inv_sigma) for c in mix_far.components)
        save("logpdf_dead_" + tag, x=x, out=out, weights=np.array(mix_far.weights), mu=mu_far, sigma0=cov[0],
             inv_sigma0=c0.inv_sigma, log_norm0=c0._local_gauss.log_normalization,
             individual_live_max=individual[:, live].max(axis=1), individual_dead_max=individual[:, ~live].max(axis=1))

    # ------------------------------------------------------------------ Student-t mixtures at dimensions where the GPU path
    # evaluates the Mahalanobis forms as a matrix product (shared scale matrix: the file stays small)
    for tag, K, D, N, seed, dof in (("d40k32", 32, 40, 512, 61, 6.), ("d64k64", 64, 64, 512, 62, 11.)):
        mu, cov, w = mk(K, D, seed)
        cov = np.repeat(cov[:1], K, axis=0)
        dofs = np.full(K, dof) + 0.5 * (np.arange(K) % 4)
        mix = create_t_mixture(mu, cov, dofs, w)
        np.random.seed(11)
        x = mix.propose(N)
        out = mix.multi_evaluate(x)
        c0 = mix.components[0]
        assert all(np.array_equal(c.inv_sigma, c0.inv_sigma) for c in mix.components)
        save("logpdf_student_shared_" + tag, x=x, out=out, weights=np.array(mix.weights), mu=mu, dof=dofs,
             inv_sigma0=c0.inv_sigma, log_norm=np.array([c._local_t.log_normalization for c in mix.components]))

    # ------------------------------------------------------------------ importance weights
    for tag, K, D, N, seed, student in (("gauss_d2", 3, 2, 400, 41, False),
                                        ("student_d5", 4, 5, 300, 42, True)):
        mu, cov, w = mk(K, D, seed)
        if student:
            prop = create_t_mixture(mu, cov, np.full(K, 6.), w)
        else:
            prop = create_gaussian_mixture(mu, cov, w)
        tmu, tcov, tw = mk(2, D, 11)
        target_mix = create_gaussian_mixture(tmu, tcov, tw)
        np.random.seed(5)
        sampler = ImportanceSampler(target_mix.evaluate, prop, save_target_values=True,
                                    rng=np.random.mtrand)
        origin = sampler.run(N, trace_sort=True)
        samples = sampler.samples[:]
        weights = sampler.weights[:][:, 0]
        np.random.seed(5)
        counts = np.random.mtrand.multinomial(N, prop.weights)
        save("is_" + tag, samples=samples, weights=weights,
             target_values=sampler.target_values[:][:, 0], origin=origin, counts=counts,
             perp=perp(weights), ess=ess(weights),
             target_mu=tmu, target_sigma=tcov, target_weights=tw,
             **{"prop_" + k: v for k, v in comp_params(prop, student).items()})

    # combine_weights (two proposals, log and linear branch)
    mu, cov, w = mk(2, 2, 51)
    p1 = create_gaussian_mixture(mu, cov, w)
    mu2, cov2, w2 = mk(3, 2, 52)
    p2 = create_gaussian_mixture(mu2, cov2, w2)
    np.random.seed(9)
    s1, s2 = p1.propose(60), p2.propose(45)
    tmu, tcov, tw = mk(2, 2, 11)
    tm = create_gaussian_mixture(tmu, tcov, tw)
    w1 = np.exp(tm.multi_evaluate(s1) - p1.multi_evaluate(s1))
    w2_ = np.exp(tm.multi_evaluate(s2) - p2.multi_evaluate(s2))
    comb_log = combine_weights([s1, s2], [w1, w2_], [p1, p2])[:][:, 0]
    w1z = w1.copy()
    w1z[::7] = 0.0
    comb_lin = combine_weights([s1, s2], [w1z, w2_], [p1, p2])[:][:, 0]
    save("combine_weights", s1=s1, s2=s2, w1=w1, w2=w2_, w1_zeros=w1z, combined_log=comb_log,
         combined_linear=comb_lin,
         **{"p1_" + k: v for k, v in comp_params(p1).items()},
         **{"p2_" + k: v for k, v in comp_params(p2).items()})

    # ------------------------------------------------------------------ variational Bayes
    def vb_state(vb, prefix):
        return {prefix + k: np.array(getattr(vb, k)) for k in
                ("alpha", "beta", "nu", "m", "W", "log_det_W", "expectation_det_ln_lambda",
                 "expectation_ln_pi", "expectation_gauss_exponent", "log_rho", "r", "N_comp",
                 "inv_N_comp", "x_mean_comp", "S")}

    for tag, K, D, N, seed, weighted, init in (("d2k3", 3, 2, 500, 61, False, "mixture"),
                                               ("d5k4w", 4, 5, 400, 62, True, "mixture"),
                                               ("d20k8", 8, 20, 300, 3, False, "mixture"),
                                               ("d3k5first", 5, 3, 350, 63, True, "first"),
                                               # the headline's shape (BASELINE metric: K = 32, D = 20)
                                               ("d20k32", 32, 20, 400, 64, False, "mixture")):
        mu, cov, w = mk(K, D, seed)
        gen = create_gaussian_mixture(mu, cov, w)
        np.random.seed(7)
        data = gen.propose(N)
        rs = np.random.RandomState(seed + 100)
        sw = rs.uniform(0.5, 1.5, size=N) if weighted else None
        kwargs = dict(weights=sw)
        if init == "mixture":
            vb = GaussianInference(data, initial_guess=gen, **kwargs)
        else:
            vb = GaussianInference(data, components=K, initial_guess="first", **kwargs)
        out = dict(data=data, sample_weights=(sw if weighted else np.zeros(0)),
                   init_mu=mu, init_sigma=cov, init_weights=w, init_kind=init,
                   alpha0=vb.alpha0, beta0=vb.beta0, nu0=vb.nu0, m0=vb.m0, W0=vb.W0)
        out.update(vb_state(vb, "e0_"))
        out["e0_bound"] = vb.likelihood_bound()
        out["e0_log_q_Z"] = vb._update_expectation_log_q_Z()
        if tag == "d20k32":
            # the headline's shape: the first E-step only, without the two N x K matrices the smaller fixtures cover
            # (K x D x D arrays are 100 KB each here)
            for k in ("e0_log_rho", "e0_expectation_gauss_exponent", "e0_log_det_W", "e0_inv_N_comp", "init_sigma", "W0", "m0"):
                out.pop(k)
            save("vb_" + tag, **out)
            continue
        vb.update()
        out.update(vb_state(vb, "u1_"))
        out["u1_bound"] = vb.likelihood_bound()
        out["u1_log_q_Z"] = vb._update_expectation_log_q_Z()
        # a fresh object driven by run(): iteration count and final posterior
        if init == "mixture":
            vb2 = GaussianInference(data, initial_guess=gen, **kwargs)
        else:
            vb2 = GaussianInference(data, components=K, initial_guess="first", **kwargs)
        nit = vb2.run(iterations=25, prune=1., rel_tol=1e-10, abs_tol=1e-5)
        out["run_iterations"] = -1 if nit is None else nit
        out["run_K"] = vb2.K
        post = vb2.posterior2prior()
        for k in ("alpha0", "beta0", "nu0", "m0", "W0"):
            out["run_post_" + k] = post[k]
        out["run_bound"] = vb2.likelihood_bound()
        mm = vb2.make_mixture()
        out["run_mix_weights"] = np.array(mm.weights)
        out["run_mix_mu"] = np.array([c.mu for c in mm.components])
        out["run_mix_sigma"] = np.array([c.sigma for c in mm.components])
        save("vb_" + tag, **out)

    # ------------------------------------------------------------------ VBMerge (mixture reduction)
    rs = np.random.RandomState(17)
    centres = np.array([[-4., 0.], [3., 3.], [2., -5.]])
    L = 18
    in_mu = centres[np.arange(L) % 3] + rs.normal(0, 0.6, (L, 2))
    in_cov = np.array([np.eye(2) * rs.uniform(0.2, 0.6) + 0.05 * np.outer(v, v) for v in rs.normal(size=(L, 2))])
    in_w = rs.uniform(0.5, 1.5, L)
    in_w /= in_w.sum()
    big = create_gaussian_mixture(in_mu, in_cov, in_w)
    merge = VBMerge(big, N=5000, components=6, initial_guess='first')
    out = dict(in_mu=in_mu, in_sigma=in_cov, in_weights=in_w, N=5000, components=6)
    out.update(vb_state(merge, "e0_"))
    out["e0_bound"] = merge.likelihood_bound()
    merge.update()
    out.update(vb_state(merge, "u1_"))
    out["u1_bound"] = merge.likelihood_bound()
    merge2 = VBMerge(big, N=5000, components=6, initial_guess='first')
    nit = merge2.run(100, prune=1.)
    out["run_iterations"] = -1 if nit is None else nit
    out["run_K"] = merge2.K
    mm = merge2.make_mixture()
    out["run_mix_weights"] = np.array(mm.weights)
    out["run_mix_mu"] = np.array([c.mu for c in mm.components])
    out["run_mix_sigma"] = np.array([c.sigma for c in mm.components])
    save("vbmerge", **out)

    # ------------------------------------------------------------------ PMC updates
    def mix_out(mix, prefix, student):
        d = {prefix + "weights": np.array(mix.weights),
             prefix + "mu": np.array([c.mu for c in mix.components]),
             prefix + "sigma": np.array([c.sigma for c in mix.components])}
        if student:
            d[prefix + "dof"] = np.array([c.dof for c in mix.components])
        return d

    for tag, K, D, N, seed in (("d2k3", 3, 2, 1000, 71), ("d5k4", 4, 5, 800, 72)):
        mu, cov, w = mk(K, D, seed)
        prop = create_gaussian_mixture(mu, cov, w)
        # target = a perturbed copy of the proposal, so the importance weights are healthy
        rs = np.random.RandomState(seed + 7)
        target = create_gaussian_mixture(mu + 0.4 * rs.normal(size=mu.shape), 1.3 * cov, w[::-1])
        np.random.seed(13)
        samples, latent = prop.propose(N, trace=True, shuffle=False)
        iw = np.exp(target.multi_evaluate(samples) - prop.multi_evaluate(samples))
        out = dict(samples=samples, latent=latent, weights=iw)
        out.update({"in_" + k: v for k, v in comp_params(prop).items()})
        cases = dict(rb_w=dict(weights=iw, latent=None, rb=True),
                     rb_u=dict(weights=None, latent=None, rb=True),
                     rb_w_latent_min=dict(weights=iw, latent=latent, rb=True,
                                          mincount=int(0.25 * N)),
                     nrb_w=dict(weights=iw, latent=latent, rb=False),
                     nrb_u=dict(weights=None, latent=latent, rb=False))
        for cname, kw in cases.items():
            res = gaussian_pmc(samples, prop, copy=True, **kw)
            out.update(mix_out(res, cname + "_", False))
        # a mixture with one dead (zero-weight) component: its zero column joins the row max
        dead = create_gaussian_mixture(mu, cov, w)
        dead.weights[1] = 0.
        dead.normalize()
        res = gaussian_pmc(samples, dead, weights=iw, copy=True)
        out["dead_in_weights"] = np.array(dead.weights)
        out.update(mix_out(res, "dead_rb_w_", False))
        # PMC.run driver
        pmc = PMC(samples, prop, weights=iw, latent=latent, rb=True)
        ll0 = pmc.log_likelihood()
        nit = pmc.run(iterations=5, prune=0.)
        out["pmcrun_ll0"] = ll0
        out["pmcrun_iterations"] = -1 if nit is None else nit
        out["pmcrun_ll"] = pmc.log_likelihood()
        out.update(mix_out(pmc.density, "pmcrun_", False))
        save("pmc_gauss_" + tag, **out)

    for tag, K, D, N, seed in (("d2k3", 3, 2, 900, 81), ("d4k3", 3, 4, 700, 82)):
        mu, cov, w = mk(K, D, seed)
        dofs = np.array([3.5, 6., 12.])[:K]
        prop = create_t_mixture(mu, cov, dofs, w)
        rs = np.random.RandomState(seed + 7)
        target = create_gaussian_mixture(mu + 0.4 * rs.normal(size=mu.shape), 1.3 * cov, w[::-1])
        np.random.seed(17)
        samples, latent = prop.propose(N, trace=True, shuffle=False)
        iw = np.exp(target.multi_evaluate(samples) - prop.multi_evaluate(samples))
        out = dict(samples=samples, latent=latent, weights=iw)
        out.update({"in_" + k: v for k, v in comp_params(prop, True).items()})
        cases = dict(rb_w_dof=dict(weights=iw, latent=None, rb=True, dof_solver_steps=100),
                     rb_w_nodof=dict(weights=iw, latent=None, rb=True, dof_solver_steps=0),
                     rb_u_dof=dict(weights=None, latent=None, rb=True, dof_solver_steps=100),
                     nrb_w_dof=dict(weights=iw, latent=latent, rb=False, dof_solver_steps=100),
                     nrb_u_nodof=dict(weights=None, latent=latent, rb=False,
                                      dof_solver_steps=0),
                     rb_w_clamp=dict(weights=iw, latent=None, rb=True, dof_solver_steps=100,
                                     mindof=5., maxdof=5.5))
        for cname, kw in cases.items():
            res = student_t_pmc(samples, prop, copy=True, **kw)
            out.update(mix_out(res, cname + "_", True))
        save("pmc_student_" + tag, **out)

    # ------------------------------------------------------------------ propose: counts / origins
    mu, cov, w = mk(5, 3, 91)
    mix = create_gaussian_mixture(mu, cov, w)
    np.random.seed(123)
    samples, origin = mix.propose(1000, trace=True, shuffle=False)
    np.random.seed(123)
    counts = np.random.mtrand.multinomial(1000, mix.weights)
    save("propose_trace", weights=np.array(mix.weights), mu=mu, sigma=cov, N=1000, seed=123,
         origin=origin, counts=counts, sample_mean=samples.mean(axis=0))

    # ------------------------------------------------------------------ examples/pmc.py (BASELINE config 1)
    # the reference's own example, seeded: bimodal 2-D Gaussian target, 3-component proposal,
    # 10 x 1000 samples with a gaussian_pmc(mincount=20, rb=True) update after every run
    tw = np.array([0.3, 0.7])
    tmeans = [np.array([5.0, 0.01]), np.array([-4.0, 1.0])]
    tcovs = [np.array([[0.01, 0.003], [0.003, 0.0025]]), np.array([[0.1, 0.], [0., 0.02]])]
    target_mixture = create_gaussian_mixture(tmeans, tcovs, tw)
    pmeans = [np.array([4.0, 0.0]), np.array([-5.0, 0.0]), np.array([0.0, 0.0])]
    initial = MixtureDensity([Gauss(m, np.eye(2)) for m in pmeans])
    np.random.seed(42)
    sampler = ImportanceSampler(target_mixture.evaluate, initial)
    out = dict(target_weights=tw, target_means=np.array(tmeans), target_covs=np.array(tcovs),
               prop_means=np.array(pmeans), seed=42, steps=10, n_per_step=1000)
    for i in range(10):
        origin = sampler.run(10 ** 3, trace_sort=True)
        samples = sampler.samples[-1]
        weights = sampler.weights[-1][:, 0]
        gaussian_pmc(samples, sampler.proposal, weights, origin, mincount=20, rb=True, copy=False)
        out["origin_%d" % i] = origin
        out["weights_%d" % i] = weights.copy()
        if i == 0:
            out["samples_0"] = samples.copy()
        out["prop_weights_%d" % i] = np.array(sampler.proposal.weights)
        out["prop_mu_%d" % i] = np.array([c.mu for c in sampler.proposal.components])
        out["prop_sigma_%d" % i] = np.array([c.sigma for c in sampler.proposal.components])
    save("example_pmc", **out)

    print("reference version", pypmc.__version__)


if __name__ == "__main__":
    main()
