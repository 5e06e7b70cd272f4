"""Concurrent use of the kernel-level ABI from several HIP streams (include/pmc_hip.h: "calls on different streams may run
concurrently if they are given different workspaces"): the importance-weight pass and the E-step -- the common-shift
statistics with their control block, reductions and skipped fall-back launches included -- side by side on PyTorch's
non-blocking pool streams, against the serial results, bit for bit (the kernels are deterministic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_is_pass_and_estep_on_concurrent_streams():
    import torch
    from bench import mk, gauss_params, vb_params
    from pypmc_amd.backend import HipBackend, ComponentSet
    be = HipBackend()
    K, D, KT, N = 32, 20, 4, 600_000                       # N * ceil(K / 32) >= 524288: k_stats_gemm runs
    mu, cov, w = mk(K, D, 1)
    tmu, tcov, tw = mk(KT, D, 11)
    inv, ln = gauss_params(mu, cov)
    tinv, tln = gauss_params(tmu, tcov)
    W, beta, nu, ln_pi, ln_lambda = vb_params(mu, cov, w, N)
    prop = ComponentSet(0, mu, inv, c0=ln, weight=w)
    tgt = ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
    post = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
    pp, pt, pv = be.pack(prop), be.pack(tgt), be.pack(post)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g) * 1.1
    x += torch.tensor(mu, device="cuda")[torch.randint(0, K, (N,), device="cuda", generator=g)]
    x2 = (x * 0.97 + 0.05).contiguous()                    # a second batch for the second E-step stream

    ref_is = be.importance_weights(x, prop, tgt, pack=pp, target_pack=pt)
    ref_w, ref_sc = ref_is["weights"].clone(), ref_is["scalars"].clone()
    ref_e1 = be.estep(x, post, 0, pack=pv)["stats"].clone()
    ref_e2 = be.estep(x2, post, 0, pack=pv)["stats"].clone()
    torch.cuda.synchronize()
    assert float(ref_e1[8:8 + K * int(be.lib.pmc_stats_stride(D))].reshape(K, -1)[:, 0].sum()) == pytest.approx(N, rel=1e-9)

    s_is, s_e1, s_e2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    out1, out2 = be.zeros(be.stats_len(K, D)), be.zeros(be.stats_len(K, D))
    for it in range(8):
        for s in (s_is, s_e1, s_e2):
            s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_e1):
            e1 = be.estep(x, post, 0, pack=pv, out=out1)["stats"]
        with torch.cuda.stream(s_is):
            r = be.importance_weights(x, prop, tgt, pack=pp, target_pack=pt)
        with torch.cuda.stream(s_e2):
            e2 = be.estep(x2, post, 0, pack=pv, out=out2)["stats"]
        for s in (s_is, s_e1, s_e2):
            s.synchronize()
        assert torch.equal(r["weights"], ref_w) and torch.equal(r["scalars"], ref_sc), "IS pass, round %d" % it
        assert torch.equal(e1, ref_e1), "E-step on stream 1, round %d" % it
        assert torch.equal(e2, ref_e2), "E-step on stream 2, round %d" % it


def test_two_host_threads_each_with_its_stream():
    """ctypes releases the GIL during a call: two host threads drive the library at the same time, each on a stream and
    a workspace of its own (thread-local error state, mutex-guarded scratch registry and timing pool)"""
    import threading
    import torch
    from pypmc_amd.backend import HipBackend
    from pypmc_amd.density.mixture import create_gaussian_mixture, create_t_mixture, component_set
    from bench import mk
    be = HipBackend()
    cases = []
    for seed, student, D, K, N in ((1, False, 20, 16, 200_000), (2, True, 12, 24, 150_000)):
        mu, cov, w = mk(K, D, seed)
        mix = create_t_mixture(mu, cov, np.full(K, 7.), w) if student else create_gaussian_mixture(mu, cov, w)
        np.random.seed(seed)
        x = be.asdevice(mix.propose(N))
        wts = be.asdevice(np.random.uniform(0.5, 1.5, N))
        cs = component_set(mix.components, mix.weights)
        ref_l = be.logpdf(x, cs, want_scalars=True)
        ref_e = be.estep(x, cs, 1, sample_w=wts)["stats"].clone()
        cases.append((x, wts, cs, ref_l["out"].clone(), ref_l["scalars"].clone(), ref_e))
    torch.cuda.synchronize()
    be.kernel_timing(True)                                  # the event pool is shared between the threads too
    errors = []

    def work(case):
        x, wts, cs, ref_out, ref_sc, ref_e = case
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(15):
                    got = be.logpdf(x, cs, want_scalars=True)
                    e = be.estep(x, cs, 1, sample_w=wts)["stats"]
                    st.synchronize()
                    assert torch.equal(got["out"], ref_out) and torch.equal(got["scalars"], ref_sc)
                    assert torch.equal(e, ref_e)
        except Exception as exc:                            # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(c,)) for c in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    be.kernel_timing(False)
    timings = be.kernel_timings()
    assert not errors, errors
    assert timings["k_logpdf"]["calls"] == 30 and timings["k_resp"]["calls"] + timings.get("k_estep_fused", {"calls": 0})["calls"] == 30
