// Driver of the sanitizer job (tests/test_sanitizers.py): walks the HOST side of libpmc_hip.so -- built with
// -fsanitize=address,undefined against the stand-in runtime of hip_stub.cpp -- through its argument checks, pack
// building, workspace layout, scratch-slot and event bookkeeping, the handle layer's buffers and conversions, and two
// contexts in two threads.  Kernels do not run: values are not checked (the GPU suite does), only that every path the
// host code takes is clean.  Exit code 0 and no sanitizer report = pass.
#include "../../include/pmc_ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int pmc_stub_leaked_streams(void);
extern "C" {
#include <stdint.h>
}
// the stand-in's allocation calls (same signatures as the runtime's)
extern "C" int hipMalloc(void **, size_t);
extern "C" int hipFree(void *);

static int g_fail = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::fprintf(stderr, "host_checks: %s:%d: %s  [%s]\n", __FILE__, __LINE__, #cond, pmc_last_error()); \
            ++g_fail;                                                             \
        }                                                                         \
    } while (0)

struct Lcg {
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 2862933555777941757ull + 3037000493ull) {}
    double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
    double sym() { return 2.0 * uni() - 1.0; }
};

struct Mixture {
    int K, D;
    std::vector<double> w, mu, prec, ln, dof;
};
static Mixture make_mixture(int K, int D, uint64_t seed)
{
    Lcg r(seed);
    Mixture m{K, D, std::vector<double>(K), std::vector<double>((size_t)K * D), std::vector<double>((size_t)K * D * D),
              std::vector<double>(K), std::vector<double>(K)};
    double ws = 0.0;
    for (int k = 0; k < K; ++k) ws += (m.w[k] = 0.5 + r.uni());
    for (int k = 0; k < K; ++k) {
        m.w[k] /= ws;
        m.ln[k] = -0.5 * D * std::log(2.0 * M_PI) + 0.1 * r.sym();
        m.dof[k] = 3.0 + 4.0 * r.uni();
        for (int i = 0; i < D; ++i) m.mu[(size_t)k * D + i] = 3.0 * r.sym();
        std::vector<double> A((size_t)D * D);
        for (double &a : A) a = r.sym();
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double s = i == j ? 0.5 : 0.0;
                for (int l = 0; l < D; ++l) s += A[(size_t)i * D + l] * A[(size_t)j * D + l] / D;
                m.prec[((size_t)k * D + i) * D + j] = s;
            }
    }
    return m;
}

struct Dev {                                               // "device" buffer of the stand-in runtime
    void *p = nullptr;
    explicit Dev(size_t bytes) { hipMalloc(&p, bytes ? bytes : 8); }
    ~Dev() { hipFree(p); }
    double *d() const { return (double *)p; }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
};

static void kernel_level(int K, int D, int64_t N, uint64_t seed)
{
    const Mixture m = make_mixture(K, D, seed);
    const int64_t stride = pmc_pack_stride(D);
    EXPECT(stride > 0);
    std::vector<double> pack((size_t)K * stride), c1(K), c2(K);
    for (int k = 0; k < K; ++k) { c1[k] = -.5 * (m.dof[k] + D); c2[k] = 1. / m.dof[k]; }
    EXPECT(pmc_pack_components(K, D, m.mu.data(), m.prec.data(), m.ln.data(), c1.data(), c2.data(), m.dof.data(), m.w.data(),
                               nullptr, pack.data()) == PMC_OK);
    std::vector<double> means((size_t)K * stride);
    EXPECT(pmc_pack_means(K, D, m.mu.data(), means.data()) == PMC_OK);
    const int64_t wsb = pmc_workspace_bytes(N, K, D);
    EXPECT(wsb > 0);
    const int PS = (int)pmc_stats_stride(D);
    const int64_t tl = pmc_tile_buffer_len(N, K), gl = pmc_gscale_len(N, K);
    Dev x(sizeof(double) * (size_t)N * D), dpack(sizeof(double) * pack.size()), dmeans(sizeof(double) * means.size()),
        ws((size_t)wsb), out(8 * (size_t)N), ind(8 * (size_t)N * K), wts(8 * (size_t)N), lt(8 * (size_t)N), sc(8 * 8),
        u(8 * (size_t)tl), scratch(8 * (size_t)tl), gs(8 * (size_t)gl), vs(8 * 2 * (size_t)K), stats(8 * (size_t)K * PS),
        tiles(8 * (size_t)pmc_maha_tiles_size(N, K)), lat(8 * (size_t)N);
    std::memcpy(dpack.p, pack.data(), sizeof(double) * pack.size());
    std::memcpy(dmeans.p, means.data(), sizeof(double) * means.size());
    for (int kind = 0; kind < 2; ++kind) {
        EXPECT(pmc_mixture_logpdf(x.d(), N, D, dpack.d(), K, kind, 0, out.d(), ind.d(), K, lt.d(), wts.d(), nullptr, sc.d(), ws.p, nullptr) == PMC_OK);
        EXPECT(pmc_mixture_logpdf(x.d(), N, D, dpack.d(), K, kind, 1, out.d(), nullptr, K, nullptr, nullptr, wts.d(), nullptr, nullptr, nullptr) == PMC_OK);
        EXPECT(pmc_mixture_logpdf(x.d(), N, D, dpack.d(), K, kind, 0, out.d(), nullptr, K, nullptr, nullptr, nullptr, nullptr, ws.p, nullptr) == PMC_OK);
        EXPECT(pmc_mixture_logpdf_keep(x.d(), N, D, dpack.d(), K, kind, 0, out.d(), nullptr, K, nullptr, nullptr, nullptr, sc.d(), ws.p, tiles.d(), nullptr) == PMC_OK);
        EXPECT(pmc_importance_weights(x.d(), N, D, dpack.d(), K, kind, dpack.d(), K, 1 - kind, out.d(), lt.d(), wts.d(), nullptr, sc.d(), ws.p, nullptr) == PMC_OK);
        EXPECT(pmc_importance_weights_keep(x.d(), N, D, dpack.d(), K, kind, dpack.d(), K, kind, nullptr, nullptr, wts.d(), nullptr, sc.d(), ws.p, tiles.d(), nullptr) == PMC_OK);
        if (D <= 64) {
            EXPECT(pmc_importance_weights_emit(x.d(), N, D, dpack.d(), K, kind, dpack.d(), K, 0, nullptr, nullptr, wts.d(), sc.d(), ws.p, u.d(), vs.d(), nullptr) == PMC_OK);
            EXPECT(pmc_importance_weights_emit_grouped(x.d(), N, D, dpack.d(), K, kind, dpack.d(), K, 0, out.d(), nullptr, wts.d(), sc.d(), ws.p, u.d(), gs.d(), vs.d(), nullptr) == PMC_OK);
            EXPECT(pmc_estep_from_u_grouped(x.d(), N, D, dpack.d(), K, kind, u.d(), gs.d(), stats.d(), ws.p, nullptr) == PMC_OK);
        }
        EXPECT(pmc_estep_from_u(x.d(), N, D, dpack.d(), K, kind, u.d(), stats.d(), ws.p, nullptr) == PMC_OK);
        EXPECT(pmc_estep_from_tiles(x.d(), N, D, dpack.d(), K, kind, 0, wts.d(), tiles.d(), K, u.d(), vs.d(), stats.d(), sc.d(), ws.p, nullptr) == PMC_OK);
        for (int mode = 1; mode <= 2; ++mode) {
            EXPECT(pmc_responsibilities(x.d(), N, D, dpack.d(), K, kind, mode, 0, wts.d(), (const int64_t *)lat.p, u.d(), scratch.d(), vs.d(), ind.d(), nullptr, nullptr, K, sc.d(), ws.p, nullptr) == PMC_OK);
            const int fused = pmc_estep_is_fused(K, D, kind, mode);
            EXPECT(pmc_estep(x.d(), N, D, dpack.d(), K, kind, mode, 0, wts.d(), (const int64_t *)lat.p, fused ? nullptr : u.d(), scratch.d(), vs.d(), stats.d(), sc.d(), ws.p, nullptr) == PMC_OK);
        }
    }
    EXPECT(pmc_responsibilities(x.d(), N, D, dpack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, nullptr, nullptr, u.d(), nullptr, nullptr, ind.d(), ind.d(), ind.d(), K, sc.d(), ws.p, nullptr) == PMC_OK);
    EXPECT(pmc_estep_about(x.d(), N, D, dpack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, wts.d(), nullptr, u.d(), nullptr, nullptr, stats.d(), sc.d(), ws.p, dmeans.d(), nullptr) == PMC_OK);
    EXPECT(pmc_sufficient_stats(x.d(), N, D, dmeans.d(), K, u.d(), stats.d(), ws.p, nullptr) == PMC_OK);
    EXPECT(pmc_weight_sums(wts.d(), N, sc.d(), ws.p, nullptr) == PMC_OK);
    EXPECT(pmc_logsumexp2d(ind.d(), wts.d(), N, K, out.d(), nullptr) == PMC_OK);
    // error paths
    EXPECT(pmc_responsibilities(x.d(), N, D, dpack.d(), K, PMC_KIND_VB, PMC_RESP_PMC_RB, 0, nullptr, nullptr, u.d(), nullptr, nullptr, nullptr, nullptr, nullptr, K, nullptr, nullptr, nullptr) == PMC_EINVAL);
    EXPECT(pmc_mixture_logpdf(x.d(), N, D, dpack.d(), K, 7, 0, out.d(), nullptr, K, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == PMC_EINVAL);
    EXPECT(pmc_mixture_logpdf(x.d(), N, D, dpack.d(), K, 0, 0, out.d(), ind.d(), K - 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == PMC_EINVAL);
    EXPECT(pmc_estep(x.d(), N, D, dpack.d(), K, 0, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, stats.d(), nullptr, ws.p, nullptr) == PMC_EINVAL);
}

// nparts = 0: pmc_init(0); nparts > 0: a context of that many parts, all on device 0 (virtual shards: own streams, own
// scratch, own host threads, the K-sized vectors added in part order)
static void handle_layer(int nparts, int K, int D, int64_t N, uint64_t seed)
{
    const Mixture m = make_mixture(K, D, seed), t = make_mixture(3, D, seed + 1);
    pmc_ctx *ctx = nullptr;
    if (nparts == 0) {
        EXPECT(pmc_init(0, &ctx) == PMC_OK && ctx);
        nparts = 1;
    } else {
        const std::vector<int> ids((size_t)nparts, 0);
        EXPECT(pmc_init_devices(nparts, ids.data(), &ctx) == PMC_OK && ctx);
    }
    EXPECT(pmc_ctx_device_count(ctx) == nparts);
    int devs[64];
    EXPECT(pmc_ctx_devices(ctx, devs, 64) == nparts && devs[0] == 0);
    pmc_mix *q = nullptr, *tq = nullptr, *st = nullptr;
    EXPECT(pmc_mixture_create(ctx, PMC_KIND_GAUSS, K, D, m.w.data(), m.mu.data(), m.prec.data(), m.ln.data(), nullptr, &q) == PMC_OK);
    EXPECT(pmc_mixture_create(ctx, PMC_KIND_GAUSS, 3, D, t.w.data(), t.mu.data(), t.prec.data(), t.ln.data(), nullptr, &tq) == PMC_OK);
    EXPECT(pmc_mixture_create(ctx, PMC_KIND_STUDENT_T, K, D, m.w.data(), m.mu.data(), m.prec.data(), m.ln.data(), m.dof.data(), &st) == PMC_OK);
    EXPECT(pmc_mixture_create(ctx, PMC_KIND_STUDENT_T, K, D, m.w.data(), m.mu.data(), m.prec.data(), m.ln.data(), nullptr, &st) == PMC_EINVAL);
    std::vector<double> bad(m.prec);
    bad[0] = -1.0;
    pmc_mix *none = nullptr;
    EXPECT(pmc_mixture_create(ctx, PMC_KIND_GAUSS, K, D, m.w.data(), m.mu.data(), bad.data(), m.ln.data(), nullptr, &none) == PMC_ENOTPOSDEF && !none);
    EXPECT(pmc_mixture_update(q, m.w.data(), m.mu.data(), m.prec.data(), m.ln.data(), nullptr) == PMC_OK);
    Lcg r(seed + 2);
    std::vector<double> x((size_t)N * D), lt(N), w(N), out(N), ind((size_t)N * K), sums(3);
    for (double &v : x) v = 3.0 * r.sym();
    pmc_samples *s = nullptr, *gen = nullptr;
    EXPECT(pmc_samples_upload(ctx, x.data(), N, D, &s) == PMC_OK);
    EXPECT(pmc_samples_count(s) == N);
    {
        int64_t b = -1, c = -1, total = 0;
        for (int p = 0; p < nparts; ++p) {
            EXPECT(pmc_samples_shard(s, p, &b, &c) == 0 && b == total);
            total += c;
        }
        EXPECT(total == N && pmc_samples_shard(s, nparts, &b, &c) == PMC_EINVAL);
    }
    EXPECT(pmc_samples_download(s, x.data()) == PMC_OK);
    std::vector<int64_t> counts(K, N / K), origin((size_t)(N / K) * K);
    EXPECT(pmc_samples_generate(ctx, q, nullptr, counts.data(), 1234u, 0, &gen) == PMC_OK);
    EXPECT(pmc_samples_origin(gen, origin.data()) == PMC_OK && pmc_samples_origin(s, origin.data()) == PMC_EINVAL);
    EXPECT(pmc_mix_logpdf(q, s, out.data(), ind.data()) == PMC_OK);
    EXPECT(pmc_mix_logpdf(st, s, out.data(), nullptr) == PMC_OK);
    const int32_t comps[2] = {0, K - 1};
    EXPECT(pmc_mix_logpdf_components(q, s, comps, 2, ind.data()) == PMC_OK);
    const int32_t badc[1] = {K};
    EXPECT(pmc_mix_logpdf_components(q, s, badc, 1, ind.data()) == PMC_EINVAL);
    EXPECT(pmc_is_weights(q, s, lt.data(), nullptr, w.data(), nullptr, sums.data()) == PMC_OK);
    EXPECT(pmc_is_weights(q, s, nullptr, tq, w.data(), lt.data(), sums.data()) == PMC_OK);
    EXPECT(pmc_is_weights(q, s, lt.data(), tq, w.data(), nullptr, sums.data()) == PMC_EINVAL);
    // VB E-step from variational parameters
    std::vector<double> W(m.prec), nu(K), beta(K), lnpi(K), lnlam(K), Nk(K), xbar((size_t)K * D), S((size_t)K * D * D), elq(1),
        rr((size_t)N * K), lr((size_t)N * K);
    for (int k = 0; k < K; ++k) { nu[k] = D + 2.0 + r.uni(); beta[k] = 1.0 + r.uni(); lnpi[k] = -std::log((double)K); lnlam[k] = r.sym(); }
    EXPECT(pmc_vb_estep(ctx, s, nullptr, K, m.mu.data(), W.data(), nu.data(), beta.data(), lnpi.data(), lnlam.data(), nullptr,
                        Nk.data(), xbar.data(), S.data(), elq.data(), rr.data(), lr.data()) == PMC_OK);
    EXPECT(pmc_vb_estep(ctx, s, w.data(), K, m.mu.data(), W.data(), nu.data(), beta.data(), lnpi.data(), lnlam.data(), m.mu.data(),
                        Nk.data(), xbar.data(), S.data(), elq.data(), nullptr, nullptr) == PMC_OK);
    // the K-sized state of a VB fit on the device (one device, or several: the state on the first): every call path, buffers of
    // exactly the documented sizes
    {
        pmc_vb_state *vs = nullptr;
        if (D > pmc_vb_max_dim()) {
            EXPECT(pmc_vb_state_create(ctx, K, D, &vs) == PMC_EINVAL);
        } else {
            EXPECT(pmc_vb_state_create(ctx, K, D, &vs) == PMC_OK);
            EXPECT(pmc_vb_state_result_len(K) == 8 * K + 16);
            std::vector<double> res((size_t)pmc_vb_state_result_len(K)), parts(2 * (size_t)K, -1.0);
            for (int fld = 0; fld < PMC_VB_E_M; ++fld) {
                const size_t len = (fld == PMC_VB_M0 || fld == PMC_VB_M || fld == PMC_VB_X_MEAN || fld == PMC_VB_SHIFT_PREV) ? (size_t)K * D
                                   : (fld == PMC_VB_INV_W0 || fld == PMC_VB_W || fld == PMC_VB_S) ? (size_t)K * D * D : (size_t)K;
                std::vector<double> v(len, 1.0), back(len, 0.0);
                EXPECT(pmc_vb_state_put(vs, fld, v.data()) == PMC_OK);
                EXPECT(pmc_vb_state_get(vs, fld, back.data()) == PMC_OK && back == v);
            }
            std::vector<double> big((size_t)K * D * D);
            EXPECT(pmc_vb_state_put(vs, PMC_VB_E_M, big.data()) == PMC_EINVAL);          // read-only
            EXPECT(pmc_vb_state_get(vs, PMC_VB_E_W, big.data()) == PMC_EINVAL);          // no E-step yet
            EXPECT(pmc_vb_state_step(vs, nullptr, PMC_VB_DO_MSTEP, nullptr, nullptr) == PMC_OK);  // queued
            EXPECT(pmc_vb_state_step(vs, nullptr, PMC_VB_DO_ESTEP, nullptr, res.data()) == PMC_EINVAL);
            EXPECT(pmc_vb_state_step(vs, s, PMC_VB_DO_ESTEP, nullptr, nullptr) == PMC_EINVAL);
            EXPECT(pmc_vb_state_step(vs, s, PMC_VB_DO_ESTEP, nullptr, res.data()) == (nparts == 1 ? PMC_OK : PMC_EINVAL));   // (the device's psi: one device)
            EXPECT(pmc_vb_state_step(vs, s, PMC_VB_DO_ESTEP, parts.data(), res.data()) == PMC_OK);
            EXPECT(pmc_vb_state_step(vs, s, PMC_VB_DO_MSTEP | PMC_VB_DO_ESTEP | PMC_VB_DO_BOUND | PMC_VB_ABOUT_PREV, parts.data(), res.data()) == PMC_OK);
            EXPECT(pmc_vb_state_step(vs, nullptr, PMC_VB_DO_BOUND, nullptr, res.data()) == PMC_OK);
            EXPECT(pmc_vb_state_get(vs, PMC_VB_E_W, big.data()) == PMC_OK);
            // run()'s loop in the library: with the device's psi and with a callback; the stub's kernels leave zeros behind
            // (no finite entry flagged in S): the first update hands the block back for a look
            int info[4] = {-1, -1, -1, -1};
            double bounds[2] = {1.0, 1.0};
            std::vector<double> n0((size_t)K, 1.0);
            if (nparts == 1) {
                EXPECT(pmc_vb_state_run(vs, s, 3, 0.0, 1.0, 1e-10, 1e-5, 0, n0.data(), nullptr, nullptr, res.data(), info, bounds) == PMC_OK);
                EXPECT(info[0] == 1 && info[1] == PMC_VB_RUN_LOOK);
            }
            struct Seen { int calls, K; } seen = {0, 0};
            auto psi = [](void *user, int k, const double *n, double *parts) {
                Seen *sn = (Seen *)user;
                sn->calls += 1;
                sn->K = k;
                for (int i = 0; i < 2 * k; ++i) parts[i] = n[i % k];
            };
            EXPECT(pmc_vb_state_run(vs, s, 2, 5.0, 0.0, 0.0, 0.0, 1, n0.data(), psi, &seen, res.data(), info, bounds) == PMC_OK);
            EXPECT(seen.calls == info[0] && seen.K == K && info[0] >= 1);
            EXPECT(pmc_vb_state_run(vs, s, 0, 5.0, 0.0, 0.0, 0.0, 1, n0.data(), nullptr, nullptr, res.data(), info, bounds) == PMC_OK && info[0] == 0 &&
                   info[1] == PMC_VB_RUN_CAP);
            EXPECT(pmc_vb_state_run(vs, nullptr, 1, 5.0, 0.0, 0.0, 0.0, 1, n0.data(), nullptr, nullptr, res.data(), info, bounds) == PMC_EINVAL);
            EXPECT(pmc_vb_state_destroy(vs) == PMC_OK);
            EXPECT(pmc_vb_state_create(ctx, K, 65, &vs) == PMC_EINVAL);
        }
        EXPECT(pmc_vb_state_destroy(nullptr) == PMC_OK);
    }
    // PMC updates
    std::vector<double> alpha(K), mu2((size_t)K * D), sig((size_t)K * D * D), dofc(K), ll(1), nrm(1);
    std::vector<int64_t> latent(N);
    for (int64_t n = 0; n < N; ++n) latent[n] = n % K;
    EXPECT(pmc_pmc_update_stats(ctx, q, s, w.data(), 0, nullptr, 1, alpha.data(), mu2.data(), sig.data(), nullptr, ll.data(), nrm.data()) == PMC_OK);
    EXPECT(pmc_pmc_update_stats(ctx, q, s, nullptr, 1, nullptr, 1, alpha.data(), mu2.data(), sig.data(), nullptr, nullptr, nullptr) == PMC_OK);
    EXPECT(pmc_pmc_update_stats(ctx, q, s, nullptr, 0, latent.data(), 0, alpha.data(), mu2.data(), sig.data(), nullptr, nullptr, nrm.data()) == PMC_OK);
    EXPECT(pmc_pmc_update_stats(ctx, q, gen, nullptr, 0, nullptr, 0, alpha.data(), mu2.data(), sig.data(), nullptr, nullptr, nullptr) == PMC_OK);
    EXPECT(pmc_pmc_update_stats(ctx, st, s, w.data(), 0, nullptr, 1, alpha.data(), mu2.data(), sig.data(), dofc.data(), ll.data(), nrm.data()) == PMC_OK);
    std::vector<double> mean(D), cov((size_t)D * D);
    EXPECT(pmc_weighted_moments(ctx, s, w.data(), 0, mean.data(), cov.data()) == PMC_OK);
    EXPECT(pmc_weighted_moments(ctx, s, nullptr, 1, mean.data(), nullptr) == PMC_OK);
    // the context's own options and timing record
    EXPECT(pmc_ctx_configure(ctx, "stats_common_shift_limit", 0.0) == PMC_OK && pmc_ctx_configure(ctx, "nope", 1.0) == PMC_EINVAL);
    EXPECT(pmc_ctx_timing_enable(ctx, 1) == PMC_OK);
    EXPECT(pmc_mix_logpdf(q, s, out.data(), nullptr) == PMC_OK);
    pmc_timing tm[16];
    int nt = 0;
    EXPECT(pmc_ctx_get_timings(ctx, tm, 16, &nt) == PMC_OK && nt == 1 && tm[0].calls == (N >= nparts ? nparts : (int)N));
    EXPECT(pmc_ctx_get_timings(ctx, tm, 16, &nt) == PMC_OK && nt == 0);
    // mismatched handles
    pmc_ctx *other = nullptr;
    EXPECT(pmc_init(0, &other) == PMC_OK);
    pmc_samples *so = nullptr;
    EXPECT(pmc_samples_upload(other, x.data(), 4, D, &so) == PMC_OK);
    EXPECT(pmc_mix_logpdf(q, so, out.data(), nullptr) == PMC_EINVAL);
    pmc_vb_state *orphan = nullptr;
    EXPECT(pmc_vb_state_create(other, 2, 3, &orphan) == PMC_OK);
    EXPECT(pmc_samples_free(so) == PMC_OK && pmc_shutdown(other) == PMC_OK);
    // a state that outlives its context (a garbage collector's order): refused by every call, destroyed already
    std::vector<double> three(3 * 2 * 3, 0.0);
    EXPECT(pmc_vb_state_get(orphan, PMC_VB_M, three.data()) == PMC_EINVAL && pmc_vb_state_put(orphan, PMC_VB_M, three.data()) == PMC_EINVAL);
    EXPECT(pmc_vb_state_step(orphan, nullptr, PMC_VB_DO_MSTEP, nullptr, nullptr) == PMC_EINVAL);
    EXPECT(pmc_vb_state_destroy(orphan) == PMC_OK);
    EXPECT(pmc_samples_free(s) == PMC_OK && pmc_samples_free(gen) == PMC_OK);
    EXPECT(pmc_mixture_destroy(q) == PMC_OK && pmc_mixture_destroy(tq) == PMC_OK && pmc_mixture_destroy(st) == PMC_OK);
    EXPECT(pmc_shutdown(ctx) == PMC_OK);
}

int main()
{
    EXPECT(pmc_abi_version() == PMC_ABI_VERSION);
    EXPECT(pmc_device_count() == 1);
    char arch[64];
    EXPECT(pmc_device_arch(0, arch, sizeof(arch)) == PMC_OK && std::strncmp(arch, "gfx950", 6) == 0);
    EXPECT(pmc_device_arch(0, nullptr, 0) == PMC_EINVAL);
    for (int D = -1; D <= 1030; D += (D < 70 ? 1 : 97)) {
        const int p = pmc_padded_dim(D);
        EXPECT((D >= 1 && D <= 1024) ? (p >= D) : (p < 0));
    }
    // workspace sizes over the shapes the dispatcher tells apart
    const int Ks[] = {1, 9, 16, 17, 32, 33, 64, 128, 1000}, Ds[] = {1, 2, 5, 7, 8, 20, 31, 32, 40, 48, 64, 70, 300};
    const int64_t Ns[] = {0, 1, 63, 64, 65, 16384, 40000, 1000000, 12500000};
    for (int K : Ks)
        for (int D : Ds)
            for (int64_t N : Ns) {
                EXPECT(pmc_workspace_bytes(N, K, D) > 0);
                EXPECT(pmc_maha_gemm_tiles(N, K, D) >= 0);
            }
    EXPECT(pmc_workspace_bytes(-1, 3, 3) == PMC_EINVAL && pmc_workspace_bytes(10, 3, 2000) == PMC_EINVAL);
    // packs: a matrix that does not factorise, non-finite entries
    {
        double mu[2] = {0, 0}, P[4] = {1, 2, 2, 1}, pack[64];
        EXPECT(pmc_pack_components(1, 2, mu, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pack) == PMC_ENOTPOSDEF);
        P[1] = P[2] = NAN;
        P[3] = 5;
        EXPECT(pmc_pack_components(1, 2, mu, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pack) == PMC_ENOTPOSDEF);
        EXPECT(pmc_pack_components(0, 2, mu, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pack) == PMC_EINVAL);
    }
    // the K-sized host conversion with zeros, NaN and inf among the sums
    {
        const int K = 4, D = 3, PS = 1 + D + D * (D + 1) / 2;
        std::vector<double> st((size_t)K * PS, 0.0), shift((size_t)K * D, 1.0), S0(K), M1((size_t)K * D), mean((size_t)K * D),
            cov((size_t)K * D * D), ncov(K, 2.0);
        Lcg r(5);
        for (double &v : st) v = r.uni();
        st[0] = 0.0;
        st[PS] = NAN;
        st[2 * PS + 1] = INFINITY;
        int far = -1;
        EXPECT(pmc_host_convert_stats(K, D, st.data(), shift.data(), nullptr, S0.data(), M1.data(), mean.data(), cov.data(), &far) == PMC_OK);
        EXPECT(pmc_host_convert_stats(K, D, st.data(), shift.data(), ncov.data(), S0.data(), nullptr, mean.data(), cov.data(), &far) == PMC_OK);
        EXPECT(pmc_host_convert_stats(0, D, st.data(), shift.data(), nullptr, S0.data(), M1.data(), mean.data(), cov.data(), &far) == PMC_EINVAL);
    }
    EXPECT(pmc_configure("stats_common_shift_min_k", 2) == PMC_OK && pmc_configure("stats_common_shift_min_k", 17) == PMC_OK);
    EXPECT(pmc_configure("unknown", 1) == PMC_EINVAL && pmc_configure(nullptr, 1) == PMC_EINVAL);
    EXPECT(pmc_configure("maha_gemm_tolerance", 2.0) == PMC_EINVAL);
    // kernel level: one-kernel E-step, two kernels, padded and exact units, the matrix-product forms, the run-time-dimension unit
    EXPECT(pmc_timing_enable(1) == PMC_OK);
    kernel_level(3, 2, 1000, 1);
    kernel_level(5, 7, 257, 2);
    kernel_level(17, 9, 70, 3);
    kernel_level(32, 20, 20000, 4);
    kernel_level(64, 40, 40000, 5);                       // k_mgemm + its fall-back launches
    kernel_level(33, 37, 33000, 6);
    kernel_level(128, 40, 40000, 14);                     // (pack building on several host threads)
    kernel_level(4, 70, 600, 7);                          // pmc_big.hip's unit (chunked scratch)
    kernel_level(32, 20, 600000, 8);                      // k_resp_groups + k_stats_gemm
    kernel_level(1, 1, 1, 9);
    EXPECT(pmc_configure("big_dim_scratch_bytes", 4096) == PMC_OK);
    kernel_level(3, 66, 1500, 10);
    EXPECT(pmc_configure("big_dim_scratch_bytes", 256.0 * 1024 * 1024) == PMC_OK);
    pmc_timing tm[16];
    int nt = 0;
    EXPECT(pmc_get_timings(tm, 16, &nt) == PMC_OK && nt >= 3);
    EXPECT(pmc_get_timings(tm, 0, nullptr) == PMC_EINVAL);
    EXPECT(pmc_timing_enable(0) == PMC_OK);
    // handle layer
    handle_layer(0, 4, 3, 1000, 11);
    handle_layer(0, 8, 20, 20011, 12);
    handle_layer(0, 5, 70, 300, 13);
    // contexts of several parts (virtual shards on the one device of the stand-in): worker threads, slots, ordered sum
    handle_layer(2, 4, 3, 1000, 31);
    handle_layer(3, 8, 20, 20011, 32);
    handle_layer(6, 5, 70, 4, 33);                        // fewer samples than parts: empty shards
    handle_layer(8, 3, 2, 17, 34);
    pmc_ctx *none = nullptr;
    EXPECT(pmc_init(1, &none) == PMC_ENODEVICE && !none);
    {
        const int bad_ids[2] = {0, 1};
        EXPECT(pmc_init_devices(2, bad_ids, &none) == PMC_ENODEVICE && !none);
        EXPECT(pmc_init_devices(-1, nullptr, &none) == PMC_EINVAL && pmc_init_devices(2, nullptr, &none) == PMC_EINVAL);
        setenv("PMC_HIP_DEVICES", "0, 0,0", 1);
        EXPECT(pmc_init_devices(0, nullptr, &none) == PMC_OK && none && pmc_ctx_device_count(none) == 3);
        EXPECT(pmc_shutdown(none) == PMC_OK);
        none = nullptr;
        setenv("PMC_HIP_DEVICES", "0,x", 1);
        EXPECT(pmc_init_devices(0, nullptr, &none) == PMC_EINVAL && !none);
        unsetenv("PMC_HIP_DEVICES");
        EXPECT(pmc_init_devices(0, nullptr, &none) == PMC_OK && none && pmc_ctx_device_count(none) == 1);   // every visible device
        EXPECT(pmc_shutdown(none) == PMC_OK);
        none = nullptr;
    }
    EXPECT(pmc_shutdown(nullptr) == PMC_OK && pmc_mixture_destroy(nullptr) == PMC_OK && pmc_samples_free(nullptr) == PMC_OK);
    // many contexts in sequence: a destroyed stream's scratch slot goes to the next new stream (256 slots in all)
    for (int i = 0; i < 600; ++i) {
        pmc_ctx *c = nullptr;
        EXPECT(pmc_init(0, &c) == PMC_OK);
        double xs[6] = {0, 1, 2, 3, 4, 5}, wv[3] = {1, 1, 1}, mean[2];
        pmc_samples *s = nullptr;
        EXPECT(pmc_samples_upload(c, xs, 3, 2, &s) == PMC_OK);
        EXPECT(pmc_weighted_moments(c, s, wv, 0, mean, nullptr) == PMC_OK);
        EXPECT(pmc_samples_free(s) == PMC_OK && pmc_shutdown(c) == PMC_OK);
    }
    // two contexts in two threads, and a third thread changing the process-wide options meanwhile
    {
        bool stop = false;
        std::thread a([] { handle_layer(0, 6, 5, 3000, 21); }), b([] { handle_layer(0, 7, 12, 2500, 22); });
        std::thread c([&stop] {
            for (int i = 0; i < 2000 && !stop; ++i) pmc_configure("stats_common_shift_limit", i % 2 ? 1000.0 : 0.0);
        });
        a.join();
        b.join();
        stop = true;
        c.join();
        pmc_configure("stats_common_shift_limit", 1000.0);
    }
    // the one-shot exchange: two "ranks" in this process (the stand-in's IPC handle carries the pointer), through the
    // kernel-level entry points and through two contexts
    {
        // (kernels do not run here: the connect-time self-test cannot pass -- first that it fails CLOSED, then the rest
        //  of the bookkeeping with the self-test switched off)
        {
            pmc_p2p *a = nullptr, *b = nullptr;
            unsigned char h[2 * PMC_P2P_HANDLE_BYTES];
            EXPECT(pmc_p2p_create(0, 2, 1000, 0, &a) == PMC_OK && pmc_p2p_create(1, 2, 1000, 0, &b) == PMC_OK);
            EXPECT(pmc_p2p_handle(a, h) == PMC_OK && pmc_p2p_handle(b, h + PMC_P2P_HANDLE_BYTES) == PMC_OK);
            EXPECT(pmc_p2p_connect(a, h) == PMC_EHIP && std::strstr(pmc_last_error(), "self-test") != nullptr);
            Dev v(8 * 1000);
            EXPECT(pmc_p2p_allreduce_sum(a, v.d(), 10, nullptr) == PMC_EINVAL);   // not connected: nothing half-open
            char info[200];
            EXPECT(pmc_p2p_info(a, info, sizeof(info)) == PMC_OK && std::strstr(info, "selftest=failed") && std::strstr(info, "memory=finegrained"));
            unsigned char junk[2 * PMC_P2P_HANDLE_BYTES];
            std::memset(junk, 7, sizeof(junk));
            EXPECT(pmc_p2p_connect(b, junk) == PMC_EHIP && std::strstr(pmc_last_error(), "not a handle") != nullptr);
            EXPECT(pmc_p2p_destroy(a) == PMC_OK && pmc_p2p_destroy(b) == PMC_OK);
            pmc_ctx *c0 = nullptr;
            EXPECT(pmc_init(0, &c0) == PMC_OK);
            EXPECT(pmc_ctx_p2p_open(c0, 0, 2, 5000, h) == PMC_OK);
            std::memcpy(h + PMC_P2P_HANDLE_BYTES, h, PMC_P2P_HANDLE_BYTES);
            EXPECT(pmc_ctx_p2p_connect(c0, h) == PMC_EHIP);                       // self-test fails: the context has no exchange left
            EXPECT(pmc_ctx_p2p_open(c0, 0, 2, 5000, h) == PMC_OK);                // ... and can open a new one
            EXPECT(pmc_shutdown(c0) == PMC_OK);
            setenv("PMC_P2P_MEMORY", "nonsense", 1);
            EXPECT(pmc_p2p_create(0, 2, 1000, 0, &a) == PMC_EINVAL);
            setenv("PMC_P2P_MEMORY", "coarse", 1);
            EXPECT(pmc_p2p_create(0, 1, 1000, 0, &a) == PMC_OK && pmc_p2p_info(a, info, sizeof(info)) == PMC_OK && std::strstr(info, "memory=coarse"));
            EXPECT(pmc_p2p_destroy(a) == PMC_OK);
            unsetenv("PMC_P2P_MEMORY");
        }
        setenv("PMC_P2P_SELFTEST", "0", 1);
        pmc_p2p *a = nullptr, *b = nullptr;
        unsigned char h[2 * PMC_P2P_HANDLE_BYTES];
        EXPECT(pmc_p2p_create(0, 2, 1000, 0, &a) == PMC_OK && pmc_p2p_create(1, 2, 1000, 0, &b) == PMC_OK);
        EXPECT(pmc_p2p_create(2, 2, 1000, 0, &b) == PMC_EINVAL && pmc_p2p_create(0, 17, 1000, 0, &b) == PMC_EINVAL);
        EXPECT(pmc_p2p_handle(a, h) == PMC_OK && pmc_p2p_handle(b, h + PMC_P2P_HANDLE_BYTES) == PMC_OK);
        Dev v(8 * 1000);
        EXPECT(pmc_p2p_allreduce_sum(a, v.d(), 10, nullptr) == PMC_EINVAL);       // not connected
        EXPECT(pmc_p2p_connect(a, h) == PMC_OK && pmc_p2p_connect(b, h) == PMC_OK && pmc_p2p_connect(b, h) == PMC_EINVAL);
        EXPECT(pmc_p2p_allreduce_sum(a, v.d(), 1000, nullptr) == PMC_OK && pmc_p2p_allreduce_sum(b, v.d(), 1000, nullptr) == PMC_OK);
        EXPECT(pmc_p2p_allreduce_sum(a, v.d(), 1025, nullptr) == PMC_EINVAL);
        EXPECT(pmc_p2p_status(a, nullptr) == PMC_OK);
        EXPECT(pmc_p2p_destroy(a) == PMC_OK && pmc_p2p_destroy(b) == PMC_OK && pmc_p2p_destroy(nullptr) == PMC_OK);
        pmc_ctx *c0 = nullptr, *c1 = nullptr;
        EXPECT(pmc_init(0, &c0) == PMC_OK && pmc_init(0, &c1) == PMC_OK);
        EXPECT(pmc_ctx_p2p_connect(c0, h) == PMC_EINVAL);
        EXPECT(pmc_ctx_p2p_open(c0, 0, 2, 5000, h) == PMC_OK && pmc_ctx_p2p_open(c1, 1, 2, 5000, h + PMC_P2P_HANDLE_BYTES) == PMC_OK);
        EXPECT(pmc_ctx_p2p_open(c0, 0, 2, 5000, h) == PMC_EINVAL);
        EXPECT(pmc_ctx_p2p_connect(c0, h) == PMC_OK && pmc_ctx_p2p_connect(c1, h) == PMC_OK);
        double xs[6] = {0, 1, 2, 3, 4, 5}, wv[3] = {1, 1, 1}, mean[2];
        pmc_samples *s0 = nullptr;
        EXPECT(pmc_samples_upload(c0, xs, 3, 2, &s0) == PMC_OK);
        EXPECT(pmc_weighted_moments(c0, s0, wv, 0, mean, nullptr) == PMC_OK);        // (its all-reduce goes through the mailboxes)
        EXPECT(pmc_samples_free(s0) == PMC_OK && pmc_shutdown(c0) == PMC_OK && pmc_shutdown(c1) == PMC_OK);
    }
    EXPECT(pmc_stub_leaked_streams() == 0);
    if (g_fail) {
        std::fprintf(stderr, "host_checks: %d check(s) failed\n", g_fail);
        return 1;
    }
    std::printf("host_checks: ok\n");
    return 0;
}
