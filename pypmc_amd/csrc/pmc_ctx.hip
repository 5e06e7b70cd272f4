// pmc_ctx.hip -- the handle layer (include/pmc_ctx.h): host pointers in, host pointers out, the reference's
// conventions.  Host-side C++ over the kernel-level ABI of include/pmc_hip.h: it owns a stream, scratch buffers and
// (optionally) the RCCL communicator, builds the parameter packs, runs the same entry points the Python front-end
// runs, all-reduces the K-sized buffer and converts the shifted, un-normalised sums into N_comp / x_mean_comp / S
// (variational.pyx:699-932) or alpha / mu / sigma / the dof constant (pmc.pyx:188-222, :602-696).  No kernels here.
#include "../../include/pmc_ctx.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <functional>
#include <initializer_list>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" int pmc_internal_fail(int code, const char *msg);       // pmc_api.hip: sets pmc_last_error()
// (pmc_api.hip, not in the public headers) a context's own copy of the library options, installed for the calling thread
// while one of its calls runs; its launches' timing records
extern "C" void *pmc_internal_tuning_new(void);
extern "C" void pmc_internal_tuning_free(void *);
extern "C" int pmc_internal_tuning_set(void *, const char *key, double value);
extern "C" void pmc_internal_tuning_use(const void *);
extern "C" int pmc_internal_timing_stream(void *stream, int on);
extern "C" int pmc_internal_get_timings(void *stream, pmc_timing *h_out, int max_entries, int *n_entries);
// (pmc_p2p.hip) out[i] = ((slot_0[i] + slot_1[i]) + slot_2[i]) + ...: the sum over a context's devices, in device order
extern "C" int pmc_internal_ordered_sum(const double *d_slots, int nslots, int64_t stride, int64_t n, double *d_out, void *stream);

namespace {

constexpr int NSC = 8;                                              // scalars in front of the statistics
constexpr double TINY = 2.2250738585072014e-308;                    // numpy.finfo('d').tiny (_regularize.pyx:6-17)
constexpr int MAX_PARTS = 64;

int failf(int code, const char *fmt, ...)
{
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return pmc_internal_fail(code, buf);
}
int hipf(hipError_t e, const char *what) { return failf(PMC_EHIP, "%s: %s", what, hipGetErrorString(e)); }

#define CK(call)                   \
    do {                           \
        const int rc_ = (call);    \
        if (rc_ < 0) return rc_;   \
    } while (0)
#define HK(call, what)                            \
    do {                                          \
        const hipError_t e_ = (call);             \
        if (e_ != hipSuccess) return hipf(e_, what); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return PMC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        HK(hipMalloc(&p, want), "hipMalloc");
        cap = want;
        return PMC_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    double *d() const { return (double *)p; }
};

// One device's share of a context: its stream and scratch.  A context over several devices (pmc_init_devices) gives
// every part a host thread of its own that issues that part's copies and launches ("one host thread per
// device"): a single thread issuing to eight devices would start the last one ~0.4 ms after the first, a quarter of an
// E-step at the 8-way shard size.  The same device may appear more than once (virtual shards, each with its own stream
// and scratch): that is how the sharded path is tested on a one-GPU box.
struct Part {
    int index = 0, device = 0;
    hipStream_t stream = nullptr;
    DevBuf ws, u, scratch, flat, pack, spack, aux, nk1, nk2, lat, params, result;
    // the worker (multi-part contexts only)
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    const std::function<int(Part &)> *job = nullptr;
    bool pending = false, quit = false;
    int rc = PMC_OK;
    std::string err;
    const void *tuning = nullptr;
    // pinned staging for the small copies of a call (h2d / d2h below)
    char *stage = nullptr;
    size_t stage_off = 0;
    bool stage_tried = false;
    void release_buffers()
    {
        for (DevBuf *b : {&ws, &u, &scratch, &flat, &pack, &spack, &aux, &nk1, &nk2, &lat, &params, &result}) b->release();
        if (stage) (void)hipHostFree(stage);
        stage = nullptr;
        stage_off = 0;
    }
};

void worker_main(Part *p)
{
    (void)hipSetDevice(p->device);                                  // (the current device is a per-thread setting)
    std::unique_lock<std::mutex> lk(p->m);
    for (;;) {
        p->cv.wait(lk, [p] { return p->pending || p->quit; });
        if (p->quit) return;
        pmc_internal_tuning_use(p->tuning);
        int rc = (*p->job)(*p);
        if (rc < 0) p->err = pmc_last_error();
        pmc_internal_tuning_use(nullptr);
        p->rc = rc;
        p->pending = false;
        p->cv.notify_all();
    }
}

}  // namespace

struct pmc_ctx {
    std::vector<Part *> parts;    // parts[0] holds the exchange with other ranks and the sum over this context's devices
    pmc_comm *comm = nullptr;
    pmc_p2p *p2p = nullptr;       // the one-shot exchange (pmc_ctx_p2p_open / _connect) instead of the RCCL communicator
    pmc_p2p *p2p_failed = nullptr;// an exchange whose connect failed: out of use, its mailbox alive until pmc_shutdown
    std::recursive_mutex mu;      // calls of one context are serialised here: any thread may call, one at a time
    void *tuning = nullptr;       // this context's copy of the library options (pmc_ctx_configure)
    DevBuf slots;                 // on parts[0]'s device: one statistics vector per part, summed in part order
    int log = 0;                  // PMC_HIP_LOG: one line per N-sized call on stderr (samples per second)
    int nparts() const { return (int)parts.size(); }
};
struct pmc_mix {
    pmc_ctx *ctx;
    int family, K, D;
    std::vector<double> w, mu, inv_sigma, log_norm, dof;
    std::vector<DevBuf> pack;                                       // per part: all K components, column k, weight w_k
};
struct pmc_samples {
    pmc_ctx *ctx;
    int64_t N;
    int D;
    std::vector<int64_t> begin;                                     // nparts + 1: part p holds rows [begin[p], begin[p + 1])
    std::vector<DevBuf> x, w, origin;                               // per part
    bool has_w, has_origin;
    bool borrowed = false;                                          // pmc_samples_wrap: x[0] is the caller's memory
    std::vector<DevBuf> sw;                                         // per part: resident sample weights of the VB E-step
    bool has_sw = false, sw_borrowed = false;
    int64_t n(int p) const { return begin[p + 1] - begin[p]; }
};

namespace {

// Scope of one call of a context: its mutex held, its options installed for this thread (the kernel-level entry points
// below read them instead of the process-wide ones)
struct CtxCall {
    pmc_ctx *c;
    explicit CtxCall(pmc_ctx *c_) : c(c_)
    {
        c->mu.lock();
        pmc_internal_tuning_use(c->tuning);
    }
    ~CtxCall()
    {
        pmc_internal_tuning_use(nullptr);
        c->mu.unlock();
    }
    CtxCall(const CtxCall &) = delete;
    CtxCall &operator=(const CtxCall &) = delete;
};

// one line per N-sized call when PMC_HIP_LOG is set (samples-per-second logging)
struct CallLog {
    const pmc_ctx *c;
    const char *what;
    int64_t N;
    int K, D;
    std::chrono::steady_clock::time_point t0;
    CallLog(const pmc_ctx *c_, const char *w, int64_t N_, int K_, int D_) : c(c_), what(w), N(N_), K(K_), D(D_)
    {
        if (c->log) t0 = std::chrono::steady_clock::now();
    }
    ~CallLog()
    {
        if (!c->log) return;
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "[pmc_hip] %s: N=%lld K=%d D=%d on %d device%s: %.3f ms, %.3e samples/s\n", what, (long long)N, K, D,
                     c->nparts(), c->nparts() == 1 ? "" : "s", 1e3 * s, s > 0 ? (double)N / s : 0.0);
    }
};

int use(const pmc_ctx *ctx)
{
    if (!ctx) return failf(PMC_EINVAL, "NULL context");
    HK(hipSetDevice(ctx->parts[0]->device), "hipSetDevice");
    return PMC_OK;
}

// fn(part) for every part of the context: inline for a one-device context, on the parts' own threads otherwise (the
// caller waits for all of them; the first failing part's status and message are the call's)
int for_parts(pmc_ctx *ctx, const std::function<int(Part &)> &fn)
{
    const int n = ctx->nparts();
    if (n == 1) return fn(*ctx->parts[0]);
    for (Part *p : ctx->parts) {
        std::lock_guard<std::mutex> lk(p->m);
        p->job = &fn;
        p->tuning = ctx->tuning;
        p->rc = PMC_OK;
        p->pending = true;
        p->cv.notify_all();
    }
    int rc = PMC_OK;
    for (Part *p : ctx->parts) {
        std::unique_lock<std::mutex> lk(p->m);
        p->cv.wait(lk, [p] { return !p->pending; });
        if (p->rc < 0 && rc == PMC_OK) rc = failf(p->rc, "device %d (part %d): %s", p->device, p->index, p->err.c_str());
    }
    return rc;
}

// Host <-> device copies of a part.  The K-sized parameters and results of a call (up to STAGE_SMALL bytes) go through a
// pinned buffer of the part: a copy out of pageable memory has to be waited for before the caller's array may go away --
// a stream synchronisation with the GPU idle behind it, in front of the kernels of EVERY call -- while a copy out of the
// staging buffer is just queued (the caller's array has been read by the memcpy), and the runtime does not stage it a
// second time.  The buffer is filled from the front; it starts over whenever the stream is known to be drained (behind
// the wait of a d2h, or the wait it takes itself when it is full).  Everything that consumes such a copy runs on the same
// stream; whoever hands results to another stream or device (publish) drains the stream itself.
constexpr size_t STAGE_BYTES = (size_t)4 << 20, STAGE_SMALL = (size_t)1 << 20;
bool stage_ready(Part &pt)
{
    if (!pt.stage && !pt.stage_tried) {
        pt.stage_tried = true;
        void *p = nullptr;
        if (hipHostMalloc(&p, STAGE_BYTES, hipHostMallocPortable) == hipSuccess) pt.stage = (char *)p;
        else (void)hipGetLastError();                               // (no pinned memory to be had: the plain copies below)
    }
    return pt.stage != nullptr;
}
int stage_room(Part &pt, size_t bytes)
{
    if (pt.stage_off + bytes > STAGE_BYTES) {
        HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");
        pt.stage_off = 0;
    }
    return PMC_OK;
}
int h2d(Part &pt, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return PMC_OK;
    if (bytes <= STAGE_SMALL && stage_ready(pt)) {
        CK(stage_room(pt, bytes));
        char *b = pt.stage + pt.stage_off;
        std::memcpy(b, src, bytes);
        HK(hipMemcpyAsync(dst, b, bytes, hipMemcpyHostToDevice, pt.stream), "hipMemcpyAsync (host to device)");
        pt.stage_off += (bytes + 255) & ~(size_t)255;
        return PMC_OK;
    }
    HK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pt.stream), "hipMemcpyAsync (host to device)");
    HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");     // the source may be pageable and short-lived
    pt.stage_off = 0;
    return PMC_OK;
}
int d2h(Part &pt, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return PMC_OK;
    if (bytes <= STAGE_SMALL && stage_ready(pt)) {
        CK(stage_room(pt, bytes));
        char *b = pt.stage + pt.stage_off;
        HK(hipMemcpyAsync(b, src, bytes, hipMemcpyDeviceToHost, pt.stream), "hipMemcpyAsync (device to host)");
        HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");
        std::memcpy(dst, b, bytes);
        pt.stage_off = 0;
        return PMC_OK;
    }
    HK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pt.stream), "hipMemcpyAsync (device to host)");
    HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");
    pt.stage_off = 0;
    return PMC_OK;
}
int workspace(Part &pt, int64_t N, int K, int D)
{
    const int64_t need = pmc_workspace_bytes(N > 0 ? N : 1, K, D);
    if (need < 0) return (int)need;
    return pt.ws.ensure((size_t)need);
}

// Before the parts run a call whose K-sized result is summed: room for one vector of n doubles per part on parts[0]'s
// device (a one-part context needs none: its vector is the sum)
int prepare_slots(pmc_ctx *ctx, size_t n)
{
    if (ctx->nparts() == 1) return PMC_OK;
    return ctx->slots.ensure(sizeof(double) * n * (size_t)ctx->nparts());
}
// End of a part's job: its vector (pt.flat, n doubles) into its slot on parts[0]'s device, and the part's stream drained
int publish(pmc_ctx *ctx, Part &pt, const double *d_vec, size_t n)
{
    if (ctx->nparts() == 1) return PMC_OK;
    Part &p0 = *ctx->parts[0];
    double *dst = ctx->slots.d() + (size_t)pt.index * n;
    if (pt.device == p0.device)
        HK(hipMemcpyAsync(dst, d_vec, sizeof(double) * n, hipMemcpyDeviceToDevice, pt.stream), "hipMemcpyAsync (slot)");
    else
        HK(hipMemcpyPeerAsync(dst, p0.device, d_vec, pt.device, sizeof(double) * n, pt.stream), "hipMemcpyPeerAsync (slot)");
    HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");
    return PMC_OK;
}
// After all parts have published: parts[0].flat[0 .. n) = the slots added in part order (on parts[0]'s stream), then the
// sum over the ranks of a sharded run.  The calling thread's device is parts[0]'s.
int reduce(pmc_ctx *ctx, double *d_out, size_t n)
{
    Part &p0 = *ctx->parts[0];
    if (ctx->nparts() > 1) CK(pmc_internal_ordered_sum(ctx->slots.d(), ctx->nparts(), (int64_t)n, (int64_t)n, d_out, p0.stream));
    if (ctx->p2p) return pmc_p2p_allreduce_sum(ctx->p2p, d_out, (int64_t)n, p0.stream);
    if (ctx->comm) return pmc_comm_allreduce_sum(ctx->comm, d_out, (int64_t)n, p0.stream);
    return PMC_OK;
}

// contiguous blocks of N rows over n parts (sizes differ by at most one; pypmc_amd.parallel.shard_bounds)
void split_rows(int64_t N, int n, std::vector<int64_t> &begin)
{
    begin.assign((size_t)n + 1, 0);
    const int64_t base = N / n, extra = N % n;
    for (int p = 0; p < n; ++p) begin[p + 1] = begin[p] + base + (p < extra ? 1 : 0);
}

// psi(x) for x > 0: recurrence up to x >= 10, then the asymptotic series (|error| < 1e-15 there)
double digamma(double x)
{
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 +
                     f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + std::log(x) - 0.5 / x + t;
}

// lower Cholesky factor of a symmetric positive definite D x D matrix (row-major), in place in `a`'s lower triangle
bool cholesky_lower(std::vector<double> &a, int D)
{
    for (int j = 0; j < D; ++j) {
        double s = a[(size_t)j * D + j];
        for (int k = 0; k < j; ++k) s -= a[(size_t)j * D + k] * a[(size_t)j * D + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double ljj = std::sqrt(s);
        a[(size_t)j * D + j] = ljj;
        for (int i = j + 1; i < D; ++i) {
            double t = a[(size_t)i * D + j];
            for (int k = 0; k < j; ++k) t -= a[(size_t)i * D + k] * a[(size_t)j * D + k];
            a[(size_t)i * D + j] = t / ljj;
        }
        for (int i = 0; i < j; ++i) a[(size_t)i * D + j] = 0.0;
    }
    return true;
}
// lower Cholesky factor of sigma = inv(P) from the precision P
bool chol_of_inverse(const double *P, int D, double *L)
{
    std::vector<double> g(P, P + (size_t)D * D);
    if (!cholesky_lower(g, D)) return false;                        // P = G G^T
    std::vector<double> gi((size_t)D * D, 0.0);                     // G^-1, lower
    for (int j = 0; j < D; ++j) {
        gi[(size_t)j * D + j] = 1.0 / g[(size_t)j * D + j];
        for (int i = j + 1; i < D; ++i) {
            double t = 0.0;
            for (int k = j; k < i; ++k) t -= g[(size_t)i * D + k] * gi[(size_t)k * D + j];
            gi[(size_t)i * D + j] = t / g[(size_t)i * D + i];
        }
    }
    std::vector<double> s((size_t)D * D, 0.0);                      // sigma = G^-T G^-1
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
            for (int k = i; k < D; ++k) t += gi[(size_t)k * D + i] * gi[(size_t)k * D + j];
            s[(size_t)i * D + j] = s[(size_t)j * D + i] = t;
        }
    if (!cholesky_lower(s, D)) return false;
    std::memcpy(L, s.data(), sizeof(double) * (size_t)D * D);
    return true;
}

// pack of the components `sel` of a mixture (column = position in the mixture)
int build_mix_pack(const pmc_mix *m, const std::vector<int> &sel, std::vector<double> &host)
{
    const int K = (int)sel.size(), D = m->D;
    const int64_t stride = pmc_pack_stride(D);
    if (stride < 0) return (int)stride;
    std::vector<double> mu((size_t)K * D), prec((size_t)K * D * D), c0(K), c1(K, 0.0), c2(K, 0.0), c3(K, 0.0), w(K);
    std::vector<int32_t> col(K);
    for (int i = 0; i < K; ++i) {
        const int k = sel[i];
        std::memcpy(&mu[(size_t)i * D], &m->mu[(size_t)k * D], sizeof(double) * D);
        std::memcpy(&prec[(size_t)i * D * D], &m->inv_sigma[(size_t)k * D * D], sizeof(double) * (size_t)D * D);
        c0[i] = m->log_norm[k];
        if (m->family == PMC_KIND_STUDENT_T) {
            c1[i] = -.5 * (m->dof[k] + D);                          // student_t.pyx:116
            c2[i] = 1. / m->dof[k];                                 // :117
            c3[i] = m->dof[k];
        }
        w[i] = m->w[k];
        col[i] = k;
    }
    host.assign((size_t)K * stride, 0.0);
    return pmc_pack_components(K, D, mu.data(), prec.data(), c0.data(), c1.data(), c2.data(), c3.data(), w.data(),
                               col.data(), host.data());
}

int load_mix(pmc_mix *m, const double *h_w, const double *h_mu, const double *h_inv_sigma, const double *h_log_norm,
             const double *h_dof)
{
    const int K = m->K, D = m->D;
    if (!h_w || !h_mu || !h_inv_sigma || !h_log_norm) return failf(PMC_EINVAL, "mixture: weights, means, inv_sigma and log_norm are required");
    if (m->family == PMC_KIND_STUDENT_T && !h_dof) return failf(PMC_EINVAL, "mixture: StudentT components need h_dof");
    m->w.assign(h_w, h_w + K);
    m->mu.assign(h_mu, h_mu + (size_t)K * D);
    m->inv_sigma.assign(h_inv_sigma, h_inv_sigma + (size_t)K * D * D);
    m->log_norm.assign(h_log_norm, h_log_norm + K);
    if (m->family == PMC_KIND_STUDENT_T) m->dof.assign(h_dof, h_dof + K);
    std::vector<int> all(K);
    for (int k = 0; k < K; ++k) all[k] = k;
    std::vector<double> host;
    CK(build_mix_pack(m, all, host));                               // once; every part gets a copy
    return for_parts(m->ctx, [&](Part &pt) -> int {
        DevBuf &pk = m->pack[pt.index];
        CK(pk.ensure(host.size() * sizeof(double)));
        return h2d(pt, pk.p, host.data(), host.size() * sizeof(double));
    });
}

// K x (1 + D + D(D+1)/2) statistics (pmc_sufficient_stats' layout) -> S0, M1, full symmetric M2
void split_stats(const double *body, int K, int D, std::vector<double> &S0, std::vector<double> &M1, std::vector<double> &M2)
{
    const int T = D * (D + 1) / 2, PS = 1 + D + T;
    S0.resize(K);
    M1.resize((size_t)K * D);
    M2.assign((size_t)K * D * D, 0.0);
    for (int k = 0; k < K; ++k) {
        const double *b = body + (size_t)k * PS;
        S0[k] = b[0];
        for (int i = 0; i < D; ++i) M1[(size_t)k * D + i] = b[1 + i];
        int t = 0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j <= i; ++j, ++t)
                M2[((size_t)k * D + i) * D + j] = M2[((size_t)k * D + j) * D + i] = b[1 + D + t];
    }
}

// pypmc_amd/mix_adapt/_stats.py::shift_is_far (limit 100): does some component's weighted mean lie more than 10 of
// its own standard deviations from the shift its one-pass moments were taken about?
bool shift_is_far(const std::vector<double> &S0, const std::vector<double> &M1, const std::vector<double> &M2, int K, int D)
{
    double total = 0.0;
    for (int k = 0; k < K; ++k)
        if (std::isfinite(S0[k])) total += S0[k];
    for (int k = 0; k < K; ++k) {
        if (!std::isfinite(S0[k]) || !(S0[k] > 1e-200) || !(S0[k] > 1e-6 * total)) continue;
        const double n = S0[k];
        for (int i = 0; i < D; ++i) {
            const double db = M1[(size_t)k * D + i] / n, dbar2 = db * db;
            const double raw = M2[((size_t)k * D + i) * D + i] / n;
            const double v = raw - dbar2, t = 1e-14 * raw;               // numpy.maximum(v, t): a NaN operand gives NaN,
            const double var = (v != v || t != t) ? v + t : (v > t ? v : t);   // and the comparison below is false then
            if (dbar2 > 100. * var) return true;
        }
    }
    return false;
}

// _stats.py::centred_moments: mean = shift + M1 / reg(n_mean); cov = (M2 - n_mean dbar dbar^T) / reg(n_cov)
void centred_moments(const double *n_mean_in, const double *n_cov_in, const std::vector<double> &M1,
                     const std::vector<double> &M2, const double *shift, int K, int D, double *mean, double *cov)
{
    std::vector<double> dbar(D);
    for (int k = 0; k < K; ++k) {
        const double nm = n_mean_in[k] == 0.0 ? TINY : n_mean_in[k];
        const double nc = n_cov_in[k] == 0.0 ? TINY : n_cov_in[k];
        for (int i = 0; i < D; ++i) {
            dbar[i] = M1[(size_t)k * D + i] / nm;
            mean[(size_t)k * D + i] = shift[(size_t)k * D + i] + dbar[i];
        }
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                const double prod = dbar[i] * dbar[j];
                cov[((size_t)k * D + i) * D + j] = (M2[((size_t)k * D + i) * D + j] - nm * prod) / nc;
            }
    }
}

// shifts of a second pass: the mean just found where a component holds weight at all
void new_shifts(const std::vector<double> &S0, const std::vector<double> &M1, int K, int D, std::vector<double> &shift)
{
    for (int k = 0; k < K; ++k) {
        if (!(S0[k] > 1e-200)) continue;
        const double n = S0[k] == 0.0 ? TINY : S0[k];
        for (int i = 0; i < D; ++i) shift[(size_t)k * D + i] += M1[(size_t)k * D + i] / n;
    }
}

// the K x D shifts of a statistics pass as a pack of means, built once per pass on the host
int means_pack_host(const std::vector<double> &shift, int K, int D, std::vector<double> &host)
{
    const int64_t stride = pmc_pack_stride(D);
    if (stride < 0) return (int)stride;
    host.resize((size_t)K * stride);
    return pmc_pack_means(K, D, shift.data(), host.data());
}
int upload_spack(Part &pt, const std::vector<double> &host)
{
    CK(pt.spack.ensure(host.size() * sizeof(double)));
    return h2d(pt, pt.spack.p, host.data(), host.size() * sizeof(double));
}

int parse_devices(const char *text, std::vector<int> &ids)
{
    ids.clear();
    const char *c = text;
    while (*c) {
        while (*c == ' ' || *c == ',') ++c;
        if (!*c) break;
        char *end = nullptr;
        const long v = std::strtol(c, &end, 10);
        if (end == c || v < 0) return failf(PMC_EINVAL, "PMC_HIP_DEVICES: cannot read \"%s\" (a comma separated list of device ordinals)", text);
        ids.push_back((int)v);
        c = end;
    }
    return PMC_OK;
}

void destroy_ctx(pmc_ctx *ctx)
{
    for (Part *p : ctx->parts) {
        if (p->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(p->m);
                p->quit = true;
                p->cv.notify_all();
            }
            p->th.join();
        }
        (void)hipSetDevice(p->device);
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        p->release_buffers();
        if (p->stream) {
            (void)pmc_internal_timing_stream(p->stream, 0);
            (void)pmc_stream_release(p->stream);                    // the library's per-stream scratch slot
            (void)hipStreamDestroy(p->stream);
        }
        delete p;
    }
    ctx->parts.clear();
    if (ctx->tuning) pmc_internal_tuning_free(ctx->tuning);
    delete ctx;
}

}  // namespace

extern "C" {

// ---- context ----------------------------------------------------------------------------------------------
int pmc_init_devices(int n_devices, const int *device_ids, pmc_ctx **out)
{
    if (!out || n_devices < 0 || (n_devices > 0 && !device_ids)) return failf(PMC_EINVAL, "pmc_init_devices: bad argument");
    const int have = pmc_device_count();
    if (have < 0) return have;
    std::vector<int> ids;
    if (n_devices > 0) {
        ids.assign(device_ids, device_ids + n_devices);
    } else if (const char *e = std::getenv("PMC_HIP_DEVICES")) {
        CK(parse_devices(e, ids));
    }
    if (ids.empty())
        for (int d = 0; d < have; ++d) ids.push_back(d);            // every visible device
    if (ids.empty()) return failf(PMC_ENODEVICE, "pmc_init_devices: no HIP device");
    if ((int)ids.size() > MAX_PARTS) return failf(PMC_EINVAL, "pmc_init_devices: at most %d parts", MAX_PARTS);
    for (int d : ids)
        if (d < 0 || d >= have) return failf(PMC_ENODEVICE, "pmc_init_devices: device %d of %d", d, have);
    pmc_ctx *ctx = new pmc_ctx();
    ctx->tuning = pmc_internal_tuning_new();
    if (const char *e = std::getenv("PMC_HIP_LOG")) ctx->log = std::atoi(e);
    for (size_t i = 0; i < ids.size(); ++i) {
        Part *p = new Part();
        p->index = (int)i;
        p->device = ids[i];
        ctx->parts.push_back(p);
        hipError_t e = hipSetDevice(p->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            destroy_ctx(ctx);
            return hipf(e, "pmc_init_devices: hipStreamCreateWithFlags");
        }
        // the part's vector goes to parts[0]'s device by a peer copy: directly over xGMI where the devices allow it
        // (staged through the host by the runtime where they do not)
        if (i > 0 && p->device != ids[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, p->device, ids[0]) == hipSuccess && can) {
                e = hipDeviceEnablePeerAccess(ids[0], 0);
                if (e != hipSuccess) (void)hipGetLastError();       // (enabled already: fine)
            }
        }
    }
    if (ids.size() > 1)
        for (Part *p : ctx->parts) p->th = std::thread(worker_main, p);
    (void)hipSetDevice(ids[0]);
    *out = ctx;
    return PMC_OK;
}

int pmc_init(int device, pmc_ctx **out)
{
    if (!out) return failf(PMC_EINVAL, "pmc_init: NULL output");
    const int n = pmc_device_count();
    if (n < 0) return n;
    if (device < 0 || device >= n) return failf(PMC_ENODEVICE, "pmc_init: device %d of %d", device, n);
    return pmc_init_devices(1, &device, out);
}

int pmc_ctx_device_count(const pmc_ctx *ctx) { return ctx ? ctx->nparts() : failf(PMC_EINVAL, "NULL context"); }

int pmc_ctx_devices(const pmc_ctx *ctx, int *h_device_ids, int max_ids)
{
    if (!ctx || !h_device_ids) return failf(PMC_EINVAL, "pmc_ctx_devices: bad argument");
    for (int p = 0; p < ctx->nparts() && p < max_ids; ++p) h_device_ids[p] = ctx->parts[p]->device;
    return ctx->nparts();
}

int pmc_ctx_join(pmc_ctx *ctx, int rank, int world, const void *h_id)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (ctx->comm) return failf(PMC_EINVAL, "pmc_ctx_join: the context has a communicator already");
    return pmc_comm_init(rank, world, h_id, ctx->parts[0]->device, &ctx->comm);
}

int pmc_ctx_p2p_open(pmc_ctx *ctx, int rank, int world, int64_t max_doubles, void *h_handle)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (ctx->comm || ctx->p2p) return failf(PMC_EINVAL, "pmc_ctx_p2p_open: the context has an exchange already");
    CK(pmc_p2p_create(rank, world, max_doubles, ctx->parts[0]->device, &ctx->p2p));
    return pmc_p2p_handle(ctx->p2p, h_handle);
}

int pmc_ctx_p2p_connect(pmc_ctx *ctx, const void *h_handles)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!ctx->p2p) return failf(PMC_EINVAL, "pmc_ctx_p2p_connect: call pmc_ctx_p2p_open first");
    const int rc = pmc_p2p_connect(ctx->p2p, h_handles);
    if (rc < 0) {
        // Mapping, peer access or the self-test failed: the exchange is taken out of use -- the caller joins an RCCL communicator
        // instead, on ALL ranks -- but its mailbox is NOT freed here (advice r5): a peer that mapped it may still be inside its
        // own self-test round, whose kernel writes into every peer's mailbox, and this layer has no barrier across ranks.  It
        // stays allocated until pmc_shutdown (or a second failure: then the older one has had a whole connect to drain).
        if (ctx->p2p_failed) (void)pmc_p2p_destroy(ctx->p2p_failed);
        ctx->p2p_failed = ctx->p2p;
        ctx->p2p = nullptr;
    }
    return rc;
}

static void vb_states_of_context_gone(pmc_ctx *ctx);                // (below, with the VB state)

int pmc_shutdown(pmc_ctx *ctx)
{
    if (!ctx) return PMC_OK;
    (void)hipSetDevice(ctx->parts[0]->device);
    int rc = PMC_OK;
    ctx->mu.lock();                                                 // (a call still running in another thread finishes first)
    vb_states_of_context_gone(ctx);
    if (ctx->comm) rc = pmc_comm_destroy(ctx->comm);
    if (ctx->p2p) (void)pmc_p2p_destroy(ctx->p2p);
    if (ctx->p2p_failed) (void)pmc_p2p_destroy(ctx->p2p_failed);
    ctx->slots.release();
    ctx->mu.unlock();
    destroy_ctx(ctx);
    return rc;
}

// ---- mixture ----------------------------------------------------------------------------------------------
int pmc_mixture_create(pmc_ctx *ctx, int family, int K, int D, const double *h_w, const double *h_mu,
                       const double *h_inv_sigma, const double *h_log_norm, const double *h_dof, pmc_mix **out)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!out || K < 1 || D < 1) return failf(PMC_EINVAL, "pmc_mixture_create: bad K / D / output");
    if (family != PMC_KIND_GAUSS && family != PMC_KIND_STUDENT_T)
        return failf(PMC_EINVAL, "pmc_mixture_create: family must be PMC_KIND_GAUSS or PMC_KIND_STUDENT_T (got %d)", family);
    if (pmc_padded_dim(D) < 0) return PMC_EINVAL;
    pmc_mix *m = new pmc_mix();
    m->ctx = ctx;
    m->family = family;
    m->K = K;
    m->D = D;
    m->pack.resize((size_t)ctx->nparts());
    const int rc = load_mix(m, h_w, h_mu, h_inv_sigma, h_log_norm, h_dof);
    if (rc < 0) {
        (void)pmc_mixture_destroy(m);
        return rc;
    }
    *out = m;
    return PMC_OK;
}

int pmc_mixture_update(pmc_mix *mix, const double *h_w, const double *h_mu, const double *h_inv_sigma,
                       const double *h_log_norm, const double *h_dof)
{
    if (!mix) return failf(PMC_EINVAL, "pmc_mixture_update: NULL mixture");
    CK(use(mix->ctx));
    CtxCall call_(mix->ctx);
    return load_mix(mix, h_w, h_mu, h_inv_sigma, h_log_norm, h_dof);
}

int pmc_mixture_destroy(pmc_mix *mix)
{
    if (!mix) return PMC_OK;
    CtxCall call_(mix->ctx);
    for (int p = 0; p < mix->ctx->nparts(); ++p) {
        (void)hipSetDevice(mix->ctx->parts[p]->device);
        mix->pack[p].release();
    }
    (void)hipSetDevice(mix->ctx->parts[0]->device);
    delete mix;
    return PMC_OK;
}

// ---- samples ----------------------------------------------------------------------------------------------
static pmc_samples *new_samples(pmc_ctx *ctx, int64_t N, int D)
{
    pmc_samples *s = new pmc_samples();
    s->ctx = ctx;
    s->N = N;
    s->D = D;
    s->has_w = s->has_origin = false;
    const size_t n = (size_t)ctx->nparts();
    split_rows(N, (int)n, s->begin);
    s->x.resize(n);
    s->w.resize(n);
    s->origin.resize(n);
    s->sw.resize(n);
    return s;
}

int pmc_samples_upload(pmc_ctx *ctx, const double *h_x, int64_t N, int D, pmc_samples **out)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!out || N < 0 || D < 1 || (N > 0 && !h_x)) return failf(PMC_EINVAL, "pmc_samples_upload: bad argument");
    if (pmc_padded_dim(D) < 0) return PMC_EINVAL;
    pmc_samples *s = new_samples(ctx, N, D);
    const int rc = for_parts(ctx, [&](Part &pt) -> int {
        const int64_t n = s->n(pt.index);
        CK(s->x[pt.index].ensure(sizeof(double) * (size_t)(n > 0 ? n : 1) * D));
        return h2d(pt, s->x[pt.index].p, h_x + (size_t)s->begin[pt.index] * D, sizeof(double) * (size_t)n * D);
    });
    if (rc < 0) {
        (void)pmc_samples_free(s);
        return rc;
    }
    *out = s;
    return PMC_OK;
}

int pmc_samples_wrap(pmc_ctx *ctx, const double *d_x, int64_t N, int D, pmc_samples **out)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!out || N < 0 || D < 1 || (N > 0 && !d_x)) return failf(PMC_EINVAL, "pmc_samples_wrap: bad argument");
    if (ctx->nparts() != 1) return failf(PMC_EINVAL, "pmc_samples_wrap: a context of one device only (the array lives on one)");
    if (pmc_padded_dim(D) < 0) return PMC_EINVAL;
    pmc_samples *s = new_samples(ctx, N, D);
    s->borrowed = true;
    s->x[0].p = const_cast<double *>(d_x);
    s->x[0].cap = sizeof(double) * (size_t)N * D;
    *out = s;
    return PMC_OK;
}

int pmc_samples_generate(pmc_ctx *ctx, const pmc_mix *mix, const double *h_chol, const int64_t *h_counts,
                         uint64_t seed, int64_t first_sample, pmc_samples **out)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!out || !mix || !h_counts || mix->ctx != ctx) return failf(PMC_EINVAL, "pmc_samples_generate: bad argument");
    const int K = mix->K, D = mix->D;
    std::vector<int64_t> off(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        if (h_counts[k] < 0) return failf(PMC_EINVAL, "pmc_samples_generate: negative count for component %d", k);
        off[k + 1] = off[k] + h_counts[k];
    }
    const int64_t N = off[K];
    std::vector<double> chol((size_t)K * D * D);
    if (h_chol) {
        std::memcpy(chol.data(), h_chol, sizeof(double) * chol.size());
    } else {
        for (int k = 0; k < K; ++k)
            if (!chol_of_inverse(&mix->inv_sigma[(size_t)k * D * D], D, &chol[(size_t)k * D * D]))
                return failf(PMC_ENOTPOSDEF, "pmc_samples_generate: inv_sigma of component %d is not positive definite", k);
    }
    pmc_samples *s = new_samples(ctx, N, D);
    s->has_origin = true;
    CallLog log_(ctx, "pmc_samples_generate", N, K, D);
    // The samples are ordered by component and numbered 0 ... N-1 (mixture.pyx:192-206); part p generates the rows
    // [begin_p, begin_p+1): the counts clipped to that range, the random stream counted from first_sample + begin_p --
    // the very numbers one device would have produced for those rows.
    const int rc = for_parts(ctx, [&](Part &pt) -> int {
        const int64_t b = s->begin[pt.index], e = s->begin[pt.index + 1], n = e - b;
        std::vector<int64_t> loc(K + 1);
        for (int k = 0; k <= K; ++k) loc[k] = (off[k] < b ? b : (off[k] > e ? e : off[k])) - b;
        // parameters: [mu K*D | chol K*D*D | dof K] doubles, then the K+1 offsets
        const size_t nd = (size_t)K * D + (size_t)K * D * D + K;
        CK(pt.aux.ensure(nd * sizeof(double) + (K + 1) * sizeof(int64_t)));
        double *d_mu = pt.aux.d(), *d_chol = d_mu + (size_t)K * D, *d_dof = d_chol + (size_t)K * D * D;
        int64_t *d_off = (int64_t *)(d_dof + K);
        CK(h2d(pt, d_mu, mix->mu.data(), sizeof(double) * (size_t)K * D));
        CK(h2d(pt, d_chol, chol.data(), sizeof(double) * chol.size()));
        if (mix->family == PMC_KIND_STUDENT_T) CK(h2d(pt, d_dof, mix->dof.data(), sizeof(double) * K));
        CK(h2d(pt, d_off, loc.data(), sizeof(int64_t) * (K + 1)));
        CK(s->x[pt.index].ensure(sizeof(double) * (size_t)(n > 0 ? n : 1) * D));
        CK(s->origin[pt.index].ensure(sizeof(int64_t) * (size_t)(n > 0 ? n : 1)));
        if (n > 0)
            CK(pmc_propose(d_mu, d_chol, mix->family == PMC_KIND_STUDENT_T ? d_dof : nullptr, d_off, K, D, n, first_sample + b,
                           seed, s->x[pt.index].d(), (int64_t *)s->origin[pt.index].p, pt.stream));
        HK(hipStreamSynchronize(pt.stream), "pmc_samples_generate: stream");
        return PMC_OK;
    });
    if (rc < 0) {
        (void)pmc_samples_free(s);
        return rc;
    }
    *out = s;
    return PMC_OK;
}

int pmc_samples_set_sample_weights(pmc_samples *s, const double *h_w)
{
    if (!s) return failf(PMC_EINVAL, "pmc_samples_set_sample_weights: NULL samples");
    pmc_ctx *ctx = s->ctx;
    CK(use(ctx));
    CtxCall call_(ctx);
    if (s->sw_borrowed)
        for (DevBuf &b : s->sw) b.p = nullptr, b.cap = 0;
    s->sw_borrowed = false;
    s->has_sw = false;
    if (!h_w) return PMC_OK;
    CK(for_parts(ctx, [&](Part &pt) -> int {
        const int64_t n = s->n(pt.index);
        CK(s->sw[pt.index].ensure(sizeof(double) * (size_t)(n > 0 ? n : 1)));
        return h2d(pt, s->sw[pt.index].p, h_w + s->begin[pt.index], sizeof(double) * (size_t)n);
    }));
    s->has_sw = true;
    return PMC_OK;
}

int pmc_samples_wrap_sample_weights(pmc_samples *s, const double *d_w)
{
    if (!s || !d_w) return failf(PMC_EINVAL, "pmc_samples_wrap_sample_weights: bad argument");
    if (s->ctx->nparts() != 1) return failf(PMC_EINVAL, "pmc_samples_wrap_sample_weights: a context of one device only");
    CtxCall call_(s->ctx);
    if (!s->sw_borrowed) s->sw[0].release();
    s->sw[0].p = const_cast<double *>(d_w);
    s->sw[0].cap = sizeof(double) * (size_t)s->N;
    s->sw_borrowed = s->has_sw = true;
    return PMC_OK;
}

int64_t pmc_samples_count(const pmc_samples *s) { return s ? s->N : (int64_t)failf(PMC_EINVAL, "NULL samples"); }

int pmc_samples_shard(const pmc_samples *s, int part, int64_t *begin, int64_t *count)
{
    if (!s || part < 0 || part >= s->ctx->nparts()) return failf(PMC_EINVAL, "pmc_samples_shard: bad argument");
    if (begin) *begin = s->begin[part];
    if (count) *count = s->n(part);
    return s->ctx->parts[part]->device;
}

int pmc_samples_download(const pmc_samples *s, double *h_x)
{
    if (!s || (!h_x && s->N > 0)) return failf(PMC_EINVAL, "pmc_samples_download: bad argument");
    CK(use(s->ctx));
    CtxCall call_(s->ctx);
    return for_parts(s->ctx, [&](Part &pt) -> int {
        return d2h(pt, h_x + (size_t)s->begin[pt.index] * s->D, s->x[pt.index].p, sizeof(double) * (size_t)s->n(pt.index) * s->D);
    });
}

int pmc_samples_origin(const pmc_samples *s, int64_t *h_origin)
{
    if (!s || (!h_origin && s->N > 0)) return failf(PMC_EINVAL, "pmc_samples_origin: bad argument");
    if (!s->has_origin) return failf(PMC_EINVAL, "pmc_samples_origin: these samples were uploaded, not generated");
    CK(use(s->ctx));
    CtxCall call_(s->ctx);
    return for_parts(s->ctx, [&](Part &pt) -> int {
        return d2h(pt, h_origin + s->begin[pt.index], s->origin[pt.index].p, sizeof(int64_t) * (size_t)s->n(pt.index));
    });
}

int pmc_samples_free(pmc_samples *s)
{
    if (!s) return PMC_OK;
    CtxCall call_(s->ctx);
    for (int p = 0; p < s->ctx->nparts(); ++p) {
        (void)hipSetDevice(s->ctx->parts[p]->device);
        if (s->borrowed) s->x[p].p = nullptr, s->x[p].cap = 0;      // (not ours to free)
        if (s->sw_borrowed) s->sw[p].p = nullptr, s->sw[p].cap = 0;
        s->sw[p].release();
        s->x[p].release();
        s->w[p].release();
        s->origin[p].release();
    }
    (void)hipSetDevice(s->ctx->parts[0]->device);
    delete s;
    return PMC_OK;
}

// ---- evaluation -------------------------------------------------------------------------------------------
int pmc_mix_logpdf(const pmc_mix *mix, const pmc_samples *s, double *h_out, double *h_individual)
{
    if (!mix || !s || mix->ctx != s->ctx || mix->D != s->D) return failf(PMC_EINVAL, "pmc_mix_logpdf: mixture and samples do not belong together");
    pmc_ctx *ctx = mix->ctx;
    CK(use(ctx));
    CtxCall call_(ctx);
    const int K = mix->K, D = mix->D;
    if (s->N == 0) return PMC_OK;
    CallLog log_(ctx, "pmc_mix_logpdf", s->N, K, D);
    return for_parts(ctx, [&](Part &pt) -> int {
        const int64_t N = s->n(pt.index), b = s->begin[pt.index];
        if (N == 0) return PMC_OK;
        CK(workspace(pt, N, K, D));
        CK(pt.nk1.ensure(sizeof(double) * (size_t)N * (h_individual ? K + 1 : 1)));
        double *d_out = pt.nk1.d(), *d_ind = h_individual ? d_out + N : nullptr;
        CK(pmc_mixture_logpdf(s->x[pt.index].d(), N, D, mix->pack[pt.index].d(), K, mix->family, 0, d_out, d_ind, K, nullptr, nullptr,
                              nullptr, nullptr, pt.ws.p, pt.stream));
        if (h_out) CK(d2h(pt, h_out + b, d_out, sizeof(double) * (size_t)N));
        if (h_individual) CK(d2h(pt, h_individual + (size_t)b * K, d_ind, sizeof(double) * (size_t)N * K));
        return PMC_OK;
    });
}

int pmc_mix_logpdf_components(const pmc_mix *mix, const pmc_samples *s, const int32_t *h_components, int ncomponents,
                              double *h_individual)
{
    if (!mix || !s || mix->ctx != s->ctx || mix->D != s->D) return failf(PMC_EINVAL, "pmc_mix_logpdf_components: mixture and samples do not belong together");
    if (!h_components || ncomponents < 1 || !h_individual) return failf(PMC_EINVAL, "pmc_mix_logpdf_components: components and h_individual are required");
    pmc_ctx *ctx = mix->ctx;
    CK(use(ctx));
    CtxCall call_(ctx);
    const int K = mix->K, D = mix->D, n = ncomponents;
    std::vector<int> sel(n);
    for (int i = 0; i < n; ++i) {
        if (h_components[i] < 0 || h_components[i] >= K) return failf(PMC_EINVAL, "pmc_mix_logpdf_components: component %d of %d", (int)h_components[i], K);
        sel[i] = h_components[i];
    }
    if (s->N == 0) return PMC_OK;
    // the listed components as a pack of their own whose output columns are 0 ... n-1: an N x n matrix comes back and
    // is scattered into the listed columns of the caller's N x K array (every other column stays as it is)
    pmc_mix view;
    view.ctx = ctx; view.family = mix->family; view.K = n; view.D = D;
    view.w.resize(n); view.mu.resize((size_t)n * D); view.inv_sigma.resize((size_t)n * D * D); view.log_norm.resize(n);
    if (mix->family == PMC_KIND_STUDENT_T) view.dof.resize(n);
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) {
        const int k = sel[i];
        all[i] = i;
        view.w[i] = mix->w[k];
        view.log_norm[i] = mix->log_norm[k];
        if (mix->family == PMC_KIND_STUDENT_T) view.dof[i] = mix->dof[k];
        std::memcpy(&view.mu[(size_t)i * D], &mix->mu[(size_t)k * D], sizeof(double) * D);
        std::memcpy(&view.inv_sigma[(size_t)i * D * D], &mix->inv_sigma[(size_t)k * D * D], sizeof(double) * (size_t)D * D);
    }
    std::vector<double> host;
    CK(build_mix_pack(&view, all, host));
    return for_parts(ctx, [&](Part &pt) -> int {
        const int64_t N = s->n(pt.index), b = s->begin[pt.index];
        if (N == 0) return PMC_OK;
        CK(pt.pack.ensure(host.size() * sizeof(double)));
        CK(h2d(pt, pt.pack.p, host.data(), host.size() * sizeof(double)));
        CK(workspace(pt, N, n, D));
        CK(pt.nk1.ensure(sizeof(double) * (size_t)N * n));
        CK(pmc_mixture_logpdf(s->x[pt.index].d(), N, D, pt.pack.d(), n, mix->family, 0, nullptr, pt.nk1.d(), n, nullptr, nullptr,
                              nullptr, nullptr, pt.ws.p, pt.stream));
        std::vector<double> cols((size_t)N * n);
        CK(d2h(pt, cols.data(), pt.nk1.p, sizeof(double) * cols.size()));
        for (int64_t r = 0; r < N; ++r)
            for (int i = 0; i < n; ++i) h_individual[(size_t)(b + r) * K + sel[i]] = cols[(size_t)r * n + i];
        return PMC_OK;
    });
}

int pmc_ctx_configure(pmc_ctx *ctx, const char *key, double value)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    return pmc_internal_tuning_set(ctx->tuning, key, value);
}

int pmc_ctx_timing_enable(pmc_ctx *ctx, int on)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    for (Part *p : ctx->parts) {
        HK(hipSetDevice(p->device), "hipSetDevice");
        CK(pmc_internal_timing_stream(p->stream, on));
    }
    return use(ctx);
}

int pmc_ctx_get_timings(pmc_ctx *ctx, pmc_timing *h_out, int max_entries, int *n_entries)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (ctx->nparts() == 1) return pmc_internal_get_timings(ctx->parts[0]->stream, h_out, max_entries, n_entries);
    // several devices: the entries of all parts merged by kernel name -- calls, flops and bytes added, ms = the LONGEST
    // part's (the parts run side by side: that is the time the call saw)
    if (!h_out || !n_entries || max_entries < 1) return failf(PMC_EINVAL, "pmc_ctx_get_timings: bad argument");
    std::vector<pmc_timing> all;
    for (Part *p : ctx->parts) {
        HK(hipSetDevice(p->device), "hipSetDevice");
        std::vector<pmc_timing> one(32);
        int n = 0;
        CK(pmc_internal_get_timings(p->stream, one.data(), 32, &n));
        for (int i = 0; i < n && i < 32; ++i) {
            size_t j = 0;
            while (j < all.size() && std::strcmp(all[j].name, one[i].name) != 0) ++j;
            if (j == all.size()) {
                all.push_back(one[i]);
            } else {
                all[j].calls += one[i].calls;
                all[j].flops += one[i].flops;
                all[j].bytes += one[i].bytes;
                if (one[i].ms > all[j].ms) all[j].ms = one[i].ms;
            }
        }
    }
    *n_entries = (int)all.size();
    for (size_t i = 0; i < all.size() && (int)i < max_entries; ++i) h_out[i] = all[i];
    return use(ctx);
}

int pmc_is_weights(const pmc_mix *q, pmc_samples *s, const double *h_log_target, const pmc_mix *target, double *h_w,
                   double *h_log_target_out, double *h_sums)
{
    if (!q || !s || q->ctx != s->ctx || q->D != s->D) return failf(PMC_EINVAL, "pmc_is_weights: proposal and samples do not belong together");
    if ((h_log_target != nullptr) == (target != nullptr)) return failf(PMC_EINVAL, "pmc_is_weights: give h_log_target or a target mixture (one of them)");
    if (target && (target->ctx != q->ctx || target->D != q->D)) return failf(PMC_EINVAL, "pmc_is_weights: target of another context / dimension");
    pmc_ctx *ctx = q->ctx;
    CK(use(ctx));
    CtxCall call_(ctx);
    const int D = q->D, Kmax = target && target->K > q->K ? target->K : q->K;
    CallLog log_(ctx, "pmc_is_weights", s->N, q->K, D);
    CK(prepare_slots(ctx, NSC));
    CK(for_parts(ctx, [&](Part &pt) -> int {
        const int i = pt.index;
        const int64_t N = s->n(i), b = s->begin[i];
        CK(pt.flat.ensure(sizeof(double) * NSC));
        double *d_sc = pt.flat.d();
        HK(hipMemsetAsync(d_sc, 0, sizeof(double) * NSC, pt.stream), "hipMemsetAsync");
        if (N > 0) {
            CK(workspace(pt, N, Kmax, D));
            CK(s->w[i].ensure(sizeof(double) * (size_t)N));
            CK(pt.nk1.ensure(sizeof(double) * (size_t)N));
            if (target) {
                CK(pmc_importance_weights(s->x[i].d(), N, D, q->pack[i].d(), q->K, q->family, target->pack[i].d(), target->K,
                                          target->family, nullptr, h_log_target_out ? pt.nk1.d() : nullptr, s->w[i].d(), nullptr,
                                          d_sc, pt.ws.p, pt.stream));
                if (h_log_target_out) CK(d2h(pt, h_log_target_out + b, pt.nk1.p, sizeof(double) * (size_t)N));
            } else {
                CK(h2d(pt, pt.nk1.p, h_log_target + b, sizeof(double) * (size_t)N));
                CK(pmc_mixture_logpdf(s->x[i].d(), N, D, q->pack[i].d(), q->K, q->family, 0, nullptr, nullptr, q->K, pt.nk1.d(),
                                      s->w[i].d(), nullptr, d_sc, pt.ws.p, pt.stream));
            }
            if (h_w) CK(d2h(pt, h_w + b, s->w[i].p, sizeof(double) * (size_t)N));
        }
        return publish(ctx, pt, d_sc, NSC);
    }));
    if (s->N > 0) s->has_w = true;
    Part &p0 = *ctx->parts[0];
    CK(reduce(ctx, p0.flat.d(), NSC));
    double sc[NSC];
    CK(d2h(p0, sc, p0.flat.d(), sizeof(sc)));
    if (ctx->p2p) CK(pmc_p2p_status(ctx->p2p, p0.stream));
    if (sc[4] != 0.0) return failf(PMC_EINVAL, "pmc_is_weights: %g importance weights overflowed (math range error, importance_sampling.py:207)", sc[4]);
    if (h_sums) {
        h_sums[0] = sc[0];
        h_sums[1] = sc[1];
        h_sums[2] = sc[2];
    }
    return PMC_OK;
}

// ---- VB E-step --------------------------------------------------------------------------------------------
int pmc_vb_estep(pmc_ctx *ctx, const pmc_samples *s, const double *h_sample_w, int K, const double *h_m, const double *h_W,
                 const double *h_nu, const double *h_beta, const double *h_ln_pi, const double *h_ln_lambda,
                 const double *h_shift, double *h_N_k, double *h_xbar, double *h_S, double *h_elogqz, double *h_r,
                 double *h_log_rho)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!s || s->ctx != ctx || K < 1 || !h_m || !h_W || !h_nu || !h_beta || !h_ln_pi || !h_ln_lambda || !h_N_k || !h_xbar || !h_S)
        return failf(PMC_EINVAL, "pmc_vb_estep: bad argument");
    const int D = s->D;
    const int64_t stride = pmc_pack_stride(D), PS = pmc_stats_stride(D);
    if (stride < 0) return (int)stride;
    CallLog log_(ctx, "pmc_vb_estep", s->N, K, D);
    // Compiled dimensions: NOTHING K-sized is computed on the host between the caller's arrays and the results (round 5,
    // verdict r4 #3) -- the raw parameters go up in one copy, the pack (K Cholesky factorisations), the shift pack and the
    // conversion to N_comp / x_mean_comp / S are kernels on the parts' streams (pmc_hip.h: same operations in the same
    // order as the host functions, same bits), and one copy brings the results back.  Beyond (D > 64): the host builds
    // the pack and converts, as before.
    const bool on_device = D <= pmc_max_compiled_dim() && std::getenv("PMC_CTX_HOST_PACKS") == nullptr;
    const size_t KD = (size_t)K * D, KDD = KD * D;
    // the posterior's pack (enum pmc_kind, PMC_KIND_VB): c0 = D / beta, c1 = nu, c2 = E[ln pi], c3 = E[ln|Lambda|] - D ln 2 pi
    // staging: [m | W | c0 | nu | ln_pi | c3 | shift]
    std::vector<double> stage(KD + KDD + 4 * (size_t)K + KD), host, shost;
    double *g_m = stage.data(), *g_W = g_m + KD, *g_c0 = g_W + KDD, *g_nu = g_c0 + K, *g_lp = g_nu + K, *g_c3 = g_lp + K,
           *g_shift = g_c3 + K;
    std::memcpy(g_m, h_m, sizeof(double) * KD);
    std::memcpy(g_W, h_W, sizeof(double) * KDD);
    std::memcpy(g_nu, h_nu, sizeof(double) * K);
    std::memcpy(g_lp, h_ln_pi, sizeof(double) * K);
    const double dl2pi = D * std::log(2. * 3.14159265358979323846);
    for (int k = 0; k < K; ++k) {
        g_c0[k] = D / h_beta[k];
        g_c3[k] = h_ln_lambda[k] - dl2pi;
    }
    std::memcpy(g_shift, h_shift ? h_shift : h_m, sizeof(double) * KD);
    if (!on_device) {
        host.resize((size_t)K * stride);
        CK(pmc_pack_components(K, D, h_m, h_W, g_c0, h_nu, h_ln_pi, g_c3, nullptr, nullptr, host.data()));
    }
    // results of the device path: [S0 K | M1 K D | mean K D | cov K D D | far K | scalars NSC] (pmc_convert_stats_device) and,
    // behind them, the pack builder's status [pivot K | value K]: one copy to the host
    const size_t nflat = NSC + (size_t)K * PS, nconv = (size_t)pmc_convert_stats_len(K, D), nres = nconv + 2 * (size_t)K;
    CK(prepare_slots(ctx, nflat));
    std::vector<double> shift(g_shift, g_shift + KD), flat(nflat), res(nres), S0, M1, M2;
    const bool want_nk = (h_r || h_log_rho) && s->N > 0;
    Part &p0 = *ctx->parts[0];
    if (on_device) CK(p0.result.ensure(sizeof(double) * nres));
    bool far = false;
    for (int pass = 0; pass < 2; ++pass) {
        const bool shifted = pass == 1 || h_shift != nullptr;
        if (shifted && !on_device) CK(means_pack_host(shift, K, D, shost));
        CK(for_parts(ctx, [&](Part &pt) -> int {
            const int i = pt.index;
            const int64_t N = s->n(i), b = s->begin[i];
            if (pass == 0) {
                CK(pt.pack.ensure((size_t)K * stride * sizeof(double)));
                CK(pt.flat.ensure(sizeof(double) * nflat));
                if (on_device) {
                    CK(pt.params.ensure(sizeof(double) * (stage.size() + 2 * (size_t)K)));
                    CK(h2d(pt, pt.params.p, stage.data(), sizeof(double) * stage.size()));
                    double *d = pt.params.d();
                    double *d_status = (&pt == &p0) ? p0.result.d() + nconv : d + stage.size();
                    if (shifted) CK(pt.spack.ensure((size_t)K * stride * sizeof(double)));
                    // the pack -- and, in the same launch, the pack of the caller's shifts
                    CK(pmc_pack_components_device(K, D, d, d + KD, d + KD + KDD, d + KD + KDD + K, d + KD + KDD + 2 * (size_t)K,
                                                  d + KD + KDD + 3 * (size_t)K, nullptr, nullptr, pt.pack.d(), d_status,
                                                  shifted ? d + KD + KDD + 4 * (size_t)K : nullptr, shifted ? pt.spack.d() : nullptr,
                                                  pt.stream));
                } else {
                    CK(h2d(pt, pt.pack.p, host.data(), host.size() * sizeof(double)));
                }
                CK(workspace(pt, N, K, D));
                CK(pt.u.ensure(sizeof(double) * (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, K)));
                if (h_sample_w && N > 0) {
                    CK(pt.aux.ensure(sizeof(double) * (size_t)N));
                    CK(h2d(pt, pt.aux.p, h_sample_w + b, sizeof(double) * (size_t)N));
                }
            }
            const double *d_sw = N > 0 ? (h_sample_w ? pt.aux.d() : (s->has_sw ? s->sw[i].d() : nullptr)) : nullptr;
            double *d_flat = pt.flat.d();
            if (shifted) {
                if (on_device) {
                    if (pass == 1) {                                   // (pass 0: built with the pack, above)
                        double *d_shift = pt.params.d() + KD + KDD + 4 * (size_t)K;
                        CK(h2d(pt, d_shift, shift.data(), sizeof(double) * KD));
                        CK(pt.spack.ensure((size_t)K * stride * sizeof(double)));
                        CK(pmc_pack_means_device(K, D, d_shift, pt.spack.d(), pt.stream));
                    }
                } else {
                    CK(upload_spack(pt, shost));
                }
            }
            const double *d_spack = shifted ? pt.spack.d() : nullptr;
            if (want_nk && pass == 0 && N > 0) {
                CK(pt.nk1.ensure(sizeof(double) * (size_t)N * K));
                CK(pt.nk2.ensure(sizeof(double) * (size_t)N * K));
                CK(pmc_responsibilities(s->x[i].d(), N, D, pt.pack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, d_sw, nullptr, pt.u.d(),
                                        nullptr, nullptr, h_r ? pt.nk1.d() : nullptr, h_log_rho ? pt.nk2.d() : nullptr, nullptr,
                                        K, d_flat, pt.ws.p, pt.stream));
                CK(pmc_sufficient_stats(s->x[i].d(), N, D, d_spack ? d_spack : pt.pack.d(), K, pt.u.d(), d_flat + NSC, pt.ws.p,
                                        pt.stream));
                if (h_r) CK(d2h(pt, h_r + (size_t)b * K, pt.nk1.p, sizeof(double) * (size_t)N * K));
                if (h_log_rho) CK(d2h(pt, h_log_rho + (size_t)b * K, pt.nk2.p, sizeof(double) * (size_t)N * K));
            } else {
                CK(pmc_estep_about(s->x[i].d(), N, D, pt.pack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, d_sw, nullptr, pt.u.d(), nullptr,
                                   nullptr, d_flat + NSC, d_flat, pt.ws.p, d_spack, pt.stream));
            }
            return publish(ctx, pt, d_flat, nflat);
        }));
        CK(reduce(ctx, p0.flat.d(), nflat));
        if (on_device) {
            // conversion on the first device; [converted | scalars | pack status] come back in ONE copy
            double *d_res = p0.result.d();
            CK(pmc_convert_stats_device(K, D, p0.flat.d() + NSC, p0.params.d() + KD + KDD + 4 * (size_t)K, nullptr, p0.flat.d(),
                                        d_res, p0.stream));
            CK(d2h(p0, res.data(), d_res, sizeof(double) * nres));
            if (ctx->p2p) CK(pmc_p2p_status(ctx->p2p, p0.stream));
            if (pass == 0) CK(pmc_pack_status(K, res.data() + nconv));
            far = false;
            for (int k = 0; k < K; ++k) far = far || res[nconv - NSC - (size_t)K + k] != 0.0;
            if (pass == 1 || !far) break;
            S0.assign(res.data(), res.data() + K);
            M1.assign(res.data() + K, res.data() + K + KD);
        } else {
            CK(d2h(p0, flat.data(), p0.flat.d(), sizeof(double) * nflat));
            if (ctx->p2p) CK(pmc_p2p_status(ctx->p2p, p0.stream));
            split_stats(flat.data() + NSC, K, D, S0, M1, M2);
            if (pass == 1 || !shift_is_far(S0, M1, M2, K, D)) break;
        }
        new_shifts(S0, M1, K, D, shift);                              // second pass about the mean just found
    }
    if (on_device) {
        const double *r_S0 = res.data(), *r_mean = r_S0 + K + KD, *r_cov = r_mean + KD;
        for (int k = 0; k < K; ++k) h_N_k[k] = r_S0[k] == 0.0 ? TINY : r_S0[k];   // variational.pyx:699-709
        std::memcpy(h_xbar, r_mean, sizeof(double) * KD);
        std::memcpy(h_S, r_cov, sizeof(double) * KDD);
        if (h_elogqz) *h_elogqz = res[nconv - NSC];
        return PMC_OK;
    }
    for (int k = 0; k < K; ++k) h_N_k[k] = S0[k] == 0.0 ? TINY : S0[k];   // variational.pyx:699-709
    centred_moments(h_N_k, h_N_k, M1, M2, shift.data(), K, D, h_xbar, h_S);
    if (h_elogqz) *h_elogqz = flat[0];
    return PMC_OK;
}

// ---- a VB fit whose K-sized state stays on the device (round 6, verdict r5 #6) ------------------------------
// GaussianInference.update / .likelihood_bound (variational.pyx:571-578, :194-209) without K x D x D arrays crossing the
// bus: the prior, the posterior and the latest sums live in ONE allocation on the context's device, pmc_vb_state_step
// strings the kernels of pmc_hip.h ("the K-sized half of a variational-Bayes iteration") and the E-step's together on the
// part's stream and brings back one small block.
struct pmc_vb_state {
    pmc_ctx *ctx;
    int K, D;
    DevBuf buf, pack, spack;
    double *field[PMC_VB_NFIELDS];
    pmc_vb_fields f;
    double *c0, *c3, *log_q_Z, *result, *conv, *terms, *shift2, *shift_next, *psi_parts, *bcast;
    size_t nresult, nconv;
    bool stepped;                                                   // an E-step has run: the E_* fields mean something
};

// The states that are alive, by address: a front-end's garbage collector may finalise a state AFTER the context it lived on was
// shut down (weak references to unreachable objects are cleared before their finalisers run, so the owner cannot close them
// first).  pmc_shutdown destroys the states of its context; every entry point checks that its handle is still here.
static std::mutex g_vb_states_mu;
static std::vector<pmc_vb_state *> g_vb_states;
static bool vb_state_alive(const pmc_vb_state *st)
{
    std::lock_guard<std::mutex> lk(g_vb_states_mu);
    for (const pmc_vb_state *q : g_vb_states)
        if (q == st) return true;
    return false;
}
static void vb_states_of_context_gone(pmc_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_vb_states_mu);
    for (size_t i = 0; i < g_vb_states.size();) {
        pmc_vb_state *st = g_vb_states[i];
        if (st->ctx != ctx) {
            ++i;
            continue;
        }
        st->buf.release();
        st->pack.release();
        st->spack.release();
        delete st;
        g_vb_states.erase(g_vb_states.begin() + (long)i);
    }
}

static size_t vb_field_len(int field, int K, int D)
{
    switch (field) {
    case PMC_VB_M0: case PMC_VB_M: case PMC_VB_X_MEAN: case PMC_VB_SHIFT_PREV: case PMC_VB_E_M:
        return (size_t)K * D;
    case PMC_VB_INV_W0: case PMC_VB_W: case PMC_VB_S: case PMC_VB_E_W:
        return (size_t)K * D * D;
    default:
        return (size_t)K;
    }
}

// one pass of the E-step from the resident state: the pack (and the shifts' pack) -> responsibilities + statistics ->
// the sum over ranks -> conversion -> the state's sums and the small block
// Several devices (a context of pmc_init_devices): the state lives on the first; what the other devices' pack builders need --
// [m | W | beta | nu | E ln pi | E ln|Lambda|] | ln|W| | the psi parts | the shifts -- is gathered into one block there and
// peer-copied to each of them (K x D x D numbers per device and E-step, over the links the statistics come back on); every
// device builds the same pack, runs its shard, publishes its sums; the first adds them in device order and converts.
static int vb_state_pass_parts(pmc_vb_state *st, const pmc_samples *s, const double *d_shift, bool with_pack, bool own_psi)
{
    pmc_ctx *ctx = st->ctx;
    Part &p0 = *ctx->parts[0];
    const int K = st->K, D = st->D;
    const int64_t stride = pmc_pack_stride(D), PS = pmc_stats_stride(D);
    const size_t nflat = NSC + (size_t)K * PS, KD = (size_t)K * D, KDD = KD * D;
    if (own_psi && with_pack)
        return failf(PMC_EINVAL, "pmc_vb_state_step: the device's own psi is not available over several devices (hand in h_psi_parts)");
    // the block: [six fields KD + KDD + 4K | ln|W| K | psi parts 2K | shifts KD]
    const size_t o_ld = KD + KDD + 4 * (size_t)K, o_parts = o_ld + K, o_shift = o_parts + 2 * (size_t)K, nblock = o_shift + KD;
    double *blk = st->bcast;
    HK(hipMemcpyAsync(blk, st->field[PMC_VB_M], sizeof(double) * o_ld, hipMemcpyDeviceToDevice, p0.stream), "hipMemcpyAsync (block)");
    HK(hipMemcpyAsync(blk + o_ld, st->f.log_det_W, sizeof(double) * K, hipMemcpyDeviceToDevice, p0.stream), "hipMemcpyAsync (block)");
    HK(hipMemcpyAsync(blk + o_parts, st->psi_parts, sizeof(double) * 2 * (size_t)K, hipMemcpyDeviceToDevice, p0.stream), "hipMemcpyAsync (block)");
    if (d_shift) HK(hipMemcpyAsync(blk + o_shift, d_shift, sizeof(double) * KD, hipMemcpyDeviceToDevice, p0.stream), "hipMemcpyAsync (block)");
    HK(hipStreamSynchronize(p0.stream), "hipStreamSynchronize");     // (the M-step and the fields the host put are behind this)
    CK(prepare_slots(ctx, nflat));
    double *d_pstatus = st->result + (size_t)pmc_vb_small_len(K);
    CK(for_parts(ctx, [&](Part &pt) -> int {
        const int i = pt.index;
        const int64_t N = s->n(i);
        const bool first = &pt == &p0;
        DevBuf &pack = first ? st->pack : pt.pack, &spack = first ? st->spack : pt.spack;
        double *d = blk;
        if (!first) {
            CK(pt.params.ensure(sizeof(double) * (nblock + 2 * (size_t)K)));
            d = pt.params.d();
            if (pt.device == p0.device)
                HK(hipMemcpyAsync(d, blk, sizeof(double) * nblock, hipMemcpyDeviceToDevice, pt.stream), "hipMemcpyAsync (block to a device)");
            else
                HK(hipMemcpyPeerAsync(d, pt.device, blk, p0.device, sizeof(double) * nblock, pt.stream), "hipMemcpyPeerAsync (block to a device)");
        }
        const double *shift_here = d_shift ? (first ? d_shift : d + o_shift) : nullptr;
        if (with_pack) {
            CK(pack.ensure((size_t)K * stride * sizeof(double)));
            if (d_shift) CK(spack.ensure((size_t)K * stride * sizeof(double)));
            pmc_vb_fields f = st->f;                                // (the first device: the state itself)
            if (!first) {
                f.m = d; f.W = d + KD; f.beta = d + KD + KDD; f.nu = f.beta + K; f.ln_pi = f.nu + K; f.ln_lambda = f.ln_pi + K;
                f.log_det_W = d + o_ld;
            }
            CK(pmc_vb_pack_device(K, D, &f, first ? st->psi_parts : d + o_parts, pack.d(), first ? d_pstatus : d + nblock, shift_here,
                                  d_shift ? spack.d() : nullptr, pt.stream));
        } else {
            CK(spack.ensure((size_t)K * stride * sizeof(double)));
            CK(pmc_pack_means_device(K, D, shift_here, spack.d(), pt.stream));
        }
        CK(workspace(pt, N, K, D));
        CK(pt.u.ensure(sizeof(double) * (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, K)));
        CK(pt.flat.ensure(sizeof(double) * nflat));
        const double *d_sw = (N > 0 && s->has_sw) ? s->sw[i].d() : nullptr;
        CK(pmc_estep_about(s->x[i].d(), N, D, pack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, d_sw, nullptr, pt.u.d(), nullptr, nullptr,
                           pt.flat.d() + NSC, pt.flat.d(), pt.ws.p, d_shift ? spack.d() : nullptr, pt.stream));
        return publish(ctx, pt, pt.flat.d(), nflat);
    }));
    CK(use(ctx));
    CK(reduce(ctx, p0.flat.d(), nflat));
    return pmc_vb_convert_after_device(K, D, p0.flat.d() + NSC, d_shift ? d_shift : st->f.m, p0.flat.d(), st->conv, &st->f, st->result,
                                       st->shift_next, st->log_q_Z, p0.stream);
}

static int vb_state_pass(pmc_vb_state *st, const pmc_samples *s, Part &pt, const double *d_shift, bool with_pack, bool own_psi)
{
    pmc_ctx *ctx = st->ctx;
    if (ctx->nparts() > 1) return vb_state_pass_parts(st, s, d_shift, with_pack, own_psi);
    const int K = st->K, D = st->D;
    const int64_t stride = pmc_pack_stride(D), PS = pmc_stats_stride(D);
    const size_t nflat = NSC + (size_t)K * PS;
    const int64_t N = s->n(0);
    double *d_pstatus = st->result + (size_t)pmc_vb_small_len(K);
    if (with_pack) {
        CK(st->pack.ensure((size_t)K * stride * sizeof(double)));
        if (d_shift) CK(st->spack.ensure((size_t)K * stride * sizeof(double)));
        if (own_psi) {
            // the device's psi: the expectations in a launch of their own, then the pack
            CK(pmc_vb_expectations_device(K, D, &st->f, nullptr, st->c0, st->c3, pt.stream));
            CK(pmc_pack_components_device(K, D, st->f.m, st->f.W, st->c0, st->f.nu, st->f.ln_pi, st->c3, nullptr, nullptr, st->pack.d(),
                                          d_pstatus, d_shift, d_shift ? st->spack.d() : nullptr, pt.stream));
        } else {
            CK(pmc_vb_pack_device(K, D, &st->f, st->psi_parts, st->pack.d(), d_pstatus, d_shift, d_shift ? st->spack.d() : nullptr,
                                  pt.stream));
        }
    } else {
        CK(st->spack.ensure((size_t)K * stride * sizeof(double)));
        CK(pmc_pack_means_device(K, D, d_shift, st->spack.d(), pt.stream));
    }
    CK(workspace(pt, N, K, D));
    CK(pt.u.ensure(sizeof(double) * (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, K)));
    CK(pt.flat.ensure(sizeof(double) * nflat));
    const double *d_sw = (N > 0 && s->has_sw) ? s->sw[0].d() : nullptr;
    double *d_flat = pt.flat.d();
    CK(pmc_estep_about(s->x[0].d(), N, D, st->pack.d(), K, PMC_KIND_VB, PMC_RESP_VB, 0, d_sw, nullptr, pt.u.d(), nullptr, nullptr,
                       d_flat + NSC, d_flat, pt.ws.p, d_shift ? st->spack.d() : nullptr, pt.stream));
    CK(reduce(ctx, d_flat, nflat));
    return pmc_vb_convert_after_device(K, D, d_flat + NSC, d_shift ? d_shift : st->f.m, d_flat, st->conv, &st->f, st->result,
                                       st->shift_next, st->log_q_Z, pt.stream);
}

int pmc_vb_state_create(pmc_ctx *ctx, int K, int D, pmc_vb_state **out)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!out || K < 1 || D < 1) return failf(PMC_EINVAL, "pmc_vb_state_create: bad argument");
    if (D > pmc_vb_max_dim() || D > pmc_max_compiled_dim())
        return failf(PMC_EINVAL, "pmc_vb_state_create: D = %d: the device-resident update covers D <= %d", D, pmc_vb_max_dim());
    pmc_vb_state *st = new pmc_vb_state();
    st->ctx = ctx;
    st->K = K;
    st->D = D;
    st->stepped = false;
    st->nconv = (size_t)pmc_convert_stats_len(K, D);
    st->nresult = (size_t)pmc_vb_state_result_len(K);
    const size_t KD = (size_t)K * D;
    // the fields; [m | W | beta | nu | ln_pi | ln_lambda] and their E-step copies in the same order, each contiguous
    static const int order[PMC_VB_NFIELDS] = {PMC_VB_ALPHA0, PMC_VB_BETA0, PMC_VB_NU0, PMC_VB_M0, PMC_VB_INV_W0, PMC_VB_LOG_DET_W0,
                                              PMC_VB_ALPHA, PMC_VB_LOG_DET_W, PMC_VB_N_COMP, PMC_VB_X_MEAN, PMC_VB_S, PMC_VB_SHIFT_PREV,
                                              PMC_VB_M, PMC_VB_W, PMC_VB_BETA, PMC_VB_NU, PMC_VB_LN_PI, PMC_VB_LN_LAMBDA,
                                              PMC_VB_E_M, PMC_VB_E_W, PMC_VB_E_BETA, PMC_VB_E_NU, PMC_VB_E_LN_PI, PMC_VB_E_LN_LAMBDA};
    size_t total = 0;
    for (int i = 0; i < PMC_VB_NFIELDS; ++i) total += vb_field_len(order[i], K, D);
    const size_t nbcast = ctx->nparts() > 1 ? 2 * KD + KD * D + 7 * (size_t)K : 0;   // (the block the other devices get: vb_state_pass_parts)
    const size_t extra = 4 * (size_t)K + 8 + st->nresult + st->nconv + (size_t)pmc_vb_bound_scratch_len(K) + 2 * KD + nbcast;
    if (st->buf.ensure(sizeof(double) * (total + extra)) < 0) {
        delete st;
        return PMC_EHIP;
    }
    double *p = st->buf.d();
    for (int i = 0; i < PMC_VB_NFIELDS; ++i) {
        st->field[order[i]] = p;
        p += vb_field_len(order[i], K, D);
    }
    st->c0 = p; p += K;
    st->c3 = p; p += K;
    st->psi_parts = p; p += 2 * (size_t)K;
    st->log_q_Z = p; p += 8;
    st->result = p; p += st->nresult;
    st->conv = p; p += st->nconv;
    st->terms = p; p += (size_t)pmc_vb_bound_scratch_len(K);
    st->shift2 = p; p += KD;
    st->shift_next = p; p += KD;
    st->bcast = p; p += nbcast;
    Part &pt = *ctx->parts[0];
    const hipError_t e = hipMemsetAsync(st->buf.p, 0, sizeof(double) * (total + extra), pt.stream);
    if (e != hipSuccess) {
        st->buf.release();
        delete st;
        return hipf(e, "hipMemsetAsync");
    }
    pmc_vb_fields &f = st->f;
    f.alpha0 = st->field[PMC_VB_ALPHA0]; f.beta0 = st->field[PMC_VB_BETA0]; f.nu0 = st->field[PMC_VB_NU0];
    f.m0 = st->field[PMC_VB_M0]; f.inv_W0 = st->field[PMC_VB_INV_W0]; f.log_det_W0 = st->field[PMC_VB_LOG_DET_W0];
    f.alpha = st->field[PMC_VB_ALPHA]; f.beta = st->field[PMC_VB_BETA]; f.nu = st->field[PMC_VB_NU];
    f.m = st->field[PMC_VB_M]; f.W = st->field[PMC_VB_W]; f.log_det_W = st->field[PMC_VB_LOG_DET_W];
    f.ln_lambda = st->field[PMC_VB_LN_LAMBDA]; f.ln_pi = st->field[PMC_VB_LN_PI];
    f.N_comp = st->field[PMC_VB_N_COMP]; f.x_mean = st->field[PMC_VB_X_MEAN]; f.S = st->field[PMC_VB_S];
    {
        std::lock_guard<std::mutex> lk(g_vb_states_mu);
        g_vb_states.push_back(st);
    }
    *out = st;
    return PMC_OK;
}

int pmc_vb_state_destroy(pmc_vb_state *st)
{
    if (!st || !vb_state_alive(st)) return PMC_OK;                  // (gone with its context: pmc_shutdown)
    CK(use(st->ctx));
    CtxCall call_(st->ctx);
    (void)hipStreamSynchronize(st->ctx->parts[0]->stream);
    {
        std::lock_guard<std::mutex> lk(g_vb_states_mu);
        for (size_t i = 0; i < g_vb_states.size(); ++i)
            if (g_vb_states[i] == st) {
                g_vb_states.erase(g_vb_states.begin() + (long)i);
                break;
            }
    }
    st->buf.release();
    st->pack.release();
    st->spack.release();
    delete st;
    return PMC_OK;
}

int64_t pmc_vb_state_result_len(int K) { return K < 1 ? (int64_t)failf(PMC_EINVAL, "pmc_vb_state_result_len: bad K") : pmc_vb_small_len(K) + 4 * (int64_t)K + 8; }

int pmc_vb_state_put(pmc_vb_state *st, int field, const double *h)
{
    if (!st || !h || field < 0 || field >= PMC_VB_E_M) return failf(PMC_EINVAL, "pmc_vb_state_put: bad argument (the E_* fields are read-only)");
    if (!vb_state_alive(st)) return failf(PMC_EINVAL, "pmc_vb_state_put: this state is gone (its context was shut down)");
    CK(use(st->ctx));
    CtxCall call_(st->ctx);
    Part &pt = *st->ctx->parts[0];
    if (field == PMC_VB_W)                                          // (W from outside replaces the M-step's: a failed one's flags go)
        HK(hipMemsetAsync(st->result + (size_t)pmc_vb_small_len(st->K) + 2 * (size_t)st->K, 0, sizeof(double) * 2 * (size_t)st->K, pt.stream),
           "hipMemsetAsync");
    return h2d(pt, st->field[field], h, sizeof(double) * vb_field_len(field, st->K, st->D));
}

int pmc_vb_state_get(pmc_vb_state *st, int field, double *h)
{
    if (!st || !h || field < 0 || field >= PMC_VB_NFIELDS) return failf(PMC_EINVAL, "pmc_vb_state_get: bad argument");
    if (!vb_state_alive(st)) return failf(PMC_EINVAL, "pmc_vb_state_get: this state is gone (its context was shut down)");
    if (field >= PMC_VB_E_M && !st->stepped) return failf(PMC_EINVAL, "pmc_vb_state_get: no E-step has run on this state yet");
    CK(use(st->ctx));
    CtxCall call_(st->ctx);
    return d2h(*st->ctx->parts[0], h, st->field[field], sizeof(double) * vb_field_len(field, st->K, st->D));
}

int pmc_vb_state_step(pmc_vb_state *st, const pmc_samples *s, int flags, const double *h_psi_parts, double *h_result)
{
    if (!st) return failf(PMC_EINVAL, "pmc_vb_state_step: NULL state");
    if (!vb_state_alive(st)) return failf(PMC_EINVAL, "pmc_vb_state_step: this state is gone (its context was shut down)");
    pmc_ctx *ctx = st->ctx;
    CK(use(ctx));
    CtxCall call_(ctx);
    const bool do_m = flags & PMC_VB_DO_MSTEP, do_e = flags & PMC_VB_DO_ESTEP, do_b = flags & PMC_VB_DO_BOUND;
    if (do_e && (!s || s->ctx != ctx || s->D != st->D)) return failf(PMC_EINVAL, "pmc_vb_state_step: the E-step needs samples of this context and dimension");
    if ((do_e || do_b) && !h_result) return failf(PMC_EINVAL, "pmc_vb_state_step: h_result is required with an E-step or the bound");
    const int K = st->K, D = st->D;
    Part &pt = *ctx->parts[0];
    CallLog log_(ctx, "pmc_vb_state_step", do_e ? s->N : 0, K, D);
    const size_t nsmall = (size_t)pmc_vb_small_len(K);
    double *d_mstatus = st->result + nsmall + 2 * (size_t)K, *d_bound = d_mstatus + 2 * (size_t)K;
    if (do_m) CK(pmc_vb_mstep_device(K, D, &st->f, d_mstatus, pt.stream));
    if (do_e && h_psi_parts) CK(h2d(pt, st->psi_parts, h_psi_parts, sizeof(double) * 2 * (size_t)K));
    const size_t KD = (size_t)K * D, KDD = KD * D;
    for (int pass = 0; pass < (do_e ? 2 : 1); ++pass) {
        if (do_e) {
            const double *d_shift = pass == 1 ? st->shift2 : ((flags & PMC_VB_ABOUT_PREV) ? st->field[PMC_VB_SHIFT_PREV] : nullptr);
            CK(vb_state_pass(st, s, pt, d_shift, pass == 0, h_psi_parts == nullptr));
        }
        if (do_b) CK(pmc_vb_bound_device(K, D, &st->f, st->log_q_Z, st->terms, d_bound, pt.stream));
        if (!h_result) return PMC_OK;                               // (an M-step alone: queued; its flags travel with the next block)
        if (do_e && pass == 0)
            // the parameters this E-step runs with, for whoever asks for r / log rho later (queued in front of the wait
            // for the block: behind it, it would sit between the block's arrival and the caller's next launch)
            HK(hipMemcpyAsync(st->field[PMC_VB_E_M], st->field[PMC_VB_M], sizeof(double) * (KD + KDD + 4 * (size_t)K),
                              hipMemcpyDeviceToDevice, pt.stream), "hipMemcpyAsync (E-step parameters)");
        CK(d2h(pt, h_result, st->result, sizeof(double) * st->nresult));
        if (ctx->p2p && do_e) CK(pmc_p2p_status(ctx->p2p, pt.stream));
        CK(pmc_vb_mstep_status(K, h_result + nsmall + 2 * (size_t)K));
        if (!do_e) return PMC_OK;
        if (pass == 0) CK(pmc_pack_status(K, h_result + nsmall));
        bool far = false;
        for (int k = 0; k < K; ++k) far = far || h_result[(size_t)K + k] != 0.0;
        if (pass == 1 || !far) break;
        // second pass about the mean just found (variational.pyx:806-932 takes the mean first, then the covariance about it)
        const double *d_old = (flags & PMC_VB_ABOUT_PREV) ? st->field[PMC_VB_SHIFT_PREV] : st->f.m;
        CK(pmc_vb_newshift_device(K, D, st->conv, d_old, st->shift2, pt.stream));
    }
    // the next E-step's shifts = this one's means
    double *old = st->field[PMC_VB_SHIFT_PREV];
    st->field[PMC_VB_SHIFT_PREV] = st->shift_next;
    st->shift_next = old;
    st->stepped = true;
    return PMC_OK;
}

// GaussianInference.run's loop (variational.pyx:283-359) while no component has to go: update() = M-step + E-step, the bound,
// the reference's convergence rules -- with nothing but the psi callback between two iterations (the interpreter's share of an
// iteration was 80 us: 5 % at one GPU's share of eight, a third of an iteration at 1e4 samples).
int pmc_vb_state_run(pmc_vb_state *st, const pmc_samples *s, int max_iterations, double old_bound, double prune_threshold,
                     double rel_tol, double abs_tol, int about_prev, const double *h_N_comp, pmc_vb_psi_fn psi, void *user,
                     double *h_result, int *h_info, double *h_bounds)
{
    if (!st || !s || !h_result || !h_info || !h_bounds || !h_N_comp || max_iterations < 0)
        return failf(PMC_EINVAL, "pmc_vb_state_run: bad argument");
    const int K = st->K;
    std::vector<double> n_comp(h_N_comp, h_N_comp + K), parts(2 * (size_t)K);
    h_info[0] = 0;                                                  // updates done
    h_info[1] = PMC_VB_RUN_CAP;
    h_info[2] = 0;                                                  // times the bound decreased
    h_info[3] = about_prev ? 1 : 0;                                 // moments of the NEXT E-step about the latest means?
    h_bounds[0] = h_bounds[1] = old_bound;
    double bound = old_bound;
    for (int i = 0; i < max_iterations; ++i) {
        old_bound = bound;
        CK(pmc_vb_state_step(st, nullptr, PMC_VB_DO_MSTEP, nullptr, nullptr));             // queued: runs beside the callback
        if (psi) psi(user, K, n_comp.data(), parts.data());
        CK(pmc_vb_state_step(st, s, PMC_VB_DO_ESTEP | PMC_VB_DO_BOUND | (h_info[3] ? PMC_VB_ABOUT_PREV : 0), psi ? parts.data() : nullptr,
                             h_result));
        h_info[0] = i + 1;
        bool finite_means = true, prune = false;
        for (int k = 0; k < K; ++k) {
            n_comp[k] = h_result[k];
            finite_means = finite_means && h_result[2 * (size_t)K + k] != 0.0;
            prune = prune || h_result[k] < prune_threshold;         // (variational.pyx:246: survivors have N_k >= threshold)
        }
        h_info[3] = finite_means ? 1 : 0;
        bound = h_result[8 * (size_t)K + 8];
        h_bounds[0] = bound;
        h_bounds[1] = old_bound;
        // the caller's checks of N_comp and S (variational.pyx:122-126) need a look at the block: hand it back
        bool any_n = false, any_s = false;
        for (int k = 0; k < K; ++k) {
            any_n = any_n || std::isfinite(h_result[k]);
            any_s = any_s || h_result[3 * (size_t)K + k] != 0.0;
        }
        if (!any_n || !any_s || !std::isfinite(bound)) {
            h_info[1] = PMC_VB_RUN_LOOK;
            return PMC_OK;
        }
        if (bound < old_bound) h_info[2] += 1;
        if (bound == old_bound) {
            h_info[1] = PMC_VB_RUN_CONVERGED;
            return PMC_OK;
        }
        const double diff = bound - old_bound;
        if (diff > 0) {
            if (std::fabs(bound) < abs_tol) {
                if (std::fabs(diff) < abs_tol) {
                    h_info[1] = PMC_VB_RUN_CONVERGED;
                    return PMC_OK;
                }
            } else if (std::fabs(diff / bound) < rel_tol) {
                h_info[1] = PMC_VB_RUN_CONVERGED;
                return PMC_OK;
            }
        }
        if (prune) {
            h_info[1] = PMC_VB_RUN_PRUNE;
            return PMC_OK;
        }
    }
    return PMC_OK;
}

// ---- PMC update -------------------------------------------------------------------------------------------
int pmc_pmc_update_stats(pmc_ctx *ctx, const pmc_mix *mix, const pmc_samples *s, const double *h_w, int weights_on_device,
                         const int64_t *h_latent, int rb, double *h_alpha, double *h_mu, double *h_sigma,
                         double *h_dof_const, double *h_loglik, double *h_norm)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!mix || !s || mix->ctx != ctx || s->ctx != ctx || mix->D != s->D || !h_alpha || !h_mu || !h_sigma)
        return failf(PMC_EINVAL, "pmc_pmc_update_stats: bad argument");
    if (h_w && weights_on_device) return failf(PMC_EINVAL, "pmc_pmc_update_stats: h_w or weights_on_device, not both");
    if (weights_on_device && !s->has_w && s->N > 0) return failf(PMC_EINVAL, "pmc_pmc_update_stats: no importance weights on the device (pmc_is_weights first)");
    if (!rb && !h_latent && !s->has_origin) return failf(PMC_EINVAL, "`rb` must be True if `latent` is not provided!");   // pmc.pyx:81-83
    const bool student = mix->family == PMC_KIND_STUDENT_T;
    if (student && !h_dof_const) return failf(PMC_EINVAL, "pmc_pmc_update_stats: StudentT components need h_dof_const");
    const int K = mix->K, D = mix->D;
    const int64_t PS = pmc_stats_stride(D);
    CallLog log_(ctx, "pmc_pmc_update_stats", s->N, K, D);
    std::vector<int> live;
    for (int k = 0; k < K; ++k)
        if (mix->w[k] != 0.0) live.push_back(k);                       // pmc.pyx:66
    const int L = (int)live.size();
    const size_t nstat = NSC + (size_t)L * PS + 2 * (size_t)L, nflat = nstat + 1;   // ... | this part's sum of weights
    CK(prepare_slots(ctx, nflat));
    std::vector<double> shift((size_t)(L > 0 ? L : 1) * D, 0.0);
    for (int i = 0; i < L; ++i) std::memcpy(&shift[(size_t)i * D], &mix->mu[(size_t)live[i] * D], sizeof(double) * D);
    std::vector<double> host, shost;
    if (L > 0) CK(build_mix_pack(mix, live, host));
    std::vector<double> flat(nflat, 0.0), S0, M1, M2;
    std::vector<double> norms((size_t)ctx->nparts(), 0.0);            // (kept across the passes)
    const int mode = rb ? PMC_RESP_PMC_RB : PMC_RESP_PMC_LATENT;
    Part &p0 = *ctx->parts[0];
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) CK(means_pack_host(shift, L, D, shost));
        CK(for_parts(ctx, [&](Part &pt) -> int {
            const int i = pt.index;
            const int64_t N = s->n(i), b = s->begin[i];
            CK(pt.flat.ensure(sizeof(double) * (NSC + (size_t)(L > 0 ? L : 1) * (PS + 2) + 1)));
            double *d_flat = pt.flat.d();
            const double *d_w = nullptr;
            if (pass == 0) {
                CK(workspace(pt, N, L > 0 ? L : 1, D));
                // weights and their local sum
                double local_norm = (double)N;
                if (h_w && N > 0) {
                    CK(pt.aux.ensure(sizeof(double) * (size_t)N));
                    CK(h2d(pt, pt.aux.p, h_w + b, sizeof(double) * (size_t)N));
                    long double acc = 0.0L;
                    for (int64_t n = 0; n < N; ++n) acc += h_w[b + n];
                    local_norm = (double)acc;
                } else if (weights_on_device && N > 0) {
                    CK(pmc_weight_sums(s->w[i].d(), N, d_flat, pt.ws.p, pt.stream));
                    double sc3[3];
                    CK(d2h(pt, sc3, d_flat, sizeof(sc3)));
                    local_norm = sc3[0];
                } else if (h_w || weights_on_device) {
                    local_norm = 0.0;
                }
                norms[(size_t)i] = local_norm;
                if (!rb && N > 0 && h_latent) {
                    CK(pt.lat.ensure(sizeof(int64_t) * (size_t)N));
                    CK(h2d(pt, pt.lat.p, h_latent + b, sizeof(int64_t) * (size_t)N));
                }
                if (L > 0) {
                    CK(pt.pack.ensure(host.size() * sizeof(double)));
                    CK(h2d(pt, pt.pack.p, host.data(), host.size() * sizeof(double)));
                    CK(pt.u.ensure(sizeof(double) * (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, L)));
                    if (student) CK(pt.scratch.ensure(sizeof(double) * (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, L)));
                }
            }
            if (N > 0) d_w = h_w ? pt.aux.d() : (weights_on_device ? s->w[i].d() : nullptr);
            const int64_t *d_lat = nullptr;
            if (!rb && N > 0) d_lat = h_latent ? (const int64_t *)pt.lat.p : (const int64_t *)s->origin[i].p;
            HK(hipMemsetAsync(d_flat, 0, sizeof(double) * nflat, pt.stream), "hipMemsetAsync");
            if (L > 0) {
                if (pass == 1) CK(upload_spack(pt, shost));
                // dead components' all-zero columns take part in the row maximum (pmc.pyx:24-34)
                CK(pmc_estep_about(s->x[i].d(), N, D, pt.pack.d(), L, mix->family, mode, L < K ? 1 : 0, d_w, d_lat, pt.u.d(),
                                   student ? pt.scratch.d() : nullptr, student ? d_flat + NSC + (size_t)L * PS : nullptr,
                                   d_flat + NSC, d_flat, pt.ws.p, pass == 1 ? pt.spack.d() : nullptr, pt.stream));
            }
            HK(hipMemcpyAsync(d_flat + nstat, &norms[(size_t)i], sizeof(double), hipMemcpyHostToDevice, pt.stream), "hipMemcpyAsync");
            HK(hipStreamSynchronize(pt.stream), "hipStreamSynchronize");
            return publish(ctx, pt, d_flat, nflat);
        }));
        CK(reduce(ctx, p0.flat.d(), nflat));
        CK(d2h(p0, flat.data(), p0.flat.d(), sizeof(double) * nflat));
        if (ctx->p2p) CK(pmc_p2p_status(ctx->p2p, p0.stream));
        if (L == 0) break;
        split_stats(flat.data() + NSC, L, D, S0, M1, M2);
        if (pass == 1 || !shift_is_far(S0, M1, M2, L, D)) break;
        new_shifts(S0, M1, L, D, shift);
    }
    const double norm = flat[nstat];
    if (h_norm) *h_norm = norm;
    if (h_loglik) *h_loglik = flat[3];
    if (L == 0) return PMC_OK;
    std::vector<double> mean((size_t)L * D), cov((size_t)L * D * D), V1(L), V2(L);
    const double *vs = flat.data() + NSC + (size_t)L * PS;
    for (int i = 0; i < L; ++i) {
        V1[i] = vs[2 * i];
        V2[i] = vs[2 * i + 1];
    }
    // Gauss: both normalised by sum w rho (pmc.pyx:194-204); Student-t: the mean by sum w rho gamma, the covariance by
    // sum w rho (pmc.pyx:620-630)
    centred_moments(S0.data(), student ? V1.data() : S0.data(), M1, M2, shift.data(), L, D, mean.data(), cov.data());
    for (int i = 0; i < L; ++i) {
        const int k = live[i];
        const double wr = student ? V1[i] : S0[i];
        h_alpha[k] = wr / norm;                                          // pmc.pyx:191-193, :612-617
        std::memcpy(&h_mu[(size_t)k * D], &mean[(size_t)i * D], sizeof(double) * D);
        std::memcpy(&h_sigma[(size_t)k * D * D], &cov[(size_t)i * D * D], sizeof(double) * (size_t)D * D);
        if (student) {
            // sum_n w_n (xi + delta)_nk of pmc.pyx:659-679 from the device sums (pypmc_amd/mix_adapt/pmc.py::student_t_pmc)
            const double nu = mix->dof[k];
            const double total = V2[i] - digamma(.5 * (D + nu)) * V1[i] + (norm - V1[i]) * (std::log(.5 * nu) - digamma(.5 * nu)) +
                                 S0[i] + (norm - V1[i]);
            h_dof_const[k] = 1. - total / norm;
        }
    }
    return PMC_OK;
}

// ---- weighted moments -------------------------------------------------------------------------------------
int pmc_weighted_moments(pmc_ctx *ctx, const pmc_samples *s, const double *h_w, int weights_on_device, double *h_mean,
                         double *h_cov)
{
    CK(use(ctx));
    CtxCall call_(ctx);
    if (!s || s->ctx != ctx || !h_mean) return failf(PMC_EINVAL, "pmc_weighted_moments: bad argument");
    if (h_w && weights_on_device) return failf(PMC_EINVAL, "pmc_weighted_moments: h_w or weights_on_device, not both");
    if (weights_on_device && !s->has_w) return failf(PMC_EINVAL, "pmc_weighted_moments: no importance weights on the device (pmc_is_weights first)");
    const int D = s->D;
    const int64_t PS = pmc_stats_stride(D), stride = pmc_pack_stride(D);
    if (stride < 0) return (int)stride;
    CallLog log_(ctx, "pmc_weighted_moments", s->N, 1, D);
    // the shift: the first sample of rank 0 (every rank and every part must take its moments about the same point)
    std::vector<double> shift(D, 0.0);
    int rank = 0;
    if (ctx->comm) CK(pmc_comm_rank(ctx->comm, &rank, nullptr));
    Part &p0 = *ctx->parts[0];
    const size_t nflat = (size_t)(PS + NSC);
    CK(p0.flat.ensure(sizeof(double) * (nflat + D)));
    {
        double *d_shift = p0.flat.d() + nflat;
        HK(hipMemsetAsync(d_shift, 0, sizeof(double) * D, p0.stream), "hipMemsetAsync");
        if (rank == 0 && s->N > 0) {
            int first = 0;                                            // (the first part that holds a row)
            while (s->n(first) == 0) ++first;
            if (ctx->parts[first]->device == p0.device)
                HK(hipMemcpyAsync(d_shift, s->x[first].p, sizeof(double) * D, hipMemcpyDeviceToDevice, p0.stream), "hipMemcpyAsync");
            else
                HK(hipMemcpyPeerAsync(d_shift, p0.device, s->x[first].p, ctx->parts[first]->device, sizeof(double) * D, p0.stream), "hipMemcpyPeerAsync");
        }
        if (ctx->p2p) CK(pmc_p2p_allreduce_sum(ctx->p2p, d_shift, D, p0.stream));
        else if (ctx->comm) CK(pmc_comm_allreduce_sum(ctx->comm, d_shift, D, p0.stream));
        CK(d2h(p0, shift.data(), d_shift, sizeof(double) * D));
    }
    std::vector<double> hp((size_t)stride);
    CK(pmc_pack_means(1, D, shift.data(), hp.data()));
    CK(prepare_slots(ctx, nflat));
    CK(for_parts(ctx, [&](Part &pt) -> int {
        const int i = pt.index;
        const int64_t N = s->n(i), b = s->begin[i];
        CK(pt.flat.ensure(sizeof(double) * (nflat + D)));
        double *d_flat = pt.flat.d();                                     // [stats PS | sum w, sum w log w, sum w^2 ... (NSC)]
        HK(hipMemsetAsync(d_flat, 0, sizeof(double) * nflat, pt.stream), "hipMemsetAsync");
        CK(upload_spack(pt, hp));
        CK(workspace(pt, N, 1, D));
        // u = the weights in the library's tile-major layout with K = 1: the weight vector itself, zero behind the samples
        const size_t ulen = (size_t)pmc_tile_buffer_len(N > 0 ? N : 1, 1);
        CK(pt.u.ensure(sizeof(double) * ulen));
        HK(hipMemsetAsync(pt.u.p, 0, sizeof(double) * ulen, pt.stream), "hipMemsetAsync");
        if (N > 0) {
            if (h_w) CK(h2d(pt, pt.u.p, h_w + b, sizeof(double) * (size_t)N));
            else if (weights_on_device) HK(hipMemcpyAsync(pt.u.p, s->w[i].p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, pt.stream), "hipMemcpyAsync");
            else {
                std::vector<double> ones((size_t)N, 1.0);
                CK(h2d(pt, pt.u.p, ones.data(), sizeof(double) * (size_t)N));
            }
            CK(pmc_sufficient_stats(s->x[i].d(), N, D, pt.spack.d(), 1, pt.u.d(), d_flat, pt.ws.p, pt.stream));
            CK(pmc_weight_sums(pt.u.d(), N, d_flat + PS, pt.ws.p, pt.stream));
        }
        return publish(ctx, pt, d_flat, nflat);
    }));
    CK(reduce(ctx, p0.flat.d(), nflat));
    std::vector<double> flat(nflat);
    CK(d2h(p0, flat.data(), p0.flat.d(), sizeof(double) * nflat));
    if (ctx->p2p) CK(pmc_p2p_status(ctx->p2p, p0.stream));
    std::vector<double> S0, M1, M2;
    split_stats(flat.data(), 1, D, S0, M1, M2);
    const double sw = S0[0], q = flat[(size_t)PS + 2];
    std::vector<double> dbar(D);
    for (int i = 0; i < D; ++i) {
        dbar[i] = M1[i] / sw;
        h_mean[i] = shift[i] + dbar[i];                                   // importance_sampling.py:58-61
    }
    if (h_cov) {
        const double corr = sw * sw / (sw * sw - q);                      // :76-83
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) h_cov[(size_t)i * D + j] = corr * (M2[(size_t)i * D + j] / sw - dbar[i] * dbar[j]);
    }
    return PMC_OK;
}

// ---- host-side conversion ---------------------------------------------------------------------------------
int pmc_host_convert_stats(int K, int D, const double *h_stats, const double *h_shift, const double *h_n_cov, double *h_S0,
                           double *h_M1, double *h_mean, double *h_cov, int *h_far)
{
    if (K < 1 || D < 1 || !h_stats || !h_shift || !h_S0 || !h_mean || !h_cov)
        return failf(PMC_EINVAL, "pmc_host_convert_stats: bad argument");
    std::vector<double> S0, M1, M2;
    split_stats(h_stats, K, D, S0, M1, M2);
    std::memcpy(h_S0, S0.data(), sizeof(double) * K);
    if (h_M1) std::memcpy(h_M1, M1.data(), sizeof(double) * (size_t)K * D);
    if (h_far) *h_far = shift_is_far(S0, M1, M2, K, D) ? 1 : 0;
    centred_moments(S0.data(), h_n_cov ? h_n_cov : S0.data(), M1, M2, h_shift, K, D, h_mean, h_cov);
    return PMC_OK;
}

// ---- K-sized host linear algebra ------------------------------------------------------------------------------
// chol_inv_det (pypmc/tools/_linalg.pyx:41-95) of a stack of K symmetric D x D matrices: potrf, potri of the lower
// factor, the inverse mirrored, log det = 2 sum log L_ii summed left to right -- the SAME LAPACK routines the reference
// calls through scipy, handed in by address (scipy.linalg.cython_lapack's dpotrf / dpotri: Fortran calling convention,
// column-major), so the numbers are scipy's own.  A symmetric matrix reads the same row- and column-major, the factor
// comes back transposed and is turned here.
// h_failed (K ints, may be NULL): potrf's / potri's info per matrix (0 = fine), or -1 for a non-finite log det.
typedef void (*pmc_lapack_fn)(char *uplo, int *n, double *a, int *lda, int *info);
int pmc_host_chol_inv_det_batch(int K, int D, const double *h_m, void *dpotrf, void *dpotri, double *h_lower,
                                double *h_inverse, double *h_log_det, int *h_failed)
{
    if (K < 1 || D < 1 || !h_m || !dpotrf || !dpotri || !h_lower || !h_inverse || !h_log_det)
        return failf(PMC_EINVAL, "pmc_host_chol_inv_det_batch: bad argument");
    const pmc_lapack_fn potrf = (pmc_lapack_fn)dpotrf, potri = (pmc_lapack_fn)dpotri;
    std::vector<int> info((size_t)K, 0);
    auto work = [&](int k0, int k1) {
        std::vector<double> a((size_t)D * D);
        char lo = 'L';
        for (int k = k0; k < k1; ++k) {
            int n = D, lda = D, inf = 0;
            std::memcpy(a.data(), h_m + (size_t)k * D * D, sizeof(double) * (size_t)D * D);
            potrf(&lo, &n, a.data(), &lda, &inf);
            if (inf != 0) {
                info[k] = inf;
                continue;
            }
            // column-major lower factor: element (i, j), i >= j, at a[j * D + i]
            double *L = h_lower + (size_t)k * D * D;
            double ld = 0.0;
            for (int i = 0; i < D; ++i) {
                for (int j = 0; j < D; ++j) L[(size_t)i * D + j] = j <= i ? a[(size_t)j * D + i] : 0.0;
                ld += std::log(a[(size_t)i * D + i]);                // left to right, as the reference sums
            }
            ld *= 2.0;
            h_log_det[k] = ld;
            if (!std::isfinite(ld)) info[k] = -1;
            potri(&lo, &n, a.data(), &lda, &inf);
            if (inf != 0) {
                info[k] = inf;
                continue;
            }
            double *I = h_inverse + (size_t)k * D * D;
            for (int i = 0; i < D; ++i)
                for (int j = 0; j <= i; ++j) I[(size_t)i * D + j] = I[(size_t)j * D + i] = a[(size_t)j * D + i];
        }
    };
    // On the calling thread unless PMC_HOST_THREADS asks for more: scipy's OpenBLAS serialises calls that arrive from
    // several threads at once (measured: K = 128, D = 40: 2.6 ms on one thread, 4.2 ms on 2 ... 8), so spreading the
    // matrices only pays with a LAPACK that does not.  The gain over the numpy loop this replaces (6.4 ms on the same
    // host) is the glue: no per-matrix Python call, no transposed copies, no triangle masks.
    int nt = 1;
    if (const char *e = std::getenv("PMC_HOST_THREADS")) nt = std::atoi(e);
    if (nt > K / 8) nt = K / 8;                                     // (a thread is worth starting for eight 40 x 40 matrices)
    if ((size_t)D * D * K < 20000 || nt < 2) {
        work(0, K);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, (int)((long long)K * t / nt), (int)((long long)K * (t + 1) / nt));
        for (std::thread &t : th) t.join();
    }
    int bad = 0;
    for (int k = 0; k < K; ++k) {
        if (h_failed) h_failed[k] = info[k];
        bad += info[k] != 0;
    }
    return bad ? failf(PMC_ENOTPOSDEF, "pmc_host_chol_inv_det_batch: %d of %d matrices do not factorise", bad, K) : PMC_OK;
}

}  // extern "C"
