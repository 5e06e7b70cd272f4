// pmc_p2p.hip -- the one exchange of the path without a ring: one-shot all-gather + ordered local sum among the ranks of
// ONE node ("one-shot P2P all-gather + ordered local sum as the tuned variant").
//
// What is exchanged is small -- the statistics vector of an update, 7 464 doubles at K = 32, D = 20 and 110 336 at
// K = 128, D = 40 -- so a ring all-reduce (2 (G - 1) dependent hops, each a kernel-side handshake) is pure latency.  Here
// every rank owns a mailbox in its device memory that its peers map through HIP IPC handles; an all-reduce is
//   k_p2p_put:  my vector -> slot [my rank] of EVERY rank's mailbox (peer stores over xGMI), then -- behind a system-scope
//               fence -- my sequence number into that rank's flag [my rank];
//   k_p2p_sum:  wait for all G flags of MY mailbox to show this round's sequence number, then out[i] = ((s_0[i] + s_1[i]) +
//               s_2[i]) + ... in RANK ORDER: one hop, no dependence on arrival order, the same bits on every rank and from
//               run to run (a ring's result depends on where the ring is cut).
// Two sets of slots alternate (a rank that is a round ahead writes the other set: it cannot start round s + 2 before
// every rank has finished reading round s, because round s + 1 completes only when all ranks have PUT for it, which each
// does after its own sum of round s).  The reference has no counterpart: it gathers whole sample histories with mpi4py
// (pypmc/tools/parallel_sampler.py:58-71); this replaces the RCCL all-reduce of pmc_comm_allreduce_sum on request.
//
// Tested on one GPU with 2 and 4 processes sharing the device (tests/test_gpu_p2p.py: bit-equal to the sum in rank order,
// identical on all ranks); RCCL stays the default until a multi-GPU box has timed both.  A rank that waits longer than
// PMC_P2P_TIMEOUT_S (wall clock, default 20 s) for its peers gives up, raises the mailbox's error word and the next call
// (or pmc_p2p_status) reports it -- no kernel of this file can hang a GPU for good.
#include "../../include/pmc_hip.h"
#include "pmc_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int pmc_internal_fail(int code, const char *msg);

namespace {

constexpr int P2P_MAX_WORLD = 16;
constexpr size_t P2P_HEAD = 4096;                           // flags[2][16] uint64 + error word, then the slots

struct Mailbox {                                            // layout of the head of a mailbox (device memory)
    unsigned long long flags[2][P2P_MAX_WORLD];
    unsigned long long error;
};
static_assert(sizeof(Mailbox) <= P2P_HEAD, "mailbox head");

__device__ __forceinline__ double *slot_of(char *box, int set, int src, int world, long long cap)
{
    return (double *)(box + P2P_HEAD) + ((size_t)set * world + src) * (size_t)cap;
}

struct PutArgs {
    char *box[P2P_MAX_WORLD];                               // every rank's mailbox as THIS process maps it
    const double *src;
    long long n, cap;
    int rank, world;
    unsigned long long seq;
};

// block b = peer b: copy, fence, flag
__global__ __launch_bounds__(256) void k_p2p_put(const PutArgs a)
{
    char *box = a.box[blockIdx.x];
    double *dst = slot_of(box, (int)(a.seq & 1), a.rank, a.world, a.cap);
    for (long long i = threadIdx.x; i < a.n; i += 256) __builtin_nontemporal_store(a.src[i], dst + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        Mailbox *m = (Mailbox *)box;
        __hip_atomic_store(&m->flags[a.seq & 1][a.rank], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct SumArgs {
    char *box;                                              // my mailbox
    double *out;
    long long n, cap;
    int world;
    unsigned long long seq;
    long long timeout_ticks;                                // wall_clock64 ticks (100 MHz)
};

__global__ __launch_bounds__(256) void k_p2p_sum(const SumArgs a)
{
    Mailbox *m = (Mailbox *)a.box;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        for (int r = 0; r < a.world && good; ++r) {
            while (__hip_atomic_load(&m->flags[a.seq & 1][r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    good = 0;
                    __hip_atomic_store(&m->error, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(20);
            }
        }
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    __threadfence_system();
    const int set = (int)(a.seq & 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        double s = __builtin_nontemporal_load(slot_of(a.box, set, 0, a.world, a.cap) + i);
        for (int r = 1; r < a.world; ++r) s += __builtin_nontemporal_load(slot_of(a.box, set, r, a.world, a.cap) + i);
        a.out[i] = s;
    }
}

int failf(int code, const char *fmt, const char *what)
{
    char buf[300];
    snprintf(buf, sizeof(buf), fmt, what);
    return pmc_internal_fail(code, buf);
}

}  // namespace

struct pmc_p2p {
    int rank, world, device;
    long long cap;
    size_t bytes;
    char *mine;                                             // my mailbox (my allocation)
    char *box[P2P_MAX_WORLD];                               // all mailboxes as mapped here (box[rank] == mine)
    bool connected;
    unsigned long long seq;
    hipIpcMemHandle_t handle;
};

extern "C" {

int pmc_p2p_create(int rank, int world, int64_t max_doubles, int device, pmc_p2p **out)
{
    if (!out || world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || max_doubles < 1)
        return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_create: bad argument (1 <= world <= 16)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return failf(PMC_EHIP, "pmc_p2p_create: hipSetDevice: %s", hipGetErrorString(e));
    pmc_p2p *p = new pmc_p2p();
    p->rank = rank; p->world = world; p->device = device;
    p->cap = (max_doubles + 31) / 32 * 32;
    p->bytes = P2P_HEAD + sizeof(double) * 2 * (size_t)world * (size_t)p->cap;
    p->connected = false;
    p->seq = 0;
    void *mem = nullptr;
    e = hipMalloc(&mem, p->bytes);
    if (e == hipSuccess) e = hipMemset(mem, 0, P2P_HEAD);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&p->handle, mem);
    if (e != hipSuccess) {
        if (mem) (void)hipFree(mem);
        delete p;
        return failf(PMC_EHIP, "pmc_p2p_create: %s (hipIpcGetMemHandle needs HSA_ENABLE_IPC_MODE_LEGACY=0 on these hosts)",
                     hipGetErrorString(e));
    }
    p->mine = (char *)mem;
    for (int r = 0; r < P2P_MAX_WORLD; ++r) p->box[r] = nullptr;
    p->box[rank] = p->mine;
    *out = p;
    return PMC_OK;
}

int pmc_p2p_handle(const pmc_p2p *p, void *h_handle)
{
    if (!p || !h_handle) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_handle: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= PMC_P2P_HANDLE_BYTES, "IPC handle size");
    std::memset(h_handle, 0, PMC_P2P_HANDLE_BYTES);
    std::memcpy(h_handle, &p->handle, sizeof(hipIpcMemHandle_t));
    return PMC_OK;
}

int pmc_p2p_connect(pmc_p2p *p, const void *h_handles)
{
    if (!p || !h_handles) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_connect: bad argument");
    if (p->connected) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_connect: connected already");
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return failf(PMC_EHIP, "pmc_p2p_connect: hipSetDevice: %s", hipGetErrorString(e));
    for (int r = 0; r < p->world; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char *)h_handles + (size_t)r * PMC_P2P_HANDLE_BYTES, sizeof(h));
        void *mem = nullptr;
        e = hipIpcOpenMemHandle(&mem, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return failf(PMC_EHIP, "pmc_p2p_connect: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        p->box[r] = (char *)mem;
    }
    p->connected = true;
    return PMC_OK;
}

int pmc_p2p_allreduce_sum(pmc_p2p *p, double *d_buf, int64_t n, void *stream)
{
    if (!p || n < 0 || (n > 0 && !d_buf)) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_allreduce_sum: bad argument");
    if (!p->connected && p->world > 1) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_allreduce_sum: not connected");
    if (n > p->cap) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_allreduce_sum: more doubles than the mailbox was made for");
    if (n == 0) return PMC_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long seq = ++p->seq;
    PutArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int r = 0; r < p->world; ++r) a.box[r] = p->box[r];
    a.src = d_buf; a.n = n; a.cap = p->cap; a.rank = p->rank; a.world = p->world; a.seq = seq;
    hipLaunchKernelGGL(k_p2p_put, dim3((unsigned)p->world), dim3(256), 0, st, a);
    SumArgs s;
    s.box = p->mine; s.out = d_buf; s.n = n; s.cap = p->cap; s.world = p->world; s.seq = seq;
    double secs = 20.0;
    if (const char *e = std::getenv("PMC_P2P_TIMEOUT_S")) secs = std::atof(e);
    s.timeout_ticks = (long long)(secs * 1e8);
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_p2p_sum, dim3((unsigned)(blocks < 64 ? blocks : 64)), dim3(256), 0, st, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(PMC_EHIP, "pmc_p2p_allreduce_sum: launch: %s", hipGetErrorString(e));
    return PMC_OK;
}

int pmc_p2p_status(pmc_p2p *p, void *stream)
{
    if (!p) return pmc_internal_fail(PMC_EINVAL, "pmc_p2p_status: NULL");
    unsigned long long err = 0;
    hipError_t e = hipMemcpyAsync(&err, p->mine + offsetof(Mailbox, error), sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return failf(PMC_EHIP, "pmc_p2p_status: %s", hipGetErrorString(e));
    if (err != 0) {
        char buf[160];
        snprintf(buf, sizeof(buf), "pmc_p2p: rank %d gave up waiting for its peers in round %llu (PMC_P2P_TIMEOUT_S)", p->rank, err);
        return pmc_internal_fail(PMC_EHIP, buf);
    }
    return PMC_OK;
}

int pmc_p2p_destroy(pmc_p2p *p)
{
    if (!p) return PMC_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->world; ++r)
        if (r != p->rank && p->box[r]) (void)hipIpcCloseMemHandle(p->box[r]);
    if (p->mine) (void)hipFree(p->mine);
    delete p;
    return PMC_OK;
}

}  // extern "C"
