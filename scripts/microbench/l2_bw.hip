// Calibration: how fast can one CU pull L2-resident data (per-CU L2->L1 bandwidth), as a function
// of bytes per lane and loads in flight.  hipcc --offload-arch=gfx950 -O3 l2_bw.hip -o l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int UNROLL>
__global__ __launch_bounds__(512) void k_read(const T *__restrict__ buf, size_t n_elems, int iters, double *out)
{
    // every workgroup streams the same n_elems (L2 resident after the first pass)
    double acc = 0;
    const size_t stride = 512;
    for (int it = 0; it < iters; ++it) {
        for (size_t base = threadIdx.x; base + (UNROLL - 1) * stride < n_elems; base += UNROLL * stride) {
            T v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = buf[base + u * stride];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += ((const double *)&v[u])[0];
        }
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

template <typename T, int UNROLL> void run(const char *name, const void *buf, size_t bytes, int blocks, double *out)
{
    const int iters = 64;
    size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_read<T, UNROLL>), dim3(blocks), dim3(512), 0, 0, (const T *)buf, n, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_read<T, UNROLL>), dim3(blocks), dim3(512), 0, 0, (const T *)buf, n, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double total = (double)bytes * iters * blocks;
    printf("%-28s region %6zu KB  %8.1f GB/s aggregate  %6.1f B/clk/CU (at 2.3 GHz, %d blocks)\n", name,
           bytes >> 10, total / ms * 1e-6, total / ms * 1e-6 * 1e9 / 2.3e9 / blocks, blocks);
}

int main()
{
    void *buf;
    double *out;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 0, 64 << 20);
    hipMalloc(&out, 8 * 4096);
    for (size_t kb : {40, 640, 2048}) {
        run<double, 4>("8B/lane, 4 in flight", buf, kb << 10, 256, out);
        run<double2, 2>("16B/lane, 2 in flight", buf, kb << 10, 256, out);
        run<double2, 4>("16B/lane, 4 in flight", buf, kb << 10, 256, out);
        run<double2, 8>("16B/lane, 8 in flight", buf, kb << 10, 256, out);
    }
    return 0;
}
