// Calibration microbenchmark: how many independent fp64 multiply-adds a SIMD needs in flight to keep its
// vector pipe busy -- dependent chains per wavefront (NACC) x wavefronts per SIMD (W).  Prints the fraction of
// the 4-cycle issue slots that did work, assuming the nominal clock given on the command line.
//   hipcc --offload-arch=gfx950 -O3 fp64_latency.hip -o fp64_latency && ./fp64_latency [GHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ITERS = 1 << 15;

template <int NACC> __global__ __launch_bounds__(256) void k_chain(double *out, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const double x = a + threadIdx.x * 1e-9;
    for (int it = 0; it < ITERS; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], x, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(double *d_out, int W, double ghz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * W;                 // one 4-wavefront block per SIMD-set and "W"
    hipLaunchKernelGGL(k_chain<NACC>, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_chain<NACC>, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double slots = ms * 1e-3 * ghz * 1e9 / 4.0;            // issue slots per SIMD
    const double instr = (double)W * ITERS * NACC;               // per SIMD
    printf("chains/wave %d  waves/SIMD %d   %.3f ms   busy slots %.2f   cycles per dependent step %.1f\n", NACC, W, ms,
           instr / slots, ms * 1e-3 * ghz * 1e9 / ITERS);
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.1;
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 256 * 16);
    for (int W : {1, 2, 3, 4, 8}) {
        run<1>(d_out, W, ghz);
        run<2>(d_out, W, ghz);
        run<4>(d_out, W, ghz);
        run<8>(d_out, W, ghz);
    }
    return 0;
}
