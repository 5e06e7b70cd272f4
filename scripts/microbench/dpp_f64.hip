// Microbenchmark: does v_fmac_f64 with a DPP row_newbcast source run at the full fp64 rate on gfx950,
// and does it broadcast lane n of every 16-lane row as expected?  (A coefficient held once per row in
// a VGPR could then feed the FMA like an SGPR operand does -- without the scalar cache.)
//   hipcc --offload-arch=gfx950 -O3 dpp_f64.hip -o dpp_f64 && ./dpp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NACC = 16;
constexpr int ITERS = 4096;

#define FMAC_DPP(acc, coef, d, n) \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(coef), "v"(d))

__global__ __launch_bounds__(256) void k_dpp(double *out, double a)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double coef = 1e-9 * (threadIdx.x & 15), d = a + threadIdx.x * 1e-9;
    for (int it = 0; it < ITERS / 16; ++it) {
#define ROW(n) _Pragma("unroll") for (int i = 0; i < NACC; ++i) FMAC_DPP(acc[i], coef, d, n);
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15)
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_plain(double *out, double a)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double coef = 1e-9 * (threadIdx.x & 15), d = a + threadIdx.x * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(coef), "v"(d));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// semantics: out[lane] = coef[row*16 + 5] * d[lane]
__global__ void k_check(double *out)
{
    double coef = 100.0 + threadIdx.x, d = 2.0, acc = 0.0;
    FMAC_DPP(acc, coef, d, 5);
    out[threadIdx.x] = acc;
}

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int blocks = 256 * 16;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    float ms = timeit([&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, 0, out, 0.999); });
    printf("v_fmac_f64 vgpr           : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 256 * NACC * ITERS / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(256), 0, 0, out, 0.999); });
    printf("v_fmac_f64_dpp row_newbcast: %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 256 * NACC * ITERS / ms * 1e-9);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, out);
    std::vector<double> h(64);
    hipMemcpy(h.data(), out, sizeof(double) * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[l] != 2.0 * (100.0 + (l / 16) * 16 + 5);
    printf("row_newbcast:5 semantics: %s (lane 0 %.0f, lane 17 %.0f, lane 63 %.0f)\n", bad ? "UNEXPECTED" : "ok", h[0], h[17], h[63]);
    return 0;
}
