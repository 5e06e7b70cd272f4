// Calibration microbenchmark: sustained fp64 rate of v_fma_f64 (VGPR and SGPR operands) and of
// v_mfma_f64_16x16x4_f64 on this chip.  hipcc --offload-arch=gfx950 -O3 fp64_peak.hip -o fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NACC = 16;
constexpr int ITERS = 4096;

__global__ __launch_bounds__(256) void k_fma_vgpr(double *out, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], x, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef __attribute__((address_space(4))) const double cdouble;
__global__ __launch_bounds__(256) void k_fma_sgpr(double *out, const double *coef)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    cdouble *c = (cdouble *)coef;
    for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double cv = c[(it & 7) * 16 + r];
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fma(cv, acc[i], 1e-9);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double *out, double a, double b)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mixed: wave parity selects MFMA-only or VALU-only work -> do the two pipes overlap for fp64?
__global__ __launch_bounds__(512) void k_mixed(double *out, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave & 1) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double av = a + threadIdx.x * 1e-9, bv = b;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
        double x = a + threadIdx.x * 1e-9;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], x, b);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
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
    double *out, *coef;
    hipMalloc(&out, sizeof(double) * blocks * 512);
    hipMalloc(&coef, sizeof(double) * 128);
    std::vector<double> h(128, 0.999999);
    hipMemcpy(coef, h.data(), sizeof(double) * 128, hipMemcpyHostToDevice);
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_fma_vgpr, dim3(blocks), dim3(256), 0, 0, out, 0.999, 1e-9); });
    printf("v_fma_f64 vgpr : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 256 * NACC * ITERS / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k_fma_sgpr, dim3(blocks), dim3(256), 0, 0, out, coef); });
    printf("v_fma_f64 sgpr : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * blocks * 256 * NACC * ITERS / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, 0.999, 1e-9); });
    printf("mfma_f64_16x16x4: %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * 16 * 4 * 4.0 * ITERS * blocks * 4 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(512), 0, 0, out, 0.999, 1e-9); });
    double fl = 4.0 * blocks * (2.0 * 16 * 16 * 4 * 4.0 * ITERS) + 4.0 * blocks * 64 * (2.0 * NACC * ITERS);
    printf("mixed (4 mfma waves + 4 valu waves per block): %.3f ms  %.1f TFLOP/s total\n", ms, fl / ms * 1e-9);
    return 0;
}
