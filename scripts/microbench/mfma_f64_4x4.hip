// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout (probed, the guides do not list it)
// and sustained rate of the instruction mix the statistics kernel would issue: 15 independent
// MFMAs + NV fp64 VALU operations per (16 samples, component).
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_4x4.hip -o mfma_f64_4x4 && ./mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

constexpr int ITERS = 2000;

template <int NV>
__global__ __launch_bounds__(256) void k_rate(double *out, const double *in)
{
    double acc[15], x[5], m1[5];
    const double u = in[threadIdx.x & 63], mu = in[64 + (threadIdx.x & 3)];
    for (int i = 0; i < 15; ++i) acc[i] = 0.0;
    for (int i = 0; i < 5; ++i) { x[i] = in[128 + i * 64 + (threadIdx.x & 63)]; m1[i] = 0.0; }
    double s0 = 0.0;
    for (int it = 0; it < ITERS; ++it) {
        double d[5], a[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (NV >= 5) d[i] = x[i] - mu; else d[i] = x[i];
            if (NV >= 10) a[i] = d[i] * u; else a[i] = d[i];
            if (NV >= 15) m1[i] += a[i];
        }
        if (NV >= 16) s0 += u;
        int t = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j, ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], d[j], acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(x[i]));
    }
    double r = s0;
    for (int i = 0; i < 15; ++i) r += acc[i];
    for (int i = 0; i < 5; ++i) r += m1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&d, 64 * 8);
    std::vector<double> ha(64), hb(64), hd(64);
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + (rand() % 97); hb[i] = 1 + (rand() % 89); }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
    // candidates: A lane = 16*blk + (ak ? 4*k+i : 4*i+k), B lane likewise, D lane = 16*blk + (dt ? 4*j+i : 4*i+j)
    for (int ak = 0; ak < 2; ++ak) for (int bk = 0; bk < 2; ++bk) for (int dt = 0; dt < 2; ++dt) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k)
                s += ha[16 * blk + (ak ? 4 * k + i : 4 * i + k)] * hb[16 * blk + (bk ? 4 * k + j : 4 * j + k)];
            if (s != hd[16 * blk + (dt ? 4 * j + i : 4 * i + j)]) ok = false;
        }
        if (ok) printf("layout: A[i][k] in lane 16b+%s, B[k][j] in lane 16b+%s, D[i][j] in lane 16b+%s\n",
                       ak ? "4k+i" : "4i+k", bk ? "4k+j" : "4j+k", dt ? "4j+i" : "4i+j");
    }
    const int blocks = 256 * 8;
    double *out, *in;
    hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&in, 1024 * 8);
    hipMemset(in, 0, 1024 * 8);
    const double mf = 15.0 * 256 * 2;   // flops per wave-iteration in the MFMAs
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, in); });
    printf("15 mfma only            : %.3f ms  %.1f TFLOP/s (mfma)  %.1f clk/iter/SIMD-wave\n", ms,
           mf * ITERS * blocks * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / ITERS / (blocks * 4 / 1024.0));
    ms = timeit([&] { hipLaunchKernelGGL(k_rate<16>, dim3(blocks), dim3(256), 0, 0, out, in); });
    printf("15 mfma + 16 fp64 valu  : %.3f ms  %.1f TFLOP/s (mfma)  %.1f clk/iter/SIMD-wave\n", ms,
           mf * ITERS * blocks * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / ITERS / (blocks * 4 / 1024.0));
    ms = timeit([&] { hipLaunchKernelGGL(k_rate<10>, dim3(blocks), dim3(256), 0, 0, out, in); });
    printf("15 mfma + 10 fp64 valu  : %.3f ms  %.1f TFLOP/s (mfma)  %.1f clk/iter/SIMD-wave\n", ms,
           mf * ITERS * blocks * 4 / ms * 1e-9, ms * 1e-3 * 2.4e9 / ITERS / (blocks * 4 / 1024.0));
    return 0;
}
