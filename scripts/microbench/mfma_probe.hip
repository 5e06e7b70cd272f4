#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main()
{
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    std::vector<double> ha(64), hb(64), hd(64);
    for (int which = 0; which < 2; ++which)
        for (int p = 0; p < 64; ++p) {
            for (int i = 0; i < 64; ++i) { ha[i] = which ? 1.0 : (i == p); hb[i] = which ? (i == p) : 1.0; }
            hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
            hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
            printf("%c[%2d] ->", which ? 'B' : 'A', p);
            for (int i = 0; i < 64; ++i) if (hd[i] != 0) printf(" %d", i);
            printf("\n");
        }
    return 0;
}
