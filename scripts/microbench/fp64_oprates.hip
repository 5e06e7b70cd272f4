// Calibration microbenchmark: issue cost of the fp64 vector instructions the exp / soft-max code is made of,
// relative to v_fma_f64 (8 independent chains per wavefront, 4 wavefronts per SIMD).
//   hipcc --offload-arch=gfx950 -O3 fp64_oprates.hip -o fp64_oprates && ./fp64_oprates
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 1 << 14;
constexpr int NACC = 8;

#define KERNEL(name, BODY)                                                             \
    __global__ __launch_bounds__(256) void name(double *out, double a, double b, int n) \
    {                                                                                  \
        double acc[NACC];                                                              \
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i + a; \
        for (int it = 0; it < ITERS; it += 4) {                                        \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                              \
                _Pragma("unroll") for (int i = 0; i < NACC; ++i) { double &v = acc[i]; BODY; } \
        }                                                                              \
        double s = 0;                                                                  \
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) s += acc[i];                  \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                       \
    }

KERNEL(k_fma, v = fma(v, a, b))
KERNEL(k_add, v = v + b)
KERNEL(k_mul, v = v * a)
KERNEL(k_max, v = fmax(v, b); asm volatile("" : "+v"(v)))
KERNEL(k_rndne, asm volatile("v_rndne_f64 %0, %0" : "+v"(v)))
KERNEL(k_ldexp, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v) : "v"(n)))
KERNEL(k_cvt, int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(v)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v) : "v"(t)))
KERNEL(k_rcp, asm volatile("v_rcp_f64 %0, %0" : "+v"(v)))
KERNEL(k_cmpsel, v = (v == 0.0) ? b : v; asm volatile("" : "+v"(v)))
KERNEL(k_frexp, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(v)))
KERNEL(k_i32add, int2 t = *(int2 *)&v; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t.y) : "v"(n)); v = *(double *)&t)
KERNEL(k_maxasm, asm volatile("v_max_f64 %0, %0, %1" : "+v"(v) : "v"(b)))
KERNEL(k_cmponly, asm volatile("v_cmp_eq_f64 vcc, 0, %0" : : "v"(v) : "vcc"); asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(b)))
KERNEL(k_cmpclass, asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(v), "v"(n) : "vcc"); asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(b)))
KERNEL(k_cndonly, int2 t = *(int2 *)&v; asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(t.x), "+v"(t.y) : "v"(n) : ); v = *(double *)&t)
KERNEL(k_icmpsel, int2 t = *(int2 *)&v; int o; asm volatile("v_or_b32 %0, %1, %2" : "=v"(o) : "v"(t.x), "v"(t.y)); asm volatile("v_cmp_eq_u32 vcc, 0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(t.y) : "v"(o), "v"(n) : "vcc"); v = *(double *)&t)
KERNEL(k_i1, int2 t = *(int2 *)&v; asm volatile("v_cmp_eq_u32 vcc, 0, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t.y) : "v"(n) : "vcc"); v = *(double *)&t)
KERNEL(k_i2, int2 t = *(int2 *)&v; asm volatile("v_cmp_eq_u32 vcc, 0, %0\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(t.y), "+v"(t.x) : "v"(n) : "vcc"); v = *(double *)&t)
KERNEL(k_f1, int2 t = *(int2 *)&v; asm volatile("v_cmp_eq_f64 vcc, 0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(t.y) : "v"(v), "v"(n) : "vcc"); v = *(double *)&t)
KERNEL(k_s1, int2 t = *(int2 *)&v; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t.y) : "v"(n)); v = *(double *)&t)
KERNEL(k_e2, int2 t = *(int2 *)&v; asm volatile("v_cmp_eq_u32 s[20:21], 0, %0\n v_cndmask_b32 %0, %0, %2, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]" : "+v"(t.y), "+v"(t.x) : "v"(n) : "s20", "s21"); v = *(double *)&t)
KERNEL(k_f2nop, int2 t = *(int2 *)&v; asm volatile("v_cmp_eq_f64 vcc, 0, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(t.y), "+v"(t.x) : "v"(v), "v"(n) : "vcc"); v = *(double *)&t)
KERNEL(k_divfix, asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(v) : "v"(b)))

template <class K> double run(K kern, double *d_out, const char *name, double ref)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 4;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001, 1e-9, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001, 1e-9, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-28s %8.3f ms   %.2f x v_fma_f64\n", name, ms, ref > 0 ? ms / ref : 1.0);
    return ms;
}

int main()
{
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 256 * 16);
    const double ref = run(k_fma, d_out, "v_fma_f64", 0);
    run(k_add, d_out, "v_add_f64", ref);
    run(k_mul, d_out, "v_mul_f64", ref);
    run(k_max, d_out, "v_max_f64", ref);
    run(k_rndne, d_out, "v_rndne_f64", ref);
    run(k_ldexp, d_out, "v_ldexp_f64", ref);
    run(k_cvt, d_out, "v_cvt_i32_f64 + v_cvt_f64_i32", ref);
    run(k_rcp, d_out, "v_rcp_f64", ref);
    run(k_cmpsel, d_out, "v_cmp_eq_f64 + 2 v_cndmask", ref);
    run(k_frexp, d_out, "v_frexp_mant_f64", ref);
    run(k_i32add, d_out, "v_add_u32", ref);
    run(k_divfix, d_out, "v_div_fixup_f64", ref);
    run(k_maxasm, d_out, "v_max_f64 (asm, no canonicalize)", ref);
    run(k_cmponly, d_out, "v_cmp_eq_f64 + v_add_f64", ref);
    run(k_cmpclass, d_out, "v_cmp_class_f64 + v_add_f64", ref);
    run(k_cndonly, d_out, "2 v_cndmask_b32", ref);
    run(k_i1, d_out, "v_cmp_eq_u32 + 1 v_cndmask", ref);
    run(k_i2, d_out, "v_cmp_eq_u32 + 2 v_cndmask", ref);
    run(k_f1, d_out, "v_cmp_eq_f64 + 1 v_cndmask", ref);
    run(k_f2nop, d_out, "v_cmp_eq_f64 + 2 v_cndmask (asm)", ref);
    run(k_s1, d_out, "1 v_cndmask (vcc not written)", ref);
    run(k_e2, d_out, "v_cmp_eq_u32 sgpr + 2 v_cndmask", ref);
    run(k_icmpsel, d_out, "v_or + v_cmp_eq_u32 + v_cndmask", ref);
    return 0;
}
