// The per-dimension kernel units of libpmc_hip.so as no-ops, for the sanitizer job (see hip_stub.cpp): every launcher
// the dispatcher's registry names returns success without launching, the geometry functions answer with what the real
// units answer for the shapes the job uses.
#include "../../pypmc_amd/csrc/pmc_dims.h"
#include "../../pypmc_amd/csrc/pmc_internal.h"

#define STUB_UNIT(d, p)                                                                                                \
    extern "C" hipError_t pmc_launch_logpdf_d##d##_p##p(int, int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; } \
    extern "C" hipError_t pmc_launch_resp_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; }       \
    extern "C" hipError_t pmc_launch_resp_groups_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; } \
    extern "C" hipError_t pmc_launch_logpdf_split_d##d##_p##p(int, int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; } \
    extern "C" hipError_t pmc_launch_logpdf2_d##d##_p##p(const PmcArgsA &, unsigned, hipStream_t) { return hipErrorNotSupported; } \
    extern "C" hipError_t pmc_launch_resp_groups_split_d##d##_p##p(int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; } \
    extern "C" hipError_t pmc_launch_stats_d##d##_p##p(const PmcArgsB &, unsigned, hipStream_t) { return hipSuccess; }           \
    extern "C" void pmc_stats_config_d##d##_p##p(int *nsub, int *waves) { *nsub = d >= 40 ? 2 : 1; *waves = 16; }               \
    extern "C" hipError_t pmc_launch_propose_d##d##_p##p(const PmcArgsP &, unsigned, hipStream_t) { return hipSuccess; }         \
    extern "C" hipError_t pmc_launch_fused_d##d##_p##p(int, int, const PmcArgsF &, unsigned, hipStream_t) { return hipSuccess; } \
    extern "C" int pmc_fused_lds_bytes_d##d##_p##p(int, int) { return 0; }                                                      \
    extern "C" hipError_t pmc_launch_stats_gemm_d##d##_p##p(const PmcArgsG &, unsigned, hipStream_t) { return hipSuccess; }      \
    extern "C" void pmc_stats_gemm_config_d##d##_p##p(int *cols, int *slices, int *msp, int *wgs)                               \
    {                                                                                                                           \
        *cols = d >= 8 ? 15 : 0; *slices = 1; *msp = (((d + 1) * (d + 2) / 2 + 15) / 16) * 16; *wgs = 1;                        \
    }
#define STUB_MG(d)                                                                                                              \
    extern "C" hipError_t pmc_launch_mgemm_d##d##_p0(int, const PmcArgsQ &, unsigned, hipStream_t) { return hipSuccess; }        \
    extern "C" hipError_t pmc_launch_theta_d##d##_p0(const double *, int, int, int, double *, double *, double *,               \
                                                     unsigned long long *, hipStream_t) { return hipSuccess; }                  \
    extern "C" void pmc_mgemm_config_d##d##_p0(int *nstepp, int *nct)                                                           \
    {                                                                                                                           \
        const int q = d / 4, nstep = q * (2 * q + 1) + q + 1;                                                                   \
        const bool on = d == 32 || d == 40 || d == 48;                                                                          \
        *nstepp = on ? (nstep + 15) / 16 * 16 : 0; *nct = on ? (d <= 40 ? 4 : 2) : 0;                                           \
    }
#define STUB_X(d) STUB_UNIT(d, 0) STUB_MG(d)
#define STUB_XP(d) STUB_UNIT(d, 0) STUB_UNIT(d, 1) STUB_MG(d)
PMC_DIM_LIST(STUB_X, STUB_XP)

extern "C" hipError_t pmc_launch_resp_tiles(int, const PmcArgsT &, unsigned, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pmc_launch_dof_sums(const PmcArgsV &, unsigned, unsigned, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pmc_launch_logpdf_d0_p0(int, int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pmc_launch_resp_d0_p0(int, const PmcArgsA &, unsigned, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pmc_launch_big_maha(const PmcArgsM &, hipStream_t) { return hipSuccess; }
extern "C" hipError_t pmc_launch_big_stats(const PmcArgsB &, unsigned, hipStream_t) { return hipSuccess; }
extern "C" void pmc_big_stats_config(int, int *nsub, int *waves) { *nsub = 3; *waves = 8; }
extern "C" hipError_t pmc_launch_propose_big(const PmcArgsP &, unsigned, hipStream_t) { return hipSuccess; }
