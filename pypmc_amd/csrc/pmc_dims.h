// Sample dimensions the kernels are compiled for.  One translation unit is built per entry
// (pypmc_amd/build.py parses this list); pmc_api.hip dispatches on it.
//   X(D)   exact-dimension kernels (compile-time row stride, fully unrolled)
//   XP(D)  additionally a "padded" variant: any D' with prev < D' < D runs on it with zero padding
#pragma once
#define PMC_DIM_LIST(X, XP) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) \
    XP(10) XP(12) XP(16) XP(20) XP(24) XP(30) XP(32) XP(40) XP(48) XP(64)
#define PMC_MAX_DIM 64
