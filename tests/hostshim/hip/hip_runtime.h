// TEST INFRASTRUCTURE (tests/geod_host.cpp): stands in for <hip/hip_runtime.h> when the float64 device routines of
// opendrift_amd/csrc/odr_geodesic.hip.h are compiled for the CPU with g++, so that their arithmetic can be checked without a GPU.
// The hardware reciprocal / reciprocal-square-root seeds become the exact operations (the Newton steps behind them then change
// nothing); everything else in that header is plain C++.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __constant__ const
#define __forceinline__ inline
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
using std::fabs; using std::fma; using std::fmax; using std::fmin; using std::rint; using std::copysign; using std::signbit;
using std::sqrt; using std::atan2; using std::frexp; using std::isfinite;
