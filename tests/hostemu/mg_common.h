// TEST INFRASTRUCTURE ONLY -- stands in for michigan_amd/csrc/mg_common.h + <hip/hip_runtime.h> when
// tests/test_inputs_hostemu.py compiles the *unchanged* kernel source mg_inputs.hip for the host CPU with g++:
// one "thread" per block, blocks executed one after the other, IEEE intrinsics as plain operations
// (-ffp-contract=off).  It lets the CPU-only container check the kernels' arithmetic bit for bit against the oracle;
// the parallel decomposition (workgroup scan, grid-stride coverage) is what the -m gpu tests add.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include "michigan_hip.h"

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static dim3 blockIdx(0), blockDim(1), threadIdx(0), gridDim(1);
typedef void* hipStream_t;
#define __global__
#define __device__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static
#define __launch_bounds__(x)
#define MG_HOLE_THREADS 1
static inline void __syncthreads() {}
using std::min;
using std::max;
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) do { \
    const dim3 g_ = (grid); gridDim = g_; blockDim = dim3(1); threadIdx = dim3(0); \
    for (unsigned b_ = 0; b_ < g_.x; ++b_) { blockIdx = dim3(b_); kern(__VA_ARGS__); } } while (0)

static char g_mg_err[512];
static inline int mg_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_mg_err, sizeof(g_mg_err), fmt, ap); va_end(ap); return code;
}
#define MG_CHECK_ARG(cond, ...) do { if (!(cond)) return mg_fail(MG_ERR_ARG, __VA_ARGS__); } while (0)
#define MG_CHECK_LAUNCH(name) do { } while (0)
