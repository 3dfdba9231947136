// Shared device/host helpers for libmichigan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "michigan_hip.h"

#ifndef MG_F32_ONE_CHAIN
#define MG_F32_ONE_CHAIN 0   // 1: fp32 MFMA sums as ONE sequential chain per output (rounds 1-3) instead of two-level sums: A/B builds only
#endif

typedef __attribute__((ext_vector_type(8)))  __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4)))  short  s16x4_t;
typedef __attribute__((ext_vector_type(4)))  float  f32x4_t;
typedef __attribute__((ext_vector_type(16))) float  f32x16_t;

// ---- error plumbing (thread-local message, C ABI returns an int) ----------
extern thread_local char g_mg_err[512];
static inline int mg_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_mg_err, sizeof(g_mg_err), fmt, ap);
    va_end(ap);
    return code;
}
#define MG_CHECK_ARG(cond, ...) do { if (!(cond)) return mg_fail(MG_ERR_ARG, __VA_ARGS__); } while (0)
#define MG_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return mg_fail(MG_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even (matches torch .to(bfloat16)): gfx950's v_cvt_pk_bf16_f32, two values per
// instruction (the integer rounding sequence it replaces was ~6 VALU per value, the bulk of the conv epilogues)
typedef __attribute__((ext_vector_type(2))) float  mg_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 mg_bf16x2_t;
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    const mg_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_bf16x2_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(f2bf2(f, 0.f) & 0xffffu); }

// Element traits: a "quad" is 4 consecutive channels of one pixel.
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int DT = MG_F32;
    __device__ static __forceinline__ f32x4_t load4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
    __device__ static __forceinline__ void store4(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
    __device__ static __forceinline__ float load1(const float* p) { return *p; }
    __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct ET<uint16_t> {
    static constexpr int DT = MG_BF16;
    __device__ static __forceinline__ f32x4_t load4(const uint16_t* p) {
        uint2 u = *reinterpret_cast<const uint2*>(p);
        f32x4_t v;
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        return v;
    }
    __device__ static __forceinline__ void store4(uint16_t* p, f32x4_t v) {
        uint2 u;
        u.x = f2bf2(v[0], v[1]);
        u.y = f2bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    }
    __device__ static __forceinline__ float load1(const uint16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void store1(uint16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float mg_act(float v, int act, float slope) {
    switch (act) {
        case MG_ACT_RELU:  return v > 0.f ? v : 0.f;
        case MG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case MG_ACT_TANH:  return tanhf(v);
        default:           return v;
    }
}
// derivative expressed through the OUTPUT y (sign-preserving activations)
__device__ __forceinline__ float mg_act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case MG_ACT_RELU:  return y > 0.f ? 1.f : 0.f;
        case MG_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case MG_ACT_TANH:  return 1.f - y * y;
        default:           return 1.f;
    }
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize has to be raised once per (kernel, DEVICE): a process that drives several GPUs
// (not the one-process-per-GPU layout of this package, but legal) must not skip the call on its second device.
#include <mutex>
static inline void mg_raise_lds_cap(const void* kern, int bytes)
{
    static std::mutex mu;
    static struct { const void* k; int dev; int bytes; } seen[256];
    static int nseen = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < nseen; ++i)
        if (seen[i].k == kern && seen[i].dev == dev && seen[i].bytes >= bytes) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (nseen < 256) { seen[nseen].k = kern; seen[nseen].dev = dev; seen[nseen].bytes = bytes; ++nseen; }
}
