// mg_norm.hip -- per-channel statistics and normalisation forward/backward
// (sync-BN inside SPADE, InstanceNorm2d in the discriminator / appearance encoder).
//
// All kernels are HBM-bound streaming passes over NHWC activations: a thread
// owns one quad (4 consecutive channels) so a pixel row of C channels is read
// by C/4 consecutive lanes as 8 B (bf16) / 16 B (f32) vector loads; the
// reduction over pixels is two-stage and deterministic:
//   stage 1  grid (chunks, G): register accumulation over the block's pixel
//            chunk, LDS tree across the thread rows -> partial[g][chunk][2][C]
//   stage 2  fp64 sum over chunks in a fixed order -> sums[g][2][C]
// Forward statistics (sum x, sum x^2) are conditioned like a two-pass variance: stage 1 accumulates the SHIFTED values x - k[g][c] with
// the pivot k = x[g][pixel 0][c] (a sample of the channel: sum (x-k)^2 ~ n (var + (mean-k)^2) carries no mean^2 / var cancellation into
// the fp32 partials), stage 2 adds the partials in fp64, un-shifts in fp64 (sum x = s + n k, sum x^2 = ss + 2 k s + n k^2) and hands FP64
// sums to the cross-rank all-reduce and to finalize: nothing between the per-thread partials and var = E[x^2] - E[x]^2 is rounded to
// fp32 (the one-pass fp32 sums were; sync_batchnorm/batchnorm.py:128-145 is that formula, batchnorm_reimpl.py:18-74 the
// two-pass yardstick).  `shift = 0` (plain column sums of gradients: ops.channel_sums for bias gradients) keeps k = 0: for ZERO-mean data a
// sample pivot makes the partial sums grow like n |k| instead of sqrt(n) sigma.  Per-thread fp64 accumulation (no pivot needed) was
// measured too: same results, stats_stage1_vec 17.8 -> 21.7 us per launch (+0.24 ms per step), not kept.
#include "mg_common.h"
#include <type_traits>

int g_mg_norm_bwd_vec = 1;        // mg_set_option(19, v): 0 = the norm backward reduction stays on the 8-byte quad kernel

namespace {

constexpr int NTHR = 256;

struct StatGeom { int tpr; int rpb; int nchunks; int64_t chunk; };

static inline StatGeom stat_geom(int G, int64_t P, int C)
{
    StatGeom g;
    const int c4 = C / 4;
    g.tpr = c4 < NTHR ? c4 : NTHR;
    g.rpb = NTHR / g.tpr;
    // enough stage-1 blocks to fill 256 CUs a few times over (G groups share the budget), but at
    // least 16 rows per thread and at most 512 partials per group for the stage-2 tree.
    int64_t want = (1536 + G - 1) / G;
    const int64_t cap = (P + (int64_t)g.rpb * 16 - 1) / ((int64_t)g.rpb * 16);
    if (want > cap) want = cap;
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    g.chunk = (P + want - 1) / want;
    g.nchunks = (int)((P + g.chunk - 1) / g.chunk);
    return g;
}

// MODE 0: plain (sum x, sum x^2).  MODE 1: norm backward (sum dxhat, sum dxhat*xhat [+ dgb]).
template <typename T, int MODE, bool HAS_H = true>
__global__ __launch_bounds__(NTHR) void reduce_stage1(
    const T* __restrict__ x, const T* __restrict__ dh, const T* __restrict__ h, const T* __restrict__ g1,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dgb,
    float* __restrict__ partial, int64_t P, int C, int tpr, int rpb, int64_t chunk, int act, float slope,
    int up = 0, int H = 0, int W = 0)
{
    __shared__ float red[NTHR * 8];
    const int tid = threadIdx.x;
    const int g = blockIdx.y, ck = blockIdx.x, nchunks = gridDim.x;
    const int c4 = C / 4;
    const bool active = tid < tpr * rpb;
    const int tq = tid % tpr, tr = tid / tpr;
    const int Cr2 = 2 * ((C + 31) / 32) * 32;

    const int64_t p0 = (int64_t)ck * chunk;
    const int64_t p1 = (p0 + chunk < P) ? p0 + chunk : P;

    for (int qd0 = 0; qd0 < c4; qd0 += tpr) {         // uniform trip count (one trip unless C > 1024)
        const int qd = qd0 + tq;
        const bool qv = active && qd < c4;
        const int c = (qd < c4 ? qd : 0) * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4_t mu, rs;
        f32x4_t kv = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0 && up) kv = ET<T>::load4(x + (size_t)g * P * C + c);     // MODE 0: `up` carries the shift flag -- pivot = the group's first pixel
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = mean[(size_t)g * C + c + j]; rs[j] = rstd[(size_t)g * C + c + j]; }
        }
        if (qv) {
            for (int64_t p = p0 + tr; p < p1; p += rpb) {
                const size_t o = ((size_t)g * P + p) * C + c;
                size_t ox = o;
                if (MODE == 1 && up) {                       // x is the half-resolution source of a nearest 2x upsample (G == 1)
                    const int pp = (int)p, n = pp / (H * W), rem = pp - n * (H * W);
                    const int yy = rem / W, xx = rem - yy * W;
                    ox = ((size_t)(n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C + c;
                }
                const f32x4_t xv = ET<T>::load4(x + ox);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float t = xv[j] - kv[j]; s[j] += t; ss[j] += t * t; }
                } else {
                    const f32x4_t dv = ET<T>::load4(dh + o);
                    f32x4_t hv = {1.f, 1.f, 1.f, 1.f};              // h only matters through the sign of the activation's output
                    if constexpr (HAS_H) hv = ET<T>::load4(h + o);
                    f32x4_t gv = {1.f, 1.f, 1.f, 1.f};
                    if (g1) gv = ET<T>::load4(g1 + o);
                    f32x4_t dgam, dbet;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dpre = dv[j] * mg_act_grad_from_out(hv[j], act, slope);
                        const float xh = (xv[j] - mu[j]) * rs[j];
                        const float dxh = dpre * gv[j];
                        s[j] += dxh; ss[j] += dxh * xh;
                        dgam[j] = dpre * xh; dbet[j] = dpre;
                    }
                    if (dgb) {
                        const size_t ob = ((size_t)g * P + p) * Cr2 + (size_t)(c >> 5) * 64 + (c & 31);
                        ET<T>::store4(dgb + ob, dgam);
                        ET<T>::store4(dgb + ob + 32, dbet);
                    }
                }
            }
        }
        // cross-row reduce through LDS (rows tr = 0..rpb-1 share quad tq)
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = ss[j]; }
        __syncthreads();
        if (qv && tr == 0) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = red[tid * 8 + j];
            for (int r = 1; r < rpb; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += red[(tid + r * tpr) * 8 + j];
            }
            float* dst = partial + ((size_t)g * nchunks + ck) * 2 * C;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[c + j] = a[j]; dst[C + c + j] = a[4 + j]; }
        }
        __syncthreads();
    }
}

// stage 2: one 256-thread block per 32 consecutive outputs; 8 thread rows split the chunks,
// fp64 accumulation, fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_stage2(const float* __restrict__ partial, float* __restrict__ sums, int nchunks, int C2)
{
    __shared__ double red[256];
    const int cl = threadIdx.x & 31, kk = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    const int g = blockIdx.y;
    double a = 0.0;
    if (i < C2) {
        const float* p = partial + (size_t)g * nchunks * C2 + i;
        // eight independent loads in flight per thread (the trip count is a runtime value: without the explicit batch the loop was a
        // chain of ~64 dependent L2 round trips, 10.6 us per launch, 99 launches per step); same summation order as before
        int k = kk;
        for (; k + 56 < nchunks; k += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(k + 8 * j) * C2];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (double)v[j];
        }
        for (; k < nchunks; k += 8) a += (double)p[(size_t)k * C2];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (kk == 0 && i < C2) {
#pragma unroll
        for (int r = 1; r < 8; ++r) a += red[r * 32 + cl];
        sums[(size_t)g * C2 + i] = (float)a;
    }
}

template <typename T>
__global__ void norm_act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t nquads, int64_t P, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, int act, float slope,
                                    const T* __restrict__ resid)
{
    const int c4 = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * blockDim.x) {
        const int qd = (int)(i % c4);
        const int64_t pix = i / c4;
        const int g = (int)(pix / P);
        const int c = qd * 4;
        const f32x4_t xv = ET<T>::load4(x + i * 4);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = mg_act((xv[j] - mean[(size_t)g * C + c + j]) * rstd[(size_t)g * C + c + j], act, slope);
        if (resid) {
            const f32x4_t rv = ET<T>::load4(resid + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += rv[j];
        }
        ET<T>::store4(y + i * 4, o);
    }
}

template <typename T>
__global__ void norm_bwd_apply_kernel(const T* __restrict__ dh, const T* __restrict__ h, const T* __restrict__ x,
                                      const T* __restrict__ g1, T* __restrict__ dx, int64_t nquads, int64_t P, int C,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ s1, const float* __restrict__ s2, int sgs, float sscale, int act, float slope)
{
    const int c4 = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * blockDim.x) {
        const int qd = (int)(i % c4);
        const int64_t pix = i / c4;
        const int g = (int)(pix / P);
        const size_t sc = (size_t)g * C + qd * 4, ss = (size_t)g * sgs + qd * 4;
        const f32x4_t dv = ET<T>::load4(dh + i * 4);
        f32x4_t hv = {1.f, 1.f, 1.f, 1.f};
        if (h) hv = ET<T>::load4(h + i * 4);
        const f32x4_t xv = ET<T>::load4(x + i * 4);
        f32x4_t gv = {1.f, 1.f, 1.f, 1.f};
        if (g1) gv = ET<T>::load4(g1 + i * 4);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = rstd[sc + j];
            const float xh = (xv[j] - mean[sc + j]) * r;
            const float dxh = dv[j] * mg_act_grad_from_out(hv[j], act, slope) * gv[j];
            o[j] = r * (dxh - s1[ss + j] * sscale - xh * (s2[ss + j] * sscale));
        }
        ET<T>::store4(dx + i * 4, o);
    }
}


// ---------------------------------------------------------------------------------------------
// Channel-resident 16-byte variants (the common geometries: C a multiple of VEC = 16 B / sizeof(T) and C / VEC a
// divisor of 256).  A thread keeps ONE group of VEC channels for the whole launch, so the per-channel constants
// are loaded once, the pixel loop has no integer division, every access is a full 16-byte vector and PIX
// independent pixels are in flight per thread.  The quad kernels above (8-byte accesses, div/mod and 16 scalar
// parameter loads per quad) ran at 2.0-2.5 TB/s on [8,512,512,128] bf16; these run at 4-5 TB/s (tools/bench_pointwise.py).
template <typename T> struct VT;
template <> struct VT<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = t[j];
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        f32x4_t t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = v[j];
        *reinterpret_cast<f32x4_t*>(p) = t;
    }
};
template <> struct VT<uint16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void load(const uint16_t* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    }
    __device__ static __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
        uint4 u;
        u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]); u.z = f2bf2(v[4], v[5]); u.w = f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = u;
    }
};

template <typename T> static inline bool vec_geom_ok(int C)
{
    constexpr int VEC = VT<T>::VEC;
    if (C % VEC) return false;
    const int cv = C / VEC;
    return cv <= NTHR && NTHR % cv == 0;
}

// derivative factor of NONE / RELU / LRELU through the output: y > 0 ? 1 : neg
__device__ __forceinline__ float act_factor(float y, float neg) { return y > 0.f ? 1.f : neg; }

template <typename T, int PIX>
__global__ __launch_bounds__(NTHR) void norm_bwd_apply_vec(const T* __restrict__ dh, const T* __restrict__ h, const T* __restrict__ x,
                                                         const T* __restrict__ g1, T* __restrict__ dx, int64_t P, int C,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ s1, const float* __restrict__ s2, int sgs, float sscale, float neg)
{
    constexpr int VEC = VT<T>::VEC;
    const int cv = C / VEC, rows = NTHR / cv;
    const int tq = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int g = blockIdx.y, c = tq * VEC;
    float m[VEC], r[VEC], c1[VEC], c2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const size_t i = (size_t)g * C + c + j, k = (size_t)g * sgs + c + j;
        m[j] = mean[i]; r[j] = rstd[i]; c1[j] = r[j] * (s1[k] * sscale); c2[j] = r[j] * r[j] * (s2[k] * sscale);
    }
    const size_t base = (size_t)g * P * C + c;
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t p0 = (int64_t)blockIdx.x * rows + tr; p0 < P; p0 += step * PIX) {
        float dv[PIX][VEC], hv[PIX][VEC], xv[PIX][VEC], gv[PIX][VEC];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = p0 + k * step;
            if (p < P) {
                const size_t o = base + (size_t)p * C;
                VT<T>::load(dh + o, dv[k]); VT<T>::load(x + o, xv[k]);
                if (h) VT<T>::load(h + o, hv[k]);
                if (g1) VT<T>::load(g1 + o, gv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = p0 + k * step;
            if (p < P) {
                float o4[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float dxh = h ? dv[k][j] * act_factor(hv[k][j], neg) : dv[k][j];
                    if (g1) dxh *= gv[k][j];
                    o4[j] = r[j] * dxh - c1[j] - c2[j] * (xv[k][j] - m[j]);
                }
                VT<T>::store(dx + base + (size_t)p * C, o4);
            }
        }
    }
}

template <typename T, int PIX>
__global__ __launch_bounds__(NTHR) void norm_act_fwd_vec(const T* __restrict__ x, T* __restrict__ y, int64_t P, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float neg, bool relu, const T* __restrict__ resid)
{
    constexpr int VEC = VT<T>::VEC;
    const int cv = C / VEC, rows = NTHR / cv;
    const int tq = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int g = blockIdx.y, c = tq * VEC;
    float m[VEC], r[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m[j] = mean[(size_t)g * C + c + j]; r[j] = rstd[(size_t)g * C + c + j]; }
    const size_t base = (size_t)g * P * C + c;
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t p0 = (int64_t)blockIdx.x * rows + tr; p0 < P; p0 += step * PIX) {
        float xv[PIX][VEC], rv[PIX][VEC];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = p0 + k * step;
            if (p < P) { VT<T>::load(x + base + (size_t)p * C, xv[k]); if (resid) VT<T>::load(resid + base + (size_t)p * C, rv[k]); }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = p0 + k * step;
            if (p < P) {
                float o4[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float v = (xv[k][j] - m[j]) * r[j];
                    const float t = v > 0.f ? v : v * neg;
                    o4[j] = relu ? fmaxf(v, 0.f) : t;
                    if (resid) o4[j] += rv[k][j];
                }
                VT<T>::store(y + base + (size_t)p * C, o4);
            }
        }
    }
}

// stage 1 of the plain statistics (sum x, sum x^2) with the same chunking / partial layout as reduce_stage1<T, 0>
template <typename T, int PIX>
__global__ __launch_bounds__(NTHR) void stats_stage1_vec(const T* __restrict__ x, float* __restrict__ partial, int64_t P, int C, int64_t chunk, int shift)
{
    constexpr int VEC = VT<T>::VEC;
    __shared__ float red[NTHR * 2 * VEC];
    const int cv = C / VEC, rows = NTHR / cv;
    const int tq = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int g = blockIdx.y, ck = blockIdx.x, nchunks = gridDim.x, c = tq * VEC;
    const int64_t p0 = (int64_t)ck * chunk;
    const int64_t p1 = (p0 + chunk < P) ? p0 + chunk : P;
    float s[VEC], ss[VEC], kv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; ss[j] = 0.f; }
    const size_t base = (size_t)g * P * C + c;
    if (shift) VT<T>::load(x + base, kv);                                // pivot of the shifted sums: the group's first pixel
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) kv[j] = 0.f;
    }
    for (int64_t pp = p0 + tr; pp < p1; pp += (int64_t)rows * PIX) {
        float xv[PIX][VEC];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = pp + (int64_t)k * rows;
            if (p < p1) VT<T>::load(x + base + (size_t)p * C, xv[k]);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) xv[k][j] = kv[j];           // contributes (k - k) = 0
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float t = xv[k][j] - kv[j]; s[j] += t; ss[j] += t * t; }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[threadIdx.x * 2 * VEC + j] = s[j]; red[threadIdx.x * 2 * VEC + VEC + j] = ss[j]; }
    __syncthreads();
    if (tr == 0) {
        float* dst = partial + ((size_t)g * nchunks + ck) * 2 * C;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float a = s[j], b = ss[j];
            for (int rr = 1; rr < rows; ++rr) {                              // fixed order: deterministic
                a += red[(threadIdx.x + rr * cv) * 2 * VEC + j];
                b += red[(threadIdx.x + rr * cv) * 2 * VEC + VEC + j];
            }
            dst[c + j] = a; dst[C + c + j] = b;
        }
    }
}

// stage 1 of the norm backward reduction (sum dxhat, sum dxhat * xhat, optional d[gamma|beta] output), 16-byte variant of
// reduce_stage1<T, 1>: a thread keeps VEC channels (mean / rstd loaded once), PIX pixels in flight, every access a full 16-byte
// vector -- the quad kernel moved its four input streams in 8-byte pieces at 4.4 TB/s and was the largest non-MFMA kernel of the step.
// Same chunking and partial layout; `act` is NONE / RELU / LRELU only (neg = 1 / 0 / slope), TANH stays on the quad kernel.
template <typename T, int PIX, bool HAS_H, bool UP>
__global__ __launch_bounds__(NTHR) void bwd_stage1_vec(const T* __restrict__ x, const T* __restrict__ dh, const T* __restrict__ h, const T* __restrict__ g1,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dgb,
                                                     float* __restrict__ partial, int64_t P, int C, int64_t chunk, float neg, int H, int W)
{
    constexpr int VEC = VT<T>::VEC;
    __shared__ float red[NTHR * 2 * VEC];
    const int cv = C / VEC, rows = NTHR / cv;
    const int tq = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int g = blockIdx.y, ck = blockIdx.x, nchunks = gridDim.x, c = tq * VEC;
    const int Cr2 = 2 * ((C + 31) / 32) * 32;
    const int64_t p0 = (int64_t)ck * chunk;
    const int64_t p1 = (p0 + chunk < P) ? p0 + chunk : P;
    float mu[VEC], rs[VEC], s[VEC], ss[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = mean[(size_t)g * C + c + j]; rs[j] = rstd[(size_t)g * C + c + j]; s[j] = 0.f; ss[j] = 0.f; }
    const size_t base = (size_t)g * P * C + c;
    const size_t gbase = (size_t)g * P * Cr2 + (size_t)(c >> 5) * 64 + (c & 31);
    for (int64_t pp = p0 + tr; pp < p1; pp += (int64_t)rows * PIX) {
        float xv[PIX][VEC], dv[PIX][VEC], hv[PIX][VEC], gv[PIX][VEC];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = pp + (int64_t)k * rows;
            if (p < p1) {
                size_t ox = base + (size_t)p * C;
                if (UP) {                                        // x is the half-resolution source of a nearest 2x upsample (G == 1)
                    const int q = (int)p, n = q / (H * W), rem = q - n * (H * W);
                    const int yy = rem / W, xx = rem - yy * W;
                    ox = ((size_t)(n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C + c;
                }
                VT<T>::load(x + ox, xv[k]);
                VT<T>::load(dh + base + (size_t)p * C, dv[k]);
                if (HAS_H) VT<T>::load(h + base + (size_t)p * C, hv[k]);
                if (g1) VT<T>::load(g1 + base + (size_t)p * C, gv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const int64_t p = pp + (int64_t)k * rows;
            if (p < p1) {
                float dgam[VEC], dbet[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float dpre = HAS_H ? dv[k][j] * act_factor(hv[k][j], neg) : dv[k][j];
                    const float xh = (xv[k][j] - mu[j]) * rs[j];
                    const float dxh = g1 ? dpre * gv[k][j] : dpre;
                    s[j] += dxh; ss[j] += dxh * xh;
                    dgam[j] = dpre * xh; dbet[j] = dpre;
                }
                if (dgb) {
                    VT<T>::store(dgb + gbase + (size_t)p * Cr2, dgam);
                    VT<T>::store(dgb + gbase + (size_t)p * Cr2 + 32, dbet);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[threadIdx.x * 2 * VEC + j] = s[j]; red[threadIdx.x * 2 * VEC + VEC + j] = ss[j]; }
    __syncthreads();
    if (tr == 0) {
        float* dst = partial + ((size_t)g * nchunks + ck) * 2 * C;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float a = s[j], b = ss[j];
            for (int rr = 1; rr < rows; ++rr) {                              // fixed order: deterministic
                a += red[(threadIdx.x + rr * cv) * 2 * VEC + j];
                b += red[(threadIdx.x + rr * cv) * 2 * VEC + VEC + j];
            }
            dst[c + j] = a; dst[C + c + j] = b;
        }
    }
}

static inline int pix_grid(int64_t P, int rows, int pix, int G)
{
    int64_t b = (P + (int64_t)rows * pix - 1) / ((int64_t)rows * pix);
    const int64_t cap = (4096 + G - 1) / G;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

// sums[g][2][C] (+ element count) -> mean / rstd (fp64 inside), optional running-statistics update (G == 1):
// replaces a dozen [C]-sized eager ops per normalisation layer.
__global__ void norm_finalize_kernel(const double* __restrict__ sums, int G, int C, double count, float eps, float momentum,
                                     float* __restrict__ running_mean, float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ rstd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    const double s = sums[((size_t)g * 2) * C + c], ss = sums[((size_t)g * 2 + 1) * C + c];
    const double m = s / count;
    double var = ss / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

static inline int ew_grid(int64_t n) { int64_t b = (n + NTHR - 1) / NTHR; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// Input gradient of batch norm for up to TWO consumers that normalise the same x (SPADE norm_0 / norm_s of a block with a learned
// shortcut, architecture.py:68-70,79), optionally through a nearest 2x upsample of x (generator.py:166-207):
//   dx[src] = sum over branches b and over the (1 | 2x2) full-resolution pixels q of  rstd * (dxhat_b[q] - s1_b - xhat[src] * s2_b),
//   dxhat_b = dh_b * act_b'(h_b) * g1_b,  s1_b / s2_b = sums_b[0 / 1] * inv_count.
// One pass instead of two applies + autograd's add of their results + the upsample's 2x2 adjoint: x is read once at its own
// resolution and dx written once at its own resolution.  Channel-resident threads like norm_bwd_apply_vec.
template <typename T, int NB, bool UP>
__global__ __launch_bounds__(NTHR) void norm_bwd_apply2_vec(const mg_norm_apply2_desc d, float neg0, float neg1)
{
    constexpr int VEC = VT<T>::VEC;
    const int C = d.C, cv = C / VEC, rows = NTHR / cv;
    const int tq = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int c = tq * VEC;
    const T* dh0 = (const T*)d.dh[0]; const T* h0 = (const T*)d.h[0]; const T* g0 = (const T*)d.g1[0];
    const T* dh1 = (const T*)d.dh[1]; const T* h1 = (const T*)d.h[1]; const T* g1 = (const T*)d.g1[1];
    const T* x = (const T*)d.x; T* dx = (T*)d.dx;
    constexpr int Q = UP ? 4 : 1;
    float m[VEC], r[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        m[j] = d.mean[c + j]; r[j] = d.rstd[c + j];
        float s1 = d.sums[0][c + j], s2 = d.sums[0][C + c + j];
        if (NB == 2) { s1 += d.sums[1][c + j]; s2 += d.sums[1][C + c + j]; }
        k1[j] = (float)Q * r[j] * s1 * d.inv_count;
        k2[j] = (float)Q * r[j] * r[j] * s2 * d.inv_count;
    }
    const int W = d.W, H = d.H, W2 = W >> 1, H2 = H >> 1;
    const int64_t Pout = UP ? d.P / 4 : d.P;                       // pixels of x / dx
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t po = (int64_t)blockIdx.x * rows + tr; po < Pout; po += step) {
        int64_t pf[Q];
        if (UP) {
            const int pp = (int)po, n = pp / (H2 * W2), rem = pp - n * (H2 * W2);
            const int y2 = rem / W2, x2 = rem - y2 * W2;
            const int64_t base = ((int64_t)n * H + 2 * y2) * W + 2 * x2;
            pf[0] = base; if (Q > 1) { pf[1 % Q] = base + 1; pf[2 % Q] = base + W; pf[3 % Q] = base + W + 1; }
        } else pf[0] = po;
        float xv[VEC], acc[VEC];
        VT<T>::load(x + (size_t)po * C + c, xv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const size_t o = (size_t)pf[q] * C + c;
            float dv[VEC], hv[VEC], gv[VEC];
            VT<T>::load(dh0 + o, dv);
            if (h0) VT<T>::load(h0 + o, hv);
            if (g0) VT<T>::load(g0 + o, gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float t = h0 ? dv[j] * act_factor(hv[j], neg0) : dv[j];
                if (g0) t *= gv[j];
                acc[j] += t;
            }
            if (NB == 2) {
                VT<T>::load(dh1 + o, dv);
                if (h1) VT<T>::load(h1 + o, hv);
                if (g1) VT<T>::load(g1 + o, gv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float t = h1 ? dv[j] * act_factor(hv[j], neg1) : dv[j];
                    if (g1) t *= gv[j];
                    acc[j] += t;
                }
            }
        }
        float o4[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o4[j] = r[j] * acc[j] - k1[j] - k2[j] * (xv[j] - m[j]);
        VT<T>::store(dx + (size_t)po * C + c, o4);
    }
}

// Stage 2 of the forward statistics: a block owns 16 channels; lanes 0..15 of each thread row sum the channels' shifted partial sums,
// lanes 16..31 their shifted partial sums of squares (fp64, fixed order), then lanes 0..15 un-shift with the pivot x[g][0][c] (0 without `shift`) in fp64
// and write sums[g][2][C] (fp64; x sum_scale: the statistics of a nearest 2x upsample from its source, x 4, exact).
// FIN: the same lanes also run norm_finalize_kernel's arithmetic on those fp64 sums -- statistics that need no cross-rank reduction in
// between (instance norm; batch norm on one GPU) finish in two launches, bit-identical to mg_channel_stats + mg_norm_finalize.
template <typename T, bool FIN>
__global__ __launch_bounds__(256) void stats_stage2(const float* __restrict__ partial, const T* __restrict__ x, int64_t P, double* __restrict__ sums,
                                                    int nchunks, int C, int shift, double sum_scale, double count, float eps, float momentum,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ mean, float* __restrict__ rstd)
{
    __shared__ double red[256];
    __shared__ double fin[32];
    const int cl = threadIdx.x & 31, kk = threadIdx.x >> 5;
    const int c = blockIdx.x * 16 + (cl & 15);
    const int g = blockIdx.y;
    const int C2 = 2 * C;
    const int i = (cl >> 4) * C + c;                            // column of the [sum | sum of squares] vector
    double a = 0.0;
    if (c < C) {
        const float* p = partial + (size_t)g * nchunks * C2 + i;
        // eight independent loads in flight per thread (the trip count is a runtime value: without the explicit batch the loop was a
        // chain of ~64 dependent L2 round trips)
        int k = kk;
        for (; k + 56 < nchunks; k += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(k + 8 * j) * C2];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (double)v[j];
        }
        for (; k < nchunks; k += 8) a += (double)p[(size_t)k * C2];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (kk == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) a += red[r * 32 + cl];
        fin[cl] = a;
    }
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        const double kv = shift ? (double)ET<T>::load1(x + (size_t)g * P * C + c) : 0.0;
        const double sh = fin[threadIdx.x], ssh = fin[16 + threadIdx.x], n = (double)P;
        const double s = (sh + n * kv) * sum_scale;
        const double ss = (ssh + 2.0 * kv * sh + n * kv * kv) * sum_scale;
        sums[(size_t)g * C2 + c] = s;
        sums[(size_t)g * C2 + C + c] = ss;
        if (FIN) {
            const double m = s / count;
            double var = ss / count - m * m;
            if (var < 0.0) var = 0.0;
            mean[(size_t)g * C + c] = (float)m;
            rstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
}

// forward statistics: stage 1 (shifted fp32 partial sums) + stage 2 (fp64 sums [+ finalize])
template <typename T, bool FIN>
int run_stats(const void* x, int G, int64_t P, int C, int shift, double* sums, void* partial, double sum_scale, double count, float eps, float momentum,
              float* running_mean, float* running_var, float* mean, float* rstd, hipStream_t st)
{
    const StatGeom sg = stat_geom(G, P, C);
    dim3 grid(sg.nchunks, G);
    if (vec_geom_ok<T>(C))
        hipLaunchKernelGGL((stats_stage1_vec<T, 4>), grid, dim3(NTHR), 0, st, (const T*)x, (float*)partial, P, C, sg.chunk, shift);
    else
        hipLaunchKernelGGL((reduce_stage1<T, 0, true>), grid, dim3(NTHR), 0, st, (const T*)x, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (T*)nullptr, (float*)partial, P, C, sg.tpr, sg.rpb, sg.chunk, 0, 0.f, shift, 0, 0);
    MG_CHECK_LAUNCH("channel statistics (stage 1)");
    hipLaunchKernelGGL((stats_stage2<T, FIN>), dim3((C + 15) / 16, G), dim3(256), 0, st, (const float*)partial, (const T*)x, P, sums, sg.nchunks, C, shift,
                       sum_scale, count, eps, momentum, running_mean, running_var, mean, rstd);
    MG_CHECK_LAUNCH("channel statistics (stage 2)");
    return MG_OK;
}

// norm backward reduction (sum dxhat, sum dxhat * xhat [+ d[gamma|beta]]): these sums cancel by themselves (no pivot helps);
// fp32 per-thread partials over >= 16 values, fp64 across chunks.
template <typename T>
int run_reduce(const void* x, const void* dh, const void* h, const void* g1, const float* mean, const float* rstd,
               void* dgb, int G, int64_t P, int C, float* sums, void* partial, int act, float slope, hipStream_t st,
               int up = 0, int H = 0, int W = 0)
{
    const StatGeom sg = stat_geom(G, P, C);
    dim3 grid(sg.nchunks, G);
    if (g_mg_norm_bwd_vec && vec_geom_ok<T>(C) && act != MG_ACT_TANH) {
        // (d[gamma|beta] rows are 32-channel blocks; a thread's VEC channels start at a multiple of VEC and stay inside one block)
        const float neg = act == MG_ACT_NONE ? 1.f : (act == MG_ACT_RELU ? 0.f : slope);
        const bool hh = h != nullptr && act != MG_ACT_NONE;
#define MG_BWD1(HASH, UPF) hipLaunchKernelGGL((bwd_stage1_vec<T, 2, HASH, UPF>), grid, dim3(NTHR), 0, st, (const T*)x, (const T*)dh, (const T*)h, \
                           (const T*)g1, mean, rstd, (T*)dgb, (float*)partial, P, C, sg.chunk, neg, H, W)
        if (hh) { if (up) MG_BWD1(true, true); else MG_BWD1(true, false); }
        else    { if (up) MG_BWD1(false, true); else MG_BWD1(false, false); }
#undef MG_BWD1
    }
    else if (h == nullptr)
        hipLaunchKernelGGL((reduce_stage1<T, 1, false>), grid, dim3(NTHR), 0, st,
                           (const T*)x, (const T*)dh, (const T*)h, (const T*)g1, mean, rstd, (T*)dgb,
                           (float*)partial, P, C, sg.tpr, sg.rpb, sg.chunk, act, slope, up, H, W);
    else
        hipLaunchKernelGGL((reduce_stage1<T, 1, true>), grid, dim3(NTHR), 0, st,
                           (const T*)x, (const T*)dh, (const T*)h, (const T*)g1, mean, rstd, (T*)dgb,
                           (float*)partial, P, C, sg.tpr, sg.rpb, sg.chunk, act, slope, up, H, W);
    MG_CHECK_LAUNCH("reduce_stage1");
    dim3 grid2((2 * C + 31) / 32, G);
    hipLaunchKernelGGL(reduce_stage2, grid2, dim3(256), 0, st, (const float*)partial, sums, sg.nchunks, 2 * C);
    MG_CHECK_LAUNCH("reduce_stage2");
    return MG_OK;
}

}  // namespace

#define MG_CHECK_NORM_GEOM(name) \
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, name ": bad dtype %d", dtype); \
    MG_CHECK_ARG(G > 0 && P > 0 && C > 0 && (C % 4) == 0 && C <= 4096, name ": bad geometry G=%d P=%ld C=%d (C must be a multiple of 4, <= 4096)", G, (long)P, C)

extern "C" int64_t mg_stats_workspace(int32_t G, int64_t P, int32_t C)
{
    if (G <= 0 || P <= 0 || C <= 0) return 0;
    const StatGeom sg = stat_geom(G, P, C);
    return (int64_t)G * sg.nchunks * 2 * C * (int64_t)sizeof(float);
}

extern "C" int mg_channel_stats(const void* x, int32_t dtype, int32_t G, int64_t P, int32_t C, int32_t shift,
                                double* sums, void* partial, void* stream)
{
    MG_CHECK_NORM_GEOM("mg_channel_stats");
    MG_CHECK_ARG(x && sums && partial, "mg_channel_stats: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16)
        return run_stats<uint16_t, false>(x, G, P, C, shift != 0, sums, partial, 1.0, 1.0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, st);
    return run_stats<float, false>(x, G, P, C, shift != 0, sums, partial, 1.0, 1.0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, st);
}

extern "C" int mg_channel_stats_finalize(const void* x, int32_t dtype, int32_t G, int64_t P, int32_t C, float sum_scale, double count,
                                         float eps, float momentum, float* running_mean, float* running_var,
                                         double* sums, float* mean, float* rstd, void* partial, void* stream)
{
    MG_CHECK_NORM_GEOM("mg_channel_stats_finalize");
    MG_CHECK_ARG(x && sums && mean && rstd && partial, "mg_channel_stats_finalize: null pointer");
    MG_CHECK_ARG(count > 0 && sum_scale > 0.f, "mg_channel_stats_finalize: bad count / scale");
    MG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && (running_mean == nullptr || G == 1),
                 "mg_channel_stats_finalize: running statistics need both buffers and G == 1");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16)
        return run_stats<uint16_t, true>(x, G, P, C, 1, sums, partial, (double)sum_scale, count, eps, momentum, running_mean, running_var, mean, rstd, st);
    return run_stats<float, true>(x, G, P, C, 1, sums, partial, (double)sum_scale, count, eps, momentum, running_mean, running_var, mean, rstd, st);
}

extern "C" int mg_norm_act_fwd(const void* x, void* y, int32_t dtype, int32_t G, int64_t P, int32_t C,
                               const float* mean, const float* rstd, int32_t act, float slope, const void* resid, void* stream)
{
    MG_CHECK_NORM_GEOM("mg_norm_act_fwd");
    MG_CHECK_ARG(x && y && mean && rstd, "mg_norm_act_fwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = (int64_t)G * P * (C / 4);
    if (act != MG_ACT_TANH && (dtype == MG_BF16 ? vec_geom_ok<uint16_t>(C) : vec_geom_ok<float>(C))) {
        const float neg = act == MG_ACT_NONE ? 1.f : (act == MG_ACT_RELU ? 0.f : slope);
        const bool relu = act == MG_ACT_RELU;
        if (dtype == MG_BF16) {
            const int rows = NTHR / (C / 8);
            hipLaunchKernelGGL((norm_act_fwd_vec<uint16_t, 4>), dim3(pix_grid(P, rows, 4, G), G), dim3(NTHR), 0, st,
                               (const uint16_t*)x, (uint16_t*)y, P, C, mean, rstd, neg, relu, (const uint16_t*)resid);
        } else {
            const int rows = NTHR / (C / 4);
            hipLaunchKernelGGL((norm_act_fwd_vec<float, 4>), dim3(pix_grid(P, rows, 4, G), G), dim3(NTHR), 0, st,
                               (const float*)x, (float*)y, P, C, mean, rstd, neg, relu, (const float*)resid);
        }
        MG_CHECK_LAUNCH("mg_norm_act_fwd");
        return MG_OK;
    }
    if (dtype == MG_BF16)
        hipLaunchKernelGGL(norm_act_fwd_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st,
                           (const uint16_t*)x, (uint16_t*)y, nq, P, C, mean, rstd, act, slope, (const uint16_t*)resid);
    else
        hipLaunchKernelGGL(norm_act_fwd_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st,
                           (const float*)x, (float*)y, nq, P, C, mean, rstd, act, slope, (const float*)resid);
    MG_CHECK_LAUNCH("mg_norm_act_fwd");
    return MG_OK;
}

extern "C" int mg_norm_bwd_reduce(const void* dh, const void* h, const void* x, const void* g1,
                                  int32_t dtype, int32_t G, int64_t P, int32_t C,
                                  const float* mean, const float* rstd, int32_t act, float slope,
                                  void* dgb, float* sums, void* partial, void* stream)
{
    MG_CHECK_NORM_GEOM("mg_norm_bwd_reduce");
    MG_CHECK_ARG(dh && (h || act == MG_ACT_NONE) && x && mean && rstd && sums && partial, "mg_norm_bwd_reduce: null pointer");
    MG_CHECK_ARG(dgb == nullptr || G == 1, "mg_norm_bwd_reduce: dgb output requires G == 1");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16)
        return run_reduce<uint16_t>(x, dh, h, g1, mean, rstd, dgb, G, P, C, sums, partial, act, slope, st);
    return run_reduce<float>(x, dh, h, g1, mean, rstd, dgb, G, P, C, sums, partial, act, slope, st);
}

extern "C" int mg_norm_bwd_reduce_up(const void* dh, const void* h, const void* x, const void* g1,
                                     int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                                     const float* mean, const float* rstd, int32_t act, float slope,
                                     void* dgb, float* sums, void* partial, void* stream)
{
    const int32_t G = 1; const int64_t P = (int64_t)N * H * W;
    MG_CHECK_NORM_GEOM("mg_norm_bwd_reduce_up");
    MG_CHECK_ARG(dh && (h || act == MG_ACT_NONE) && x && mean && rstd && sums && partial, "mg_norm_bwd_reduce_up: null pointer");
    MG_CHECK_ARG(N > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && P < (1L << 31), "mg_norm_bwd_reduce_up: H, W must be even");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16)
        return run_reduce<uint16_t>(x, dh, h, g1, mean, rstd, dgb, G, P, C, sums, partial, act, slope, st, 1, H, W);
    return run_reduce<float>(x, dh, h, g1, mean, rstd, dgb, G, P, C, sums, partial, act, slope, st, 1, H, W);
}

template <typename T>
static int launch_apply2(const mg_norm_apply2_desc& d, hipStream_t st)
{
    constexpr int VEC = VT<T>::VEC;
    const int rows = NTHR / (d.C / VEC);
    const bool two = d.dh[1] != nullptr;
    const int64_t pout = d.up ? d.P / 4 : d.P;
    auto neg = [](int act, float slope) { return act == MG_ACT_NONE ? 1.f : (act == MG_ACT_RELU ? 0.f : slope); };
    const float n0 = neg(d.act[0], d.slope[0]), n1 = neg(d.act[1], d.slope[1]);
    const dim3 grid(pix_grid(pout, rows, 1, 1)), blk(NTHR);
    if (d.up) {
        if (two) hipLaunchKernelGGL((norm_bwd_apply2_vec<T, 2, true>), grid, blk, 0, st, d, n0, n1);
        else     hipLaunchKernelGGL((norm_bwd_apply2_vec<T, 1, true>), grid, blk, 0, st, d, n0, n1);
    } else {
        if (two) hipLaunchKernelGGL((norm_bwd_apply2_vec<T, 2, false>), grid, blk, 0, st, d, n0, n1);
        else     hipLaunchKernelGGL((norm_bwd_apply2_vec<T, 1, false>), grid, blk, 0, st, d, n0, n1);
    }
    return 0;
}

extern "C" int mg_norm_apply2_supported(int32_t dtype, int32_t C)
{
    return dtype == MG_BF16 ? vec_geom_ok<uint16_t>(C) : (dtype == MG_F32 ? vec_geom_ok<float>(C) : 0);
}

extern "C" int mg_norm_bwd_apply2(const mg_norm_apply2_desc* d, void* stream)
{
    MG_CHECK_ARG(d != nullptr, "mg_norm_bwd_apply2: null descriptor");
    MG_CHECK_ARG(d->dtype == MG_F32 || d->dtype == MG_BF16, "mg_norm_bwd_apply2: bad dtype");
    MG_CHECK_ARG(d->dh[0] && d->sums[0] && d->x && d->mean && d->rstd && d->dx, "mg_norm_bwd_apply2: null pointer");
    MG_CHECK_ARG((d->dh[1] == nullptr) == (d->sums[1] == nullptr), "mg_norm_bwd_apply2: the second branch needs dh and sums");
    for (int b = 0; b < 2; ++b)
        MG_CHECK_ARG(d->act[b] != MG_ACT_TANH && (d->h[b] || d->act[b] == MG_ACT_NONE || !d->dh[b]), "mg_norm_bwd_apply2: activation needs its output h");
    MG_CHECK_ARG(d->C > 0 && d->P > 0 && d->P < (1L << 31) && mg_norm_apply2_supported(d->dtype, d->C), "mg_norm_bwd_apply2: unsupported geometry C=%d", d->C);
    MG_CHECK_ARG(!d->up || (d->H > 0 && d->W > 0 && (d->H % 2) == 0 && (d->W % 2) == 0 && d->P % ((int64_t)d->H * d->W) == 0),
                 "mg_norm_bwd_apply2: up needs even H, W and P = N*H*W");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MG_BF16) launch_apply2<uint16_t>(*d, st); else launch_apply2<float>(*d, st);
    MG_CHECK_LAUNCH("mg_norm_bwd_apply2");
    return MG_OK;
}

extern "C" int mg_norm_bwd_apply(const void* dh, const void* h, const void* x, const void* g1,
                                 int32_t dtype, int32_t G, int64_t P, int32_t C,
                                 const float* mean, const float* rstd, const float* s1, const float* s2,
                                 int32_t sum_gstride, float sum_scale, int32_t act, float slope, void* dx, void* stream)
{
    MG_CHECK_NORM_GEOM("mg_norm_bwd_apply");
    MG_CHECK_ARG(sum_gstride >= C, "mg_norm_bwd_apply: sum_gstride < C");
    const int sgs = sum_gstride; const float sscale = sum_scale;
    MG_CHECK_ARG(dh && (h || act == MG_ACT_NONE) && x && mean && rstd && s1 && s2 && dx, "mg_norm_bwd_apply: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = (int64_t)G * P * (C / 4);
    if (act != MG_ACT_TANH && (dtype == MG_BF16 ? vec_geom_ok<uint16_t>(C) : vec_geom_ok<float>(C))) {
        const float neg = act == MG_ACT_NONE ? 1.f : (act == MG_ACT_RELU ? 0.f : slope);
        if (dtype == MG_BF16) {
            const int rows = NTHR / (C / 8);
            hipLaunchKernelGGL((norm_bwd_apply_vec<uint16_t, 2>), dim3(pix_grid(P, rows, 2, G), G), dim3(NTHR), 0, st,
                               (const uint16_t*)dh, (const uint16_t*)h, (const uint16_t*)x, (const uint16_t*)g1, (uint16_t*)dx,
                               P, C, mean, rstd, s1, s2, sgs, sscale, neg);
        } else {
            const int rows = NTHR / (C / 4);
            hipLaunchKernelGGL((norm_bwd_apply_vec<float, 2>), dim3(pix_grid(P, rows, 2, G), G), dim3(NTHR), 0, st,
                               (const float*)dh, (const float*)h, (const float*)x, (const float*)g1, (float*)dx,
                               P, C, mean, rstd, s1, s2, sgs, sscale, neg);
        }
        MG_CHECK_LAUNCH("mg_norm_bwd_apply");
        return MG_OK;
    }
    if (dtype == MG_BF16)
        hipLaunchKernelGGL(norm_bwd_apply_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st,
                           (const uint16_t*)dh, (const uint16_t*)h, (const uint16_t*)x, (const uint16_t*)g1, (uint16_t*)dx,
                           nq, P, C, mean, rstd, s1, s2, sgs, sscale, act, slope);
    else
        hipLaunchKernelGGL(norm_bwd_apply_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st,
                           (const float*)dh, (const float*)h, (const float*)x, (const float*)g1, (float*)dx,
                           nq, P, C, mean, rstd, s1, s2, sgs, sscale, act, slope);
    MG_CHECK_LAUNCH("mg_norm_bwd_apply");
    return MG_OK;
}

extern "C" int mg_norm_finalize(const double* sums, int32_t G, int32_t C, double count, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* rstd, void* stream)
{
    MG_CHECK_ARG(sums && mean && rstd, "mg_norm_finalize: null pointer");
    MG_CHECK_ARG(G > 0 && C > 0 && count > 0, "mg_norm_finalize: bad geometry");
    MG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && (running_mean == nullptr || G == 1),
                 "mg_norm_finalize: running statistics need both buffers and G == 1");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((G * C + 255) / 256), dim3(256), 0, st, sums, G, C, count, eps, momentum,
                       running_mean, running_var, mean, rstd);
    MG_CHECK_LAUNCH("mg_norm_finalize");
    return MG_OK;
}
