// mg_pointwise.hip -- HBM-bound NHWC streaming kernels around the conv engine:
// activation backward, nearest 2x upsample, 3x3/s2 average pool (discriminator
// pyramid), 2x2 max pool (VGG tower), background blend, fused flat Adam, and the
// hardware fragment-layout probes used by the test-suite.
// One thread = one quad (4 consecutive channels) -> 8 B (bf16) / 16 B (f32) accesses.
#include "mg_common.h"

namespace {

constexpr int NTHR = 256;
static inline int ew_grid(int64_t n) { int64_t b = (n + NTHR - 1) / NTHR; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dpre,
                               int64_t nquads, int act, float slope)
{
    GRID_STRIDE(i, nquads) {
        const f32x4_t d = ET<T>::load4(dy + i * 4), v = ET<T>::load4(y + i * 4);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = d[j] * mg_act_grad_from_out(v[j], act, slope);
        ET<T>::store4(dpre + i * 4, o);
    }
}

// y[n, 2h+a, 2w+b, :] = x[n, h, w, :]   (one thread per OUTPUT quad)
template <typename T>
__global__ void up2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C)
{
    const int c4 = C / 4; const int Ho = 2 * H, Wo = 2 * W;
    const int64_t n = (int64_t)N * Ho * Wo * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
        const size_t src = (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + qd * 4;
        ET<T>::store4(y + i * 4, ET<T>::load4(x + src));
    }
}
// nn.ReflectionPad2d(p): y[n, oy, ox, :] = x[n, refl(oy - p), refl(ox - p), :]   (one thread per OUTPUT quad)
__device__ __forceinline__ int mg_reflect(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * n - 2 - i : i; }
template <typename T>
__global__ void reflect_pad_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int P)
{
    const int c4 = C / 4; const int Ho = H + 2 * P, Wo = W + 2 * P;
    const int64_t n = (int64_t)N * Ho * Wo * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
        const size_t src = (((size_t)b * H + mg_reflect(oy - P, H)) * W + mg_reflect(ox - P, W)) * C + qd * 4;
        ET<T>::store4(y + i * 4, ET<T>::load4(x + src));
    }
}
// adjoint: dx[n, iy, ix, :] = sum of dy over the padded positions that read (iy, ix) -- itself plus, within P of a
// border (excluding the border row/column itself), its mirror image on that side; up to 3 x 3 terms.
template <typename T>
__global__ void reflect_pad_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int P)
{
    const int c4 = C / 4; const int Ho = H + 2 * P, Wo = W + 2 * P;
    const int64_t n = (int64_t)N * H * W * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H); const int b = (int)(p / H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = iy + P;
        if (iy >= 1 && iy <= P) ys[ny++] = P - iy;                         // mirrored above the top edge
        if (iy <= H - 2 && iy >= H - 1 - P) ys[ny++] = P + 2 * (H - 1) - iy;   // mirrored below the bottom edge
        xs[nx++] = ix + P;
        if (ix >= 1 && ix <= P) xs[nx++] = P - ix;
        if (ix <= W - 2 && ix >= W - 1 - P) xs[nx++] = P + 2 * (W - 1) - ix;
        f32x4_t o = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e) {
                const f32x4_t v = ET<T>::load4(dy + (((size_t)b * Ho + ys[a]) * Wo + xs[e]) * C + qd * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += v[j];
            }
        ET<T>::store4(dx + i * 4, o);
    }
}
// dx[n,h,w,:] = sum of the 2x2 children of dy
template <typename T>
__global__ void up2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C)
{
    const int c4 = C / 4; const int Wo = 2 * W;
    const int64_t n = (int64_t)N * H * W * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H); const int b = (int)(p / H);
        const size_t base = (((size_t)b * 2 * H + 2 * iy) * Wo + 2 * ix) * C + qd * 4;
        const f32x4_t a = ET<T>::load4(dy + base), bq = ET<T>::load4(dy + base + C);
        const f32x4_t c = ET<T>::load4(dy + base + (size_t)Wo * C), e = ET<T>::load4(dy + base + (size_t)Wo * C + C);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (a[j] + bq[j]) + (c[j] + e[j]);
        ET<T>::store4(dx + i * 4, o);
    }
}

__device__ __forceinline__ int pool_cnt(int o, int L) {   // valid taps of a k3 s2 p1 window along one axis
    const int lo = 2 * o - 1, hi = 2 * o + 1;
    return (hi < L ? hi : L - 1) - (lo > 0 ? lo : 0) + 1;
}
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo)
{
    const int c4 = C / 4;
    const int64_t n = (int64_t)N * Ho * Wo * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        for (int ky = -1; ky <= 1; ++ky) {
            const int iy = 2 * oy + ky; if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = -1; kx <= 1; ++kx) {
                const int ix = 2 * ox + kx; if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4_t v = ET<T>::load4(x + (((size_t)b * H + iy) * W + ix) * C + qd * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[j];
            }
        }
        const float inv = 1.f / (float)(pool_cnt(oy, H) * pool_cnt(ox, W));
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] *= inv;
        ET<T>::store4(y + i * 4, s);
    }
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo)
{
    const int c4 = C / 4;
    const int64_t n = (int64_t)N * H * W * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H); const int b = (int)(p / H);
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        // windows oy with |2*oy - iy| <= 1
        for (int oy = (iy) >> 1; oy <= (iy + 1) >> 1; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = (ix) >> 1; ox <= (ix + 1) >> 1; ++ox) {
                if (ox >= Wo) continue;
                const float inv = 1.f / (float)(pool_cnt(oy, H) * pool_cnt(ox, W));
                const f32x4_t v = ET<T>::load4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + qd * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[j] * inv;
            }
        }
        ET<T>::store4(dx + i * 4, s);
    }
}

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo)
{
    const int c4 = C / 4;
    const int64_t n = (int64_t)N * Ho * Wo * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); const int b = (int)(p / Ho);
        const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + qd * 4;
        const f32x4_t a = ET<T>::load4(x + base), bq = ET<T>::load4(x + base + C);
        const f32x4_t c = ET<T>::load4(x + base + (size_t)W * C), e = ET<T>::load4(x + base + (size_t)W * C + C);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(fmaxf(a[j], bq[j]), fmaxf(c[j], e[j]));
        ET<T>::store4(y + i * 4, o);
    }
}
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                   int N, int H, int W, int C, int Ho, int Wo, int relu_in)
{
    const int c4 = C / 4;
    const int64_t n = (int64_t)N * H * W * c4;
    GRID_STRIDE(i, n) {
        const int qd = (int)(i % c4); int64_t p = i / c4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H); const int b = (int)(p / H);
        const int oy = iy >> 1, ox = ix >> 1;
        f32x4_t o = {0.f, 0.f, 0.f, 0.f};
        if (oy < Ho && ox < Wo) {
            const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + qd * 4;
            f32x4_t v[4];
            v[0] = ET<T>::load4(x + base); v[1] = ET<T>::load4(x + base + C);
            v[2] = ET<T>::load4(x + base + (size_t)W * C); v[3] = ET<T>::load4(x + base + (size_t)W * C + C);
            const f32x4_t g = ET<T>::load4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + qd * 4);
            const int me = (iy & 1) * 2 + (ix & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int arg = 0; float m = v[0][j];
#pragma unroll
                for (int k = 1; k < 4; ++k) if (v[k][j] > m) { m = v[k][j]; arg = k; }   // first maximum wins (ATen order)
                o[j] = (arg == me && (!relu_in || m > 0.f)) ? g[j] : 0.f;     // relu_in: x is a ReLU's output -> its backward mask rides along
            }
        }
        ET<T>::store4(dx + i * 4, o);
    }
}

template <typename T>
__global__ void blend_fwd_kernel(const T* __restrict__ bg, const T* __restrict__ x, const float* __restrict__ hair,
                                 const float* __restrict__ back, T* __restrict__ y, int64_t nquads, int C, int act, float slope)
{
    const int c4 = C / 4;
    GRID_STRIDE(i, nquads) {
        const int64_t p = i / c4;
        const float wb = 1.f - hair[p], wx = 1.f - back[p];
        const f32x4_t b = ET<T>::load4(bg + i * 4), v = ET<T>::load4(x + i * 4);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = mg_act(b[j] * wb + v[j] * wx, act, slope);
        ET<T>::store4(y + i * 4, o);
    }
}
template <typename T>
__global__ void blend_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ hair,
                                 const float* __restrict__ back, T* __restrict__ dbg, T* __restrict__ dx, int64_t nquads, int C,
                                 int act, float slope)
{
    const int c4 = C / 4;
    GRID_STRIDE(i, nquads) {
        const int64_t p = i / c4;
        const float wb = 1.f - hair[p], wx = 1.f - back[p];
        f32x4_t d = ET<T>::load4(dy + i * 4);
        if (act != MG_ACT_NONE) {
            const f32x4_t yv = ET<T>::load4(y + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] *= mg_act_grad_from_out(yv[j], act, slope);
        }
        f32x4_t a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = d[j] * wb; b[j] = d[j] * wx; }
        if (dbg) ET<T>::store4(dbg + i * 4, a);
        if (dx)  ET<T>::store4(dx + i * 4, b);
    }
}

// torch.optim.Adam (single-tensor formulation): see pix2pix_model.py:137-145
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale)
{
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        // exp_avg.lerp_(grad, 1 - beta1) with torch's two-branch lerp (ATen Lerp.h): weight >= 0.5 evaluates end - (end - start) * (1 - weight).
        // beta1 = 0 (the reference's TTUR setting, pix2pix_model.py:137-145) must give m = g EXACTLY: the one-branch form m + (g - m) rounds
        // g away whenever |g| << |m| (m = last step's gradient), and g / (|g| + eps) -- a sign function here -- then steps on the rounding.
        const float w = 1.f - b1, mo = b1 == 0.f ? 0.f : m[i];
        const float mi = w < 0.5f ? mo + w * (gi - mo) : gi - (gi - mo) * (1.f - w);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}


// ---- fused L1 loss (feature matching / VGG perceptual taps, loss.py:163-175,199-207) -------------
// forward: mean |a - b| with ONE pass over both tensors (block partials -> fixed-order fp64 finish);
// backward: da = sign(a - b) * (*gscale) / numel in the activation dtype, ONE pass.  The eager form
// (float casts, sub, abs, mean and their autograd) is ~10 passes over each feature map.
template <typename T>
__global__ __launch_bounds__(NTHR) void l1_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          float* __restrict__ partial, int64_t nquads)
{
    __shared__ float red[NTHR / 64];
    float s = 0.f;
    GRID_STRIDE(i, nquads) {
        const f32x4_t x = ET<T>::load4(a + i * 4), y = ET<T>::load4(b + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += fabsf(x[j] - y[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < NTHR / 64; ++w) t += red[w]; partial[blockIdx.x] = t; }
}
__global__ void l1_final_kernel(const float* __restrict__ partial, int n, double inv_numel, float* __restrict__ out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_numel);
}
template <typename T>
__global__ void l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ gscale,
                              float inv_numel, T* __restrict__ da, int64_t nquads)
{
    const float g = gscale[0] * inv_numel;
    GRID_STRIDE(i, nquads) {
        const f32x4_t x = ET<T>::load4(a + i * 4), y = ET<T>::load4(b + i * 4);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float dlt = x[j] - y[j]; o[j] = dlt > 0.f ? g : (dlt < 0.f ? -g : 0.f); }
        ET<T>::store4(da + i * 4, o);
    }
}

// ---- probes -----------------------------------------------------------------
__global__ void probe_mfma_kernel(float* out)
{
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    auto bfbits = [](float f) { return (short)(__float_as_uint(f) >> 16); };   // exact for small ints
    for (int e = 0; e < 2; ++e) {
        s16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hi == 0) {
            a[0] = bfbits(e == 0 ? (float)l31 : 1.f);      // A[i][k0] : row index i = lane&31
            b[0] = bfbits(e == 0 ? 1.f : (float)l31);      // B[k0][n] : col index n = lane&31
        }
        f32x16_t c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[(e * 64 + lane) * 16 + r] = c[r];
    }
    {
        const float a = hi == 0 ? (float)l31 : 0.f, b = hi == 0 ? 1.f : 0.f;
        f32x16_t c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[(2 * 64 + lane) * 16 + r] = c[r];
    }
}
__global__ void probe_tr16_kernel(const uint16_t* in, uint16_t* out)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    const int lane = threadIdx.x;
    for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = in[lane * 4 + j];
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4_t* lp_t;
    s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(lds + lane * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}

}  // namespace

#define MG_EW_GEOM(name) \
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, name ": bad dtype %d", dtype); \
    MG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, name ": bad geometry N=%d H=%d W=%d C=%d (C %% 4 == 0)", N, H, W, C)
#define MG_LAUNCH2(kern, grid, ...) do { \
    if (dtype == MG_BF16) hipLaunchKernelGGL(kern<uint16_t>, dim3(grid), dim3(NTHR), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern<float>, dim3(grid), dim3(NTHR), 0, st, __VA_ARGS__); } while (0)

#define CT(p) reinterpret_cast<const T_*>(p)

extern "C" int mg_act_bwd(const void* dy, const void* y, void* dpre, int32_t dtype, int64_t numel,
                          int32_t act, float slope, void* stream)
{
    MG_CHECK_ARG(dy && y && dpre, "mg_act_bwd: null pointer");
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, "mg_act_bwd: bad dtype");
    MG_CHECK_ARG(numel > 0 && (numel % 4) == 0, "mg_act_bwd: numel=%ld must be a positive multiple of 4", (long)numel);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = numel / 4;
    if (dtype == MG_BF16) hipLaunchKernelGGL(act_bwd_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dpre, nq, act, slope);
    else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const float*)dy, (const float*)y, (float*)dpre, nq, act, slope);
    MG_CHECK_LAUNCH("mg_act_bwd");
    return MG_OK;
}

extern "C" int mg_upsample2x_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_EW_GEOM("mg_upsample2x_fwd"); MG_CHECK_ARG(x && y, "mg_upsample2x_fwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((int64_t)N * 4 * H * W * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(up2_fwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)x, (uint16_t*)y, N, H, W, C);
    else hipLaunchKernelGGL(up2_fwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)x, (float*)y, N, H, W, C);
    MG_CHECK_LAUNCH("mg_upsample2x_fwd");
    return MG_OK;
}
extern "C" int mg_upsample2x_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_EW_GEOM("mg_upsample2x_bwd"); MG_CHECK_ARG(dy && dx, "mg_upsample2x_bwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((int64_t)N * H * W * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(up2_bwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)dy, (uint16_t*)dx, N, H, W, C);
    else hipLaunchKernelGGL(up2_bwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)dy, (float*)dx, N, H, W, C);
    MG_CHECK_LAUNCH("mg_upsample2x_bwd");
    return MG_OK;
}

extern "C" int mg_reflect_pad_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, void* stream)
{
    MG_EW_GEOM("mg_reflect_pad_fwd"); MG_CHECK_ARG(x && y, "mg_reflect_pad_fwd: null pointer");
    MG_CHECK_ARG(P >= 1 && P < H && P < W, "mg_reflect_pad_fwd: pad %d must be in [1, min(H, W))", P);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((int64_t)N * (H + 2 * P) * (W + 2 * P) * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(reflect_pad_fwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)x, (uint16_t*)y, N, H, W, C, P);
    else hipLaunchKernelGGL(reflect_pad_fwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)x, (float*)y, N, H, W, C, P);
    MG_CHECK_LAUNCH("mg_reflect_pad_fwd");
    return MG_OK;
}
extern "C" int mg_reflect_pad_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, void* stream)
{
    MG_EW_GEOM("mg_reflect_pad_bwd"); MG_CHECK_ARG(dy && dx, "mg_reflect_pad_bwd: null pointer");
    MG_CHECK_ARG(P >= 1 && P < H && P < W, "mg_reflect_pad_bwd: pad %d must be in [1, min(H, W))", P);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((int64_t)N * H * W * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(reflect_pad_bwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)dy, (uint16_t*)dx, N, H, W, C, P);
    else hipLaunchKernelGGL(reflect_pad_bwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)dy, (float*)dx, N, H, W, C, P);
    MG_CHECK_LAUNCH("mg_reflect_pad_bwd");
    return MG_OK;
}

extern "C" int mg_avgpool3s2_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_EW_GEOM("mg_avgpool3s2_fwd"); MG_CHECK_ARG(x && y, "mg_avgpool3s2_fwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int g = ew_grid((int64_t)N * Ho * Wo * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(avgpool_fwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)x, (uint16_t*)y, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
    MG_CHECK_LAUNCH("mg_avgpool3s2_fwd");
    return MG_OK;
}
extern "C" int mg_avgpool3s2_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_EW_GEOM("mg_avgpool3s2_bwd"); MG_CHECK_ARG(dy && dx, "mg_avgpool3s2_bwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int g = ew_grid((int64_t)N * H * W * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(avgpool_bwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)dy, (uint16_t*)dx, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo);
    MG_CHECK_LAUNCH("mg_avgpool3s2_bwd");
    return MG_OK;
}

extern "C" int mg_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_EW_GEOM("mg_maxpool2_fwd"); MG_CHECK_ARG(x && y, "mg_maxpool2_fwd: null pointer");
    MG_CHECK_ARG(H >= 2 && W >= 2, "mg_maxpool2_fwd: H, W must be >= 2");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Ho = H / 2, Wo = W / 2;
    const int g = ew_grid((int64_t)N * Ho * Wo * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)x, (uint16_t*)y, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
    MG_CHECK_LAUNCH("mg_maxpool2_fwd");
    return MG_OK;
}
// out[p][0..8) = [planar[n][0..cp)[p] | nhwc[p][0..cf) | 0 ...]: the patch discriminators' input pixel (7 channels: tag one-hot,
// orientation, image; pix2pix_model.py:559-566) assembled in the kernels' NHWC layout with its zero pad channel in ONE pass from
// the reference's planar fp32 maps and the generator's NHWC image (eager: two concatenations, a permute + cast + copy, a pad).
template <typename T>
__global__ void assemble_nhwc8_kernel(const float* __restrict__ planar, int cp, const T* __restrict__ nhwc, int cs, int cf,
                                      T* __restrict__ out, int64_t npix, int64_t hw)
{
    GRID_STRIDE(i, npix) {
        const int64_t n = i / hw, p = i - n * hw;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = 0.f;
            if (c < cp) t = planar[((size_t)n * cp + c) * hw + p];
            else if (c - cp < cf) t = ET<T>::load1(nhwc + (size_t)i * cs + (c - cp));
            v[c] = t;
        }
        f32x4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        ET<T>::store4(out + (size_t)i * 8, a);
        ET<T>::store4(out + (size_t)i * 8 + 4, b);
    }
}

extern "C" int mg_assemble_nhwc8(const float* planar, int32_t cp, const void* nhwc, int32_t cs, int32_t cf, void* out, int32_t dtype,
                                 int32_t N, int64_t HW, void* stream)
{
    MG_CHECK_ARG(out && (planar || cp == 0) && (nhwc || cf == 0), "mg_assemble_nhwc8: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && N > 0 && HW > 0 && cp >= 0 && cf >= 0 && cp + cf <= 8 && cf <= cs,
                 "mg_assemble_nhwc8: bad geometry (cp + cf <= 8, cf <= cs)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)N * HW;
    if (dtype == MG_BF16) hipLaunchKernelGGL(assemble_nhwc8_kernel<uint16_t>, dim3(ew_grid(npix)), dim3(NTHR), 0, st, planar, cp, (const uint16_t*)nhwc, cs, cf, (uint16_t*)out, npix, HW);
    else hipLaunchKernelGGL(assemble_nhwc8_kernel<float>, dim3(ew_grid(npix)), dim3(NTHR), 0, st, planar, cp, (const float*)nhwc, cs, cf, (float*)out, npix, HW);
    MG_CHECK_LAUNCH("mg_assemble_nhwc8");
    return MG_OK;
}

// out = (g1 + g2) * act'(y): the gradient of an activation's output that has TWO consumers (a ReLU tap of the VGG tower feeding
// the next conv and the perceptual loss; a background-encoder feature feeding the next layer and the blend), summed and pushed
// through the activation in one pass instead of autograd's add (2r + 1w) followed by act_bwd (2r + 1w).  g2 may be NULL.
template <typename T>
__global__ void grad_sum_act_kernel(const T* __restrict__ g1, const T* __restrict__ g2, const T* __restrict__ y, T* __restrict__ out,
                                    int64_t nquads, int act, float slope)
{
    GRID_STRIDE(i, nquads) {
        f32x4_t a = ET<T>::load4(g1 + i * 4);
        if (g2) { const f32x4_t b = ET<T>::load4(g2 + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += b[j]; }
        const f32x4_t v = ET<T>::load4(y + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] *= mg_act_grad_from_out(v[j], act, slope);
        ET<T>::store4(out + i * 4, a);
    }
}

extern "C" int mg_grad_sum_act(const void* g1, const void* g2, const void* y, void* out, int32_t dtype, int64_t numel, int32_t act,
                               float slope, void* stream)
{
    MG_CHECK_ARG(g1 && y && out, "mg_grad_sum_act: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && numel > 0 && (numel % 4) == 0, "mg_grad_sum_act: numel must be a positive multiple of 4");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = numel / 4;
    if (dtype == MG_BF16) hipLaunchKernelGGL(grad_sum_act_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const uint16_t*)g1, (const uint16_t*)g2, (const uint16_t*)y, (uint16_t*)out, nq, act, slope);
    else hipLaunchKernelGGL(grad_sum_act_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const float*)g1, (const float*)g2, (const float*)y, (float*)out, nq, act, slope);
    MG_CHECK_LAUNCH("mg_grad_sum_act");
    return MG_OK;
}

extern "C" int mg_maxpool2_bwd(const void* dy, const void* x, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t relu_input, void* stream)
{
    MG_EW_GEOM("mg_maxpool2_bwd"); MG_CHECK_ARG(dy && x && dx, "mg_maxpool2_bwd: null pointer");
    MG_CHECK_ARG(H >= 2 && W >= 2, "mg_maxpool2_bwd: H, W must be >= 2");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Ho = H / 2, Wo = W / 2;
    const int g = ew_grid((int64_t)N * H * W * (C / 4));
    if (dtype == MG_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<uint16_t>, dim3(g), dim3(NTHR), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (uint16_t*)dx, N, H, W, C, Ho, Wo, relu_input);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(g), dim3(NTHR), 0, st, (const float*)dy, (const float*)x, (float*)dx, N, H, W, C, Ho, Wo, relu_input);
    MG_CHECK_LAUNCH("mg_maxpool2_bwd");
    return MG_OK;
}

extern "C" int mg_blend_fwd(const void* bg, const void* x, const float* hair, const float* back, void* y,
                            int32_t dtype, int64_t P, int32_t C, int32_t act, float slope, void* stream)
{
    MG_CHECK_ARG(bg && x && hair && back && y, "mg_blend_fwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && P > 0 && C > 0 && (C % 4) == 0, "mg_blend_fwd: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = P * (C / 4);
    if (dtype == MG_BF16) hipLaunchKernelGGL(blend_fwd_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const uint16_t*)bg, (const uint16_t*)x, hair, back, (uint16_t*)y, nq, C, act, slope);
    else hipLaunchKernelGGL(blend_fwd_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const float*)bg, (const float*)x, hair, back, (float*)y, nq, C, act, slope);
    MG_CHECK_LAUNCH("mg_blend_fwd");
    return MG_OK;
}
extern "C" int mg_blend_bwd(const void* dy, const void* y, const float* hair, const float* back, void* dbg, void* dx,
                            int32_t dtype, int64_t P, int32_t C, int32_t act, float slope, void* stream)
{
    MG_CHECK_ARG(dy && hair && back && (dbg || dx) && (y || act == MG_ACT_NONE), "mg_blend_bwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && P > 0 && C > 0 && (C % 4) == 0, "mg_blend_bwd: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = P * (C / 4);
    if (dtype == MG_BF16) hipLaunchKernelGGL(blend_bwd_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const uint16_t*)dy, (const uint16_t*)y, hair, back, (uint16_t*)dbg, (uint16_t*)dx, nq, C, act, slope);
    else hipLaunchKernelGGL(blend_bwd_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const float*)dy, (const float*)y, hair, back, (float*)dbg, (float*)dx, nq, C, act, slope);
    MG_CHECK_LAUNCH("mg_blend_bwd");
    return MG_OK;
}

extern "C" int mg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            int64_t numel, float lr, float beta1, float beta2, float eps, int32_t step,
                            float grad_scale, void* stream)
{
    MG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "mg_adam_step: null pointer");
    MG_CHECK_ARG(numel > 0 && step >= 1, "mg_adam_step: bad numel/step");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(numel)), dim3(NTHR), 0, st, param, grad, exp_avg, exp_avg_sq, numel,
                       lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
    MG_CHECK_LAUNCH("mg_adam_step");
    return MG_OK;
}

extern "C" int mg_probe_mfma_layout(float* out, void* stream)
{
    MG_CHECK_ARG(out, "mg_probe_mfma_layout: null pointer");
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out);
    MG_CHECK_LAUNCH("mg_probe_mfma_layout");
    return MG_OK;
}
extern "C" int mg_probe_tr16(const uint16_t* in, uint16_t* out, void* stream)
{
    MG_CHECK_ARG(in && out, "mg_probe_tr16: null pointer");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), in, out);
    MG_CHECK_LAUNCH("mg_probe_tr16");
    return MG_OK;
}

extern "C" int mg_l1_mean_fwd(const void* a, const void* b, int32_t dtype, int64_t numel, float* out, float* partial, void* stream)
{
    MG_CHECK_ARG(a && b && out && partial, "mg_l1_mean_fwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && numel > 0 && (numel % 4) == 0, "mg_l1_mean_fwd: numel must be a positive multiple of 4");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = numel / 4;
    int grid = ew_grid(nq); if (grid > 1024) grid = 1024;
    if (dtype == MG_BF16) hipLaunchKernelGGL(l1_partial_kernel<uint16_t>, dim3(grid), dim3(NTHR), 0, st, (const uint16_t*)a, (const uint16_t*)b, partial, nq);
    else hipLaunchKernelGGL(l1_partial_kernel<float>, dim3(grid), dim3(NTHR), 0, st, (const float*)a, (const float*)b, partial, nq);
    MG_CHECK_LAUNCH("mg_l1_mean_fwd");
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, grid, 1.0 / (double)numel, out);
    MG_CHECK_LAUNCH("mg_l1_mean_fwd(final)");
    return MG_OK;
}
extern "C" int mg_l1_mean_bwd(const void* a, const void* b, const float* gscale, int32_t dtype, int64_t numel, void* da, void* stream)
{
    MG_CHECK_ARG(a && b && gscale && da, "mg_l1_mean_bwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && numel > 0 && (numel % 4) == 0, "mg_l1_mean_bwd: numel must be a positive multiple of 4");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nq = numel / 4;
    if (dtype == MG_BF16) hipLaunchKernelGGL(l1_bwd_kernel<uint16_t>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const uint16_t*)a, (const uint16_t*)b, gscale, (float)(1.0 / (double)numel), (uint16_t*)da, nq);
    else hipLaunchKernelGGL(l1_bwd_kernel<float>, dim3(ew_grid(nq)), dim3(NTHR), 0, st, (const float*)a, (const float*)b, gscale, (float)(1.0 / (double)numel), (float*)da, nq);
    MG_CHECK_LAUNCH("mg_l1_mean_bwd");
    return MG_OK;
}
