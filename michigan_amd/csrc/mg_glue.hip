// mg_glue.hip -- the small per-pixel passes around the hot path's convolutions that the host stack used to spell as chains of
// torch element-wise ops (5-10 us of kernel + ~1.7 us of dependent-launch boundary EACH, ~460 of them per training step:
// tools/count_launches.py).  Every entry below replaces one such chain with one launch; none of them is bandwidth-relevant
// (single-channel masks, 3-channel images, the 16x16x1024 encoder output) -- the lever is the launch count.
//
//   nearest_pyramid_kernel   F.interpolate(mode='nearest') of planar fp32 maps to ALL resolutions a network needs, written
//                            directly as NHWC tensors in the activation dtype with zero-padded channels (SPADE conditioning map:
//                            normalization.py:109 x 18 layers = 7 sizes; hair / background mask pyramids: generator.py:186-219,
//                            encoder.py:331-341): one launch instead of (interpolate + permute-copy + pad) per level;
//   pconv_mask_kernel        the mask half of a partial convolution with a single-channel mask (partialconv2d.py:55-75):
//                            window sum -> mask_ratio * update_mask and update_mask in one pass (was avg_pool, mul, add,
//                            reciprocal, clamp, mul, contiguous);
//   pixel_affine_kernel      y[p, c] = x[p, c] * a[p] (+ bias[c] * b[p]): the partial conv's input masking and its output
//                            renormalisation ((raw - b) * ratio + b) * m' = raw * (ratio m') + b * m';
//   bg_compose_kernel        background-encoder input (encoder.py:288-320): k x k dilation of the hair mask (separable max in LDS),
//                            back = 1 - grown, inp = image * back + noise * (1 - back) as NHWC8 in the activation dtype;
//   orient_loss_*_kernel     the tail of the Gabor orientation loss behind the arg-max kernel (loss.py:352-385): tanh confidence,
//                            (sin 2a, cos 2a) of the winning filter, masked L1 against the label and the confidence term, forward
//                            (two-stage deterministic sums) and backward (d/d confidence response) -- ~45 launches per step before.
#include "mg_common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out)
{
    const float scale = (float)in / (float)out;                      // torch: compute_scales_value<float>, nearest (not -exact)
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// ---- nearest pyramid -------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void nearest_pyramid_kernel(const mg_pyramid_desc d, int64_t total)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int lev = 0;
        int64_t r = i;
#pragma unroll 1
        for (; lev < d.nlev - 1; ++lev) {
            const int64_t cnt = (int64_t)d.N * d.h[lev] * d.w[lev];
            if (r < cnt) break;
            r -= cnt;
        }
        const int h = d.h[lev], w = d.w[lev];
        const int x = (int)(r % w), y = (int)((r / w) % h), n = (int)(r / ((int64_t)w * h));
        const int sy = nearest_src(y, d.H, h), sx = nearest_src(x, d.W, w);
        T* __restrict__ o = reinterpret_cast<T*>(d.out[lev]) + r * d.cout;
        for (int c = 0; c < d.cout; ++c) {
            float v = 0.f;
            if (c < d.nplanes) v = d.plane[c][(int64_t)n * d.nstride[c] + (int64_t)sy * d.W + sx];
            ET<T>::store1(o + c, v);
        }
    }
}

// ---- partial-conv mask chain -----------------------------------------------------------------------------------------------------
__global__ void pconv_mask_kernel(const float* __restrict__ m, int N, int H, int W, int k, int s, int p, int h, int w,
                                  float* __restrict__ scale, float* __restrict__ upd)
{
    const int64_t total = (int64_t)N * h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h), n = (int)(i / ((int64_t)w * h));
        float sum = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const int iy = y * s - p + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = x * s - p + dx;
                if ((unsigned)ix < (unsigned)W) sum += m[((int64_t)n * H + iy) * W + ix];
            }
        }
        // avg_pool2d(count_include_pad=True) * k*k == the window sum up to one rounding of the mean; masks are 0/1 so both are exact
        const float ratio = (float)(k * k) / (sum + 1e-8f);
        const float u = fminf(fmaxf(sum, 0.f), 1.f);
        scale[i] = ratio * u;
        upd[i] = u;
    }
}

// ---- y = x * a[p] (+ bias[c] * b[p]) ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pixel_affine_kernel(const T* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                    const float* __restrict__ b, int64_t P, int C, T* __restrict__ y)
{
    const int cq = C >> 2;
    const int64_t nq = P * cq;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / cq;
        const int c = (int)(i - pix * cq) * 4;
        const float av = a[pix];
        f32x4_t v = ET<T>::load4(x + i * 4);
        if (bias) {
            // the host stack computed addcmul(bias_t * upd_t, raw, scale_t) with every operand ALREADY in the activation dtype: the
            // product bias * upd is rounded to T first, then raw * scale is added with one rounding
            const float bv = b[pix];
            const float s_t = ET<T>::DT == MG_BF16 ? bf2f(f2bf(av)) : av;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = bias[c + j];
                if (ET<T>::DT == MG_BF16) t = bf2f(f2bf(bf2f(f2bf(t)) * bf2f(f2bf(bv))));
                else t = t * bv;
                v[j] = t + v[j] * s_t;
            }
        } else {
            const float s_t = ET<T>::DT == MG_BF16 ? bf2f(f2bf(av)) : av;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= s_t;
        }
        ET<T>::store4(y + i * 4, v);
    }
}

// ---- background-encoder input ----------------------------------------------------------------------------------------------------
constexpr int BG_T = 32, BG_RMAX = 16;
template <typename T>
__global__ __launch_bounds__(256) void bg_compose_kernel(const float* __restrict__ image, const float* __restrict__ noise,
                                                         const float* __restrict__ hair, int64_t hair_nstride, int H, int W, int k,
                                                         int mode, T* __restrict__ inp, float* __restrict__ back)
{
    // mode 0: back = 1 - dilate_k(hair); 1: back = hair plane given as the background mask itself (no dilation, k ignored)
    __shared__ float tile[(BG_T + 2 * BG_RMAX) * (BG_T + 2 * BG_RMAX)];
    __shared__ float rowmax[(BG_T + 2 * BG_RMAX) * BG_T];
    const int r = mode == 0 ? k / 2 : 0;
    const int ext = BG_T + 2 * r;
    const int n = blockIdx.z, y0 = blockIdx.y * BG_T, x0 = blockIdx.x * BG_T;
    const float* __restrict__ hp = hair + (int64_t)n * hair_nstride;
    for (int i = threadIdx.x; i < ext * ext; i += 256) {
        const int ly = i / ext, lx = i - ly * ext;
        const int gy = y0 + ly - r, gx = x0 + lx - r;
        tile[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? hp[(int64_t)gy * W + gx] : -INFINITY;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ext * BG_T; i += 256) {
        const int ly = i / BG_T, lx = i - ly * BG_T;
        float m = -INFINITY;
        for (int d = 0; d < 2 * r + 1; ++d) m = fmaxf(m, tile[ly * ext + lx + d]);
        rowmax[i] = m;
    }
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    for (int i = threadIdx.x; i < BG_T * BG_T; i += 256) {
        const int ly = i / BG_T, lx = i - ly * BG_T;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float m = -INFINITY;
        for (int d = 0; d < 2 * r + 1; ++d) m = fmaxf(m, rowmax[(ly + d) * BG_T + lx]);
        const float bk = mode == 0 ? 1.f - m : m;
        const int64_t pix = (int64_t)gy * W + gx;
        back[(int64_t)n * HW + pix] = bk;
        f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = ((int64_t)n * 3 + c) * HW + pix;
            lo[c] = noise ? (image ? image[o] * bk + noise[o] * (1.f - bk) : noise[o]) : image[o] * bk;
        }
        T* __restrict__ op = inp + ((int64_t)n * HW + pix) * 8;
        ET<T>::store4(op, lo);
        ET<T>::store4(op + 4, hi4);
    }
}

// ---- orientation-loss tail -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void orient_terms(float craw, int idx, const float* __restrict__ label, int label_ch, int64_t HW, int64_t pix,
                                             float hairv, float& confidence, float& d0, float& d1, float& f0, float& f1)
{
    confidence = (tanhf(craw) + 1.f) / 2.f;
    const float ang = (float)idx * (float)(M_PI / 32.0);
    const float s2 = sinf(2.f * ang), c2 = cosf(2.f * ang);
    float l0, l1;
    if (label_ch == 2) { l0 = label[pix]; l1 = label[HW + pix]; }
    else { const float lab = label[pix] / 255.f * (float)M_PI; l0 = sinf(2.f * lab); l1 = cosf(2.f * lab); }
    f0 = s2; f1 = c2;
    d0 = s2 * confidence * hairv - l0 * hairv;
    d1 = c2 * confidence * hairv - l1 * hairv;
}

__global__ __launch_bounds__(256) void orient_loss_partial_kernel(const float* __restrict__ conf_raw, const uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ label, int label_ch, int64_t label_nstride,
                                                                  const float* __restrict__ hair, int64_t hair_nstride, int N, int64_t HW,
                                                                  float* __restrict__ ws)
{
    __shared__ float red[3][4];
    float s_abs = 0.f, s_log = 0.f, s_hair = 0.f;
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / HW);
        const int64_t pix = i - (int64_t)n * HW;
        const float hv = hair[(int64_t)n * hair_nstride + pix];
        float cf, d0, d1, f0, f1;
        orient_terms(conf_raw[i], idx[i], label + (int64_t)n * label_nstride, label_ch, HW, pix, hv, cf, d0, d1, f0, f1);
        s_abs += fabsf(d0) + fabsf(d1);
        s_log += logf(fminf(fmaxf(cf, 0.001f), 1.f)) * hv;
        s_hair += hv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_abs += __shfl_down(s_abs, o, 64); s_log += __shfl_down(s_log, o, 64); s_hair += __shfl_down(s_hair, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_abs; red[1][threadIdx.x >> 6] = s_log; red[2][threadIdx.x >> 6] = s_hair; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int w = 0; w < 4; ++w) t += red[threadIdx.x][w];
        ws[threadIdx.x * 1024 + blockIdx.x] = t;
    }
}

__global__ void orient_loss_final_kernel(const float* __restrict__ ws, int nblk, double inv_count, float* __restrict__ out)
{
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int q = 0; q < 3; ++q) s[q] += (double)ws[q * 1024 + i];
    for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] * inv_count);                    // F.l1_loss over N*2*H*W elements
        out[1] = (float)(-red[1][0] / red[2][0]);                   // -sum(log(conf) * hair) / sum(hair)
        out[2] = (float)red[2][0];
    }
}

__global__ void orient_loss_bwd_kernel(const float* __restrict__ conf_raw, const uint8_t* __restrict__ idx,
                                       const float* __restrict__ label, int label_ch, int64_t label_nstride,
                                       const float* __restrict__ hair, int64_t hair_nstride, const float* __restrict__ g_orient,
                                       const float* __restrict__ g_conf, const float* __restrict__ fwd_out, int N, int64_t HW,
                                       float* __restrict__ dconf)
{
    const int64_t total = (int64_t)N * HW;
    const float g_or = g_orient ? g_orient[0] / (float)(2 * total) : 0.f, g_cf = g_conf ? -g_conf[0] / fwd_out[2] : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int64_t pix = i - (int64_t)n * HW;
        const float hv = hair[(int64_t)n * hair_nstride + pix];
        float cf, d0, d1, f0, f1;
        const float craw = conf_raw[i];
        orient_terms(craw, idx[i], label + (int64_t)n * label_nstride, label_ch, HW, pix, hv, cf, d0, d1, f0, f1);
        const float sg0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f), sg1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
        float dcf = g_or * (sg0 * f0 + sg1 * f1) * hv;
        if (cf >= 0.001f && cf <= 1.f) dcf += g_cf * hv / cf;      // clamp passes the gradient inside [min, max]
        const float t = tanhf(craw);
        dconf[i] = dcf * (1.f - t * t) * 0.5f;
    }
}

// ---- masked mean over a region, broadcast into another region (ImageEncoder3's tail, encoder.py:211-220) -----------------------------
// out[n, q, c] = w_out[n, q] * (sum_p x[n, p, c] * w_in[n, p]) / max(sum_p w_norm[n, p], 1): the mean feature of the reference hair region
// written over the target hair region (forward: w_in = w_norm = lref, w_out = ltag) and its adjoint (backward: w_in = ltag, w_out = w_norm =
// lref).  A block owns one sample and 16 channel quads; 16 thread rows split the pixels.  P <= a few thousand (the 16 x 16 latent).
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_fill_kernel(const T* __restrict__ x, const float* __restrict__ w_in, const float* __restrict__ w_out,
                                                               const float* __restrict__ w_norm, int P, int C, float* __restrict__ out)
{
    __shared__ double red[16][16][4];
    __shared__ double nrm[16];
    const int n = blockIdx.y, tq = threadIdx.x & 15, tp = threadIdx.x >> 4;
    const int c = (blockIdx.x * 16 + tq) * 4;
    const bool cv = c < C;
    double s[4] = {0.0, 0.0, 0.0, 0.0};                       // fp64 sums: a few thousand elements per block, free; the latent statistics are ill-conditioned enough
    double a = 0.0;
    for (int p = tp; p < P; p += 16) {
        const float wi = w_in[(size_t)n * P + p];
        if (tq == 0) a += (double)w_norm[(size_t)n * P + p];
        if (cv && wi != 0.f) {
            const f32x4_t v = ET<T>::load4(x + ((size_t)n * P + p) * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (double)v[j] * (double)wi;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[tp][tq][j] = s[j];
    if (tq == 0) nrm[tp] = a;
    __syncthreads();
    double area = 0.0;
    for (int r = 0; r < 16; ++r) area += nrm[r];
    area = area > 1.0 ? area : 1.0;
    f32x4_t m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double t = 0.0;
        for (int r = 0; r < 16; ++r) t += red[r][tq][j];
        m[j] = (float)(t / area);
    }
    if (!cv) return;
    for (int q = tp; q < P; q += 16) {
        const float wo = w_out[(size_t)n * P + q];
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = m[j] * wo;
        *reinterpret_cast<f32x4_t*>(out + ((size_t)n * P + q) * C + c) = o;
    }
}

inline int ew_grid(int64_t n, int thr = 256, int cap = 4096) { const int64_t g = (n + thr - 1) / thr; return (int)(g > cap ? cap : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int mg_nearest_pyramid(const mg_pyramid_desc* d, void* stream)
{
    MG_CHECK_ARG(d, "mg_nearest_pyramid: null descriptor");
    MG_CHECK_ARG(d->nplanes >= 1 && d->nplanes <= 8 && d->nlev >= 1 && d->nlev <= 8 && d->N > 0 && d->H > 0 && d->W > 0,
                 "mg_nearest_pyramid: bad geometry");
    MG_CHECK_ARG(d->cout >= d->nplanes && d->cout <= 8 && (d->dtype == MG_F32 || d->dtype == MG_BF16), "mg_nearest_pyramid: bad output channels / dtype");
    int64_t total = 0;
    for (int l = 0; l < d->nlev; ++l) {
        MG_CHECK_ARG(d->h[l] > 0 && d->w[l] > 0 && d->out[l], "mg_nearest_pyramid: bad level %d", l);
        total += (int64_t)d->N * d->h[l] * d->w[l];
    }
    for (int c = 0; c < d->nplanes; ++c) MG_CHECK_ARG(d->plane[c], "mg_nearest_pyramid: null plane %d", c);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MG_BF16) hipLaunchKernelGGL(nearest_pyramid_kernel<uint16_t>, dim3(ew_grid(total)), dim3(256), 0, st, *d, total);
    else hipLaunchKernelGGL(nearest_pyramid_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, st, *d, total);
    MG_CHECK_LAUNCH("mg_nearest_pyramid");
    return MG_OK;
}

extern "C" int mg_pconv_mask(const float* mask_in, int32_t N, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p,
                             float* scale, float* upd, void* stream)
{
    MG_CHECK_ARG(mask_in && scale && upd, "mg_pconv_mask: null pointer");
    MG_CHECK_ARG(N > 0 && H > 0 && W > 0 && k >= 1 && s >= 1 && p >= 0 && H + 2 * p >= k && W + 2 * p >= k, "mg_pconv_mask: bad geometry");
    const int h = (H + 2 * p - k) / s + 1, w = (W + 2 * p - k) / s + 1;
    hipLaunchKernelGGL(pconv_mask_kernel, dim3(ew_grid((int64_t)N * h * w)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       mask_in, N, H, W, k, s, p, h, w, scale, upd);
    MG_CHECK_LAUNCH("mg_pconv_mask");
    return MG_OK;
}

extern "C" int mg_pixel_affine(const void* x, const float* a, const float* bias, const float* b, int32_t dtype, int64_t P, int32_t C,
                               void* y, void* stream)
{
    MG_CHECK_ARG(x && a && y, "mg_pixel_affine: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && P > 0 && C > 0 && (C & 3) == 0, "mg_pixel_affine: C must be a positive multiple of 4");
    MG_CHECK_ARG((bias == nullptr) == (b == nullptr), "mg_pixel_affine: bias and b come together");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = ew_grid(P * (C >> 2));
    if (dtype == MG_BF16) hipLaunchKernelGGL(pixel_affine_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, a, bias, b, P, C, (uint16_t*)y);
    else hipLaunchKernelGGL(pixel_affine_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, a, bias, b, P, C, (float*)y);
    MG_CHECK_LAUNCH("mg_pixel_affine");
    return MG_OK;
}

extern "C" int mg_bg_compose(const float* image, const float* noise, const float* hair, int64_t hair_nstride, int32_t dtype,
                             int32_t N, int32_t H, int32_t W, int32_t k, int32_t mode, void* inp, float* back, void* stream)
{
    MG_CHECK_ARG((image || noise) && hair && inp && back, "mg_bg_compose: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && N > 0 && H > 0 && W > 0, "mg_bg_compose: bad dtype / geometry");
    MG_CHECK_ARG(mode == 1 || (mode == 0 && k >= 1 && (k & 1) == 1 && k / 2 <= BG_RMAX), "mg_bg_compose: the dilation window must be odd and <= %d", 2 * BG_RMAX + 1);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((W + BG_T - 1) / BG_T, (H + BG_T - 1) / BG_T, N);
    if (dtype == MG_BF16) hipLaunchKernelGGL(bg_compose_kernel<uint16_t>, grid, dim3(256), 0, st, image, noise, hair, hair_nstride, H, W, k, mode, (uint16_t*)inp, back);
    else hipLaunchKernelGGL(bg_compose_kernel<float>, grid, dim3(256), 0, st, image, noise, hair, hair_nstride, H, W, k, mode, (float*)inp, back);
    MG_CHECK_LAUNCH("mg_bg_compose");
    return MG_OK;
}

extern "C" int mg_orient_loss_fwd(const float* conf_raw, const uint8_t* idx, const float* label, int32_t label_ch, int64_t label_nstride,
                                  const float* hair, int64_t hair_nstride, int32_t N, int64_t HW, float* out, float* ws, void* stream)
{
    MG_CHECK_ARG(conf_raw && idx && label && hair && out && ws, "mg_orient_loss_fwd: null pointer");
    MG_CHECK_ARG((label_ch == 1 || label_ch == 2) && N > 0 && HW > 0, "mg_orient_loss_fwd: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = ew_grid((int64_t)N * HW, 256, 1024);
    hipLaunchKernelGGL(orient_loss_partial_kernel, dim3(grid), dim3(256), 0, st, conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, N, HW, ws);
    MG_CHECK_LAUNCH("mg_orient_loss_fwd");
    hipLaunchKernelGGL(orient_loss_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, grid, 1.0 / (2.0 * (double)N * (double)HW), out);
    MG_CHECK_LAUNCH("mg_orient_loss_fwd(final)");
    return MG_OK;
}

extern "C" int mg_orient_loss_bwd(const float* conf_raw, const uint8_t* idx, const float* label, int32_t label_ch, int64_t label_nstride,
                                  const float* hair, int64_t hair_nstride, const float* g_orient, const float* g_conf, const float* fwd_out,
                                  int32_t N, int64_t HW, float* dconf, void* stream)
{
    MG_CHECK_ARG(conf_raw && idx && label && hair && fwd_out && dconf, "mg_orient_loss_bwd: null pointer");
    MG_CHECK_ARG((label_ch == 1 || label_ch == 2) && N > 0 && HW > 0, "mg_orient_loss_bwd: bad geometry");
    hipLaunchKernelGGL(orient_loss_bwd_kernel, dim3(ew_grid((int64_t)N * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       conf_raw, idx, label, label_ch, label_nstride, hair, hair_nstride, g_orient, g_conf, fwd_out, N, HW, dconf);
    MG_CHECK_LAUNCH("mg_orient_loss_bwd");
    return MG_OK;
}

extern "C" int mg_masked_mean_fill(const void* x, const float* w_in, const float* w_out, const float* w_norm, int32_t dtype, int32_t N,
                                   int32_t P, int32_t C, float* out, void* stream)
{
    MG_CHECK_ARG(x && w_in && w_out && w_norm && out, "mg_masked_mean_fill: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && N > 0 && P > 0 && C > 0 && (C & 3) == 0, "mg_masked_mean_fill: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((C / 4 + 15) / 16, N);
    if (dtype == MG_BF16) hipLaunchKernelGGL(masked_mean_fill_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)x, w_in, w_out, w_norm, P, C, out);
    else hipLaunchKernelGGL(masked_mean_fill_kernel<float>, grid, dim3(256), 0, st, (const float*)x, w_in, w_out, w_norm, P, C, out);
    MG_CHECK_LAUNCH("mg_masked_mean_fill");
    return MG_OK;
}
