// mg_loss.hip -- hinge GAN loss with the wide-edge weight mask on the patch discriminators' 1-channel logit maps
// (reference: models/networks/loss.py:60-140).
//
// The eager form is ~25 launches per logit map (nearest resize of the label, two max-pools, interpolate, clamp,
// multiply, mean and their autograd) on 67x67 / 35x35 maps -- all latency.  Here:
//   wide_edge_weight_kernel  the weight mask of ONE logit resolution, built once per batch (it depends only on the
//                            label): w = 1 + (wide - 1) * e, e = max_win(t) - min_win(t) over the k x k window (k =
//                            max(1, int(0.06 h)) from the caller, -inf padded like F.max_pool2d) of the label resized to (h, w),
//                            then the (h + 2p - k + 1)^2 pooled map resized back to (h, w) -- both resizes are
//                            F.interpolate(mode='nearest'): src = min(int(floorf(dst * (float)in / out)), in - 1);
//   hinge_fwd_kernel         loss = -mean(f(x) * w):  mode 0 f = x (generator), 1 f = min(x - 1, 0) (D on real),
//                            2 f = min(-x - 1, 0) (D on fake); one workgroup, fixed-order fp64 finish: deterministic;
//   hinge_bwd_kernel         dx = -g / n * f'(x) * w in the logits' dtype.
// Index work on tiny maps: one thread per element, no LDS tiling needed (each 67x67 map is 18 KiB).
#include "mg_common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out)
{
    const float scale = (float)in / (float)out;                      // torch: compute_scales_value<float>
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__global__ void wide_edge_weight_kernel(const float* __restrict__ label, int N, int Hl, int Wl, int h, int w,
                                        int k, int p, float wide, float* __restrict__ out)
{
    const int ho = h + 2 * p - k + 1, wo = w + 2 * p - k + 1;        // pooled size (h + 1 for even k)
    const int total = N * h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = (i / w) % h, n = i / (w * h);
        const int py = nearest_src(y, ho, h), px = nearest_src(x, wo, w);    // pooled cell this output pixel samples
        float mx = -INFINITY, mn = INFINITY;
        for (int dy = 0; dy < k; ++dy) {
            const int ly = py - p + dy;
            if ((unsigned)ly >= (unsigned)h) continue;
            const int sy = nearest_src(ly, Hl, h);
            for (int dx = 0; dx < k; ++dx) {
                const int lx = px - p + dx;
                if ((unsigned)lx >= (unsigned)w) continue;
                const float t = label[((size_t)n * Hl + sy) * Wl + nearest_src(lx, Wl, w)];
                mx = fmaxf(mx, t);
                mn = fminf(mn, t);
            }
        }
        // out - out2 = max(t) - (1 - max(1 - t)): evaluated with the reference's operation order
        const float e = mx - (1.f - (1.f - mn));
        out[i] = e * wide + (1.f - e);
    }
}

template <typename T>
__device__ __forceinline__ float hinge_term(float x, int mode) { return mode == 0 ? x : fminf((mode == 1 ? x : -x) - 1.f, 0.f); }

template <typename T>
__global__ __launch_bounds__(1024) void hinge_fwd_kernel(const T* __restrict__ x, const float* __restrict__ weight, int64_t n,
                                                         int mode, float* __restrict__ out)
{
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        float v = hinge_term<T>(ET<T>::load1(x + i), mode);
        if (weight) v *= weight[i];
        s += (double)v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)(-red[0] / (double)n);
}

template <typename T>
__global__ void hinge_bwd_kernel(const T* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ g,
                                 int64_t n, int mode, T* __restrict__ dx)
{
    const float s = -g[0] / (float)n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = ET<T>::load1(x + i);
        float d;
        if (mode == 0) d = 1.f;
        else if (mode == 1) d = (v - 1.f < 0.f) ? 1.f : 0.f;         // torch.min(a, 0): gradient to a where a < 0 (ties: half,
        else d = (-v - 1.f < 0.f) ? -1.f : 0.f;                       //   measure zero, not reproduced)
        if (weight) d *= weight[i];
        ET<T>::store1(dx + i, s * d);
    }
}

}  // namespace

extern "C" int mg_wide_edge_weight(const float* label, int32_t N, int32_t Hl, int32_t Wl, int32_t h, int32_t w, int32_t k,
                                   float wide, float* out, void* stream)
{
    MG_CHECK_ARG(label && out, "mg_wide_edge_weight: null pointer");
    MG_CHECK_ARG(N > 0 && Hl > 0 && Wl > 0 && h > 0 && w > 0 && k >= 1 && k <= h && k <= w, "mg_wide_edge_weight: bad geometry");
    const int p = k / 2;                                              // get_wide_edges: p = int(k / 2)
    const int total = N * h * w;
    hipLaunchKernelGGL(wide_edge_weight_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), label, N, Hl, Wl, h, w, k, p, wide, out);
    MG_CHECK_LAUNCH("mg_wide_edge_weight");
    return MG_OK;
}

extern "C" int mg_hinge_fwd(const void* x, const float* weight, int32_t dtype, int64_t n, int32_t mode, float* out, void* stream)
{
    MG_CHECK_ARG(x && out, "mg_hinge_fwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && n > 0 && mode >= 0 && mode <= 2, "mg_hinge_fwd: bad dtype / size / mode");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16) hipLaunchKernelGGL(hinge_fwd_kernel<uint16_t>, dim3(1), dim3(1024), 0, st, (const uint16_t*)x, weight, n, mode, out);
    else hipLaunchKernelGGL(hinge_fwd_kernel<float>, dim3(1), dim3(1024), 0, st, (const float*)x, weight, n, mode, out);
    MG_CHECK_LAUNCH("mg_hinge_fwd");
    return MG_OK;
}

extern "C" int mg_hinge_bwd(const void* x, const float* weight, const float* g, int32_t dtype, int64_t n, int32_t mode, void* dx,
                            void* stream)
{
    MG_CHECK_ARG(x && g && dx, "mg_hinge_bwd: null pointer");
    MG_CHECK_ARG((dtype == MG_F32 || dtype == MG_BF16) && n > 0 && mode >= 0 && mode <= 2, "mg_hinge_bwd: bad dtype / size / mode");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = cdiv(n, 256) > 1024 ? 1024 : cdiv(n, 256);
    if (dtype == MG_BF16) hipLaunchKernelGGL(hinge_bwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, weight, g, n, mode, (uint16_t*)dx);
    else hipLaunchKernelGGL(hinge_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, weight, g, n, mode, (float*)dx);
    MG_CHECK_LAUNCH("mg_hinge_bwd");
    return MG_OK;
}
