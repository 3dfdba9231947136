// mg_spectral.hip -- the small-vector and weight-sized pieces of torch.nn.utils.spectral_norm (1 power iteration,
// dim 0) around the two rocBLAS gemv calls:  v <- normalize(W^T u), u <- normalize(W v), sigma = u . (W v),
// W_sn = W / sigma, and the gradient through W / sigma with u, v held constant
//   dW = (g - (sum g*W_sn) u v^T) / sigma.
// The eager form is ~16 launches per layer forward and ~10 weight-sized passes backward (4.4 ms of a 87 ms step,
// tools/sn_cost.py); this is 5 launches forward and 2 backward.
#include "mg_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// dst (and dst2) = t / max(||t||_2, eps);  sigma = dst . t   (one 1024-thread block; n is a conv dimension, <= ~20k)
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* __restrict__ t, int n, float eps,
                                                          float* __restrict__ dst, float* __restrict__ dst2, float* __restrict__ sigma)
{
    __shared__ float red[16];
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss += t[i] * t[i];
    ss = block_sum(ss, red);
    const float denom = fmaxf(sqrtf(ss), eps);
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float q = t[i] / denom;
        dst[i] = q;
        if (dst2) dst2[i] = q;
        dot += q * t[i];
    }
    if (sigma) {
        dot = block_sum(dot, red);
        if (threadIdx.x == 0) *sigma = dot;
    }
}

__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ out, int64_t n4)
{
    const float s = *sigma;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4_t v = reinterpret_cast<const f32x4_t*>(w)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] / s;
        reinterpret_cast<f32x4_t*>(out)[i] = v;
    }
}

__global__ void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                              const float* __restrict__ s, const float* __restrict__ sigma, float* __restrict__ out,
                              int rows, int cols)
{
    const float sv = *s, sg = *sigma;
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        out[i] = (g[i] - sv * u[r] * v[c]) / sg;
    }
}

static inline int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int mg_sn_normalize(const float* t, int32_t n, float eps, float* dst, float* dst2, float* sigma, void* stream)
{
    MG_CHECK_ARG(t && dst && n > 0, "mg_sn_normalize: bad arguments");
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), t, n, eps, dst, dst2, sigma);
    MG_CHECK_LAUNCH("mg_sn_normalize");
    return MG_OK;
}

extern "C" int mg_sn_scale(const float* w, const float* sigma, float* out, int64_t numel, void* stream)
{
    MG_CHECK_ARG(w && sigma && out && numel > 0 && (numel % 4) == 0, "mg_sn_scale: bad arguments (numel must be a multiple of 4)");
    hipLaunchKernelGGL(sn_scale_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, sigma, out, numel / 4);
    MG_CHECK_LAUNCH("mg_sn_scale");
    return MG_OK;
}

extern "C" int mg_sn_bwd(const float* g, const float* u, const float* v, const float* s, const float* sigma, float* out,
                         int32_t rows, int32_t cols, void* stream)
{
    MG_CHECK_ARG(g && u && v && s && sigma && out && rows > 0 && cols > 0, "mg_sn_bwd: bad arguments");
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       g, u, v, s, sigma, out, rows, cols);
    MG_CHECK_LAUNCH("mg_sn_bwd");
    return MG_OK;
}
