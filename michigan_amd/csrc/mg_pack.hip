// mg_pack.hip -- parameter re-layout between the reference's [Cout][Cin][kh*kw] fp32 tensors and the
// GEMM images the conv kernels read (one launch instead of a permute / pad / cast / cat chain per call).
//
// Row map (GEMM row of output channel co): plain conv  -> co
//                                         fused SPADE -> 64*(co/32) + co%32 (+32 for the beta tensor)
#include "mg_common.h"

namespace {

constexpr int NTHR = 256;
static inline int ew_grid(int64_t n) { int64_t b = (n + NTHR - 1) / NTHR; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// GEMM row r -> (source tensor, channel) ; returns -1 for a padding row
__device__ __forceinline__ int row_to_co(int r, int cout, bool two, int& which)
{
    if (!two) { which = 0; return r < cout ? r : -1; }
    const int b = r >> 6, rem = r & 63;
    which = rem >> 5;
    const int co = b * 32 + (rem & 31);
    return co < cout ? co : -1;
}

// mode 0: dst[t][rows_p][cols_p] = W[co(r)][ci = c][t]          (forward / wgrad image)
// mode 1: dst[t][rows_p][cols_p] = W[co(c)][ci = r][t]          (dgrad image: transposed per tap)
template <typename T>
__global__ void pack_kernel(const float* __restrict__ s0, const float* __restrict__ s1, T* __restrict__ dst,
                            int cout, int cin, int taps, int rows_p, int cols_p, int mode)
{
    const bool two = s1 != nullptr;
    const int64_t total = (int64_t)taps * rows_p * cols_p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols_p);
        const int r = (int)((i / cols_p) % rows_p);
        const int t = (int)(i / ((int64_t)cols_p * rows_p));
        int which = 0;
        const int g = mode == 0 ? r : c;            // GEMM output-channel index
        const int ci = mode == 0 ? c : r;
        const int co = row_to_co(g, cout, two, which);
        float v = 0.f;
        if (co >= 0 && ci < cin) v = (which ? s1 : s0)[((size_t)co * cin + ci) * taps + t];
        ET<T>::store1(dst + i, v);
    }
}

// dW in GEMM order [taps][rows][cols] fp32 -> reference layout [cout][cin][taps] (one or two tensors)
__global__ void unpack_kernel(const float* __restrict__ dw, float* __restrict__ d0, float* __restrict__ d1,
                              int cout, int cin, int taps, int rows, int cols)
{
    const bool two = d1 != nullptr;
    const int ntens = two ? 2 : 1;
    const int64_t total = (int64_t)ntens * cout * cin * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const int ci = (int)((i / taps) % cin);
        const int co = (int)((i / ((int64_t)taps * cin)) % cout);
        const int which = (int)(i / ((int64_t)taps * cin * cout));
        const int r = two ? 64 * (co >> 5) + (co & 31) + 32 * which : co;
        (which ? d1 : d0)[((size_t)co * cin + ci) * taps + t] = dw[((size_t)t * rows + r) * cols + ci];
    }
}

}  // namespace

extern "C" int mg_pack_weight(const float* w0, const float* w1, void* dst, int32_t dtype, int32_t cout, int32_t cin,
                              int32_t taps, int32_t rows_p, int32_t cols_p, int32_t mode, void* stream)
{
    MG_CHECK_ARG(w0 && dst, "mg_pack_weight: null pointer");
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, "mg_pack_weight: bad dtype");
    MG_CHECK_ARG(cout > 0 && cin > 0 && taps > 0 && rows_p > 0 && cols_p > 0 && (mode == 0 || mode == 1), "mg_pack_weight: bad geometry");
    const int gemm_rows = w1 ? 2 * ((cout + 31) / 32) * 32 : cout;
    MG_CHECK_ARG(mode == 0 ? (rows_p >= gemm_rows && cols_p >= cin) : (cols_p >= gemm_rows && rows_p >= cin),
                 "mg_pack_weight: destination too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)taps * rows_p * cols_p;
    if (dtype == MG_BF16) hipLaunchKernelGGL(pack_kernel<uint16_t>, dim3(ew_grid(total)), dim3(NTHR), 0, st, w0, w1, (uint16_t*)dst, cout, cin, taps, rows_p, cols_p, mode);
    else hipLaunchKernelGGL(pack_kernel<float>, dim3(ew_grid(total)), dim3(NTHR), 0, st, w0, w1, (float*)dst, cout, cin, taps, rows_p, cols_p, mode);
    MG_CHECK_LAUNCH("mg_pack_weight");
    return MG_OK;
}

extern "C" int mg_unpack_wgrad(const float* dw, float* d0, float* d1, int32_t cout, int32_t cin, int32_t taps,
                               int32_t rows, int32_t cols, void* stream)
{
    MG_CHECK_ARG(dw && d0, "mg_unpack_wgrad: null pointer");
    MG_CHECK_ARG(cout > 0 && cin > 0 && taps > 0 && cols >= cin, "mg_unpack_wgrad: bad geometry");
    MG_CHECK_ARG(rows >= (d1 ? 2 * ((cout + 31) / 32) * 32 : cout), "mg_unpack_wgrad: source has too few rows");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)(d1 ? 2 : 1) * cout * cin * taps;
    hipLaunchKernelGGL(unpack_kernel, dim3(ew_grid(total)), dim3(NTHR), 0, st, dw, d0, d1, cout, cin, taps, rows, cols);
    MG_CHECK_LAUNCH("mg_unpack_wgrad");
    return MG_OK;
}
