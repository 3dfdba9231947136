// mg_gabor.hip -- orientation-loss filter bank (reference: models/networks/loss.py:215-243,274-313,352-385).
//
// L1OLoss runs 32 oriented 17x17 Gabor filters over the gray version of the generated image, keeps per pixel
// the strongest non-negative response and its index, and turns them into a confidence-weighted orientation.
// The reference does this as 32 separate F.conv2d calls + cat + clamp + argmax + max (~100 launches over a
// [N,32,H,W] fp32 tensor).  Here one launch computes the 32 x 289 x pixels contraction on the matrix cores in
// exact fp32 (v_mfma_f32_32x32x2_f32: rows = filters, columns = 32 pixels of a tile row, K = taps) from an
// LDS-staged (16+16)x(32+16) gray patch, and reduces max / first-arg-max in registers; only
// conf_raw[N,H,W] (fp32) and idx[N,H,W] (u8) reach HBM.  The backward is the gather form of the adjoint:
// dgray[p'] = sum_{p in 17x17 around p'} g[p] * K_{idx[p]}(p' - p), then the gray weights back to RGB.
#include "mg_common.h"

namespace {

constexpr int KS = 17, KR = 8, NF = 32, NTAP = KS * KS;          // 289 taps, padded to 290 for K = 2 steps
constexpr int TGH = 16, TGW = 32;                                  // pixel tile: 16 rows x 32 columns, 4 waves x 4 rows
constexpr int PGH = TGH + 2 * KR, PGW = TGW + 2 * KR;              // 32 x 48 gray patch
constexpr float GR = 0.299f, GG = 0.587f, GB = 0.144f;             // (sic) the reference's gray weights

template <typename T>
__device__ __forceinline__ float gray_at(const T* img, int n, int y, int x, int H, int W, int C)
{
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0.f;     // zero padding of the GRAY image
    const T* p = img + ((size_t)(n * H + y) * W + x) * C;
    const float r = ET<T>::load1(p), g = ET<T>::load1(p + 1), b = ET<T>::load1(p + 2);
    return GR * ((r + 1.f) * 0.5f * 255.f) + GG * ((g + 1.f) * 0.5f * 255.f) + GB * ((b + 1.f) * 0.5f * 255.f);
}

template <typename T>
__global__ __launch_bounds__(256) void gabor_fwd_kernel(const T* __restrict__ img, const float* __restrict__ bank,
                                                        float* __restrict__ conf, uint8_t* __restrict__ idx,
                                                        int N, int H, int W, int C, int tiles_x, int tiles_y)
{
    __shared__ float patch[PGH * PGW];                 // 6 KiB
    __shared__ float wts[NF * (NTAP + 1)];             // 32 x 290 (tap 289 = 0), 36.25 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int y0 = ty * TGH, x0 = tx * TGW;

    for (int i = tid; i < PGH * PGW; i += 256) {
        const int py = i / PGW, px = i - py * PGW;
        patch[i] = gray_at(img, n, y0 + py - KR, x0 + px - KR, H, W, C);
    }
    for (int i = tid; i < NF * (NTAP + 1); i += 256) {
        const int f = i / (NTAP + 1), k = i - f * (NTAP + 1);
        wts[i] = k < NTAP ? bank[f * NTAP + k] : 0.f;
    }
    __syncthreads();

    // wave w owns tile rows 4w .. 4w+3; one MFMA column tile = the 32 pixels of one row
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // K step s covers taps 2s (hi = 0) and 2s+1 (hi = 1)
        for (int s = 0; s < (NTAP + 1) / 2; ++s) {
            const int k = 2 * s + hi;
            const int ky = k / KS, kx = k - ky * KS;                       // k = 289 -> ky = 17: weight is 0, address stays inside the patch? no -> clamp
            const float a = wts[l31 * (NTAP + 1) + k];
            const int pyy = (k < NTAP) ? row + ky : row;
            const int pxx = (k < NTAP) ? l31 + kx : l31;
            const float b = patch[pyy * PGW + pxx];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        // lane holds, for pixel column l31, filters (r&3) + 8*(r>>2) + 4*hi; clamp < 0 to 0, first arg-max
        float best = -1.f; int bi = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float v = fmaxf(acc[r], 0.f);
            if (v > best || (v == best && f < bi)) { best = v; bi = f; }
        }
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        const int y = y0 + row, x = x0 + l31;
        if (hi == 0 && y < H && x < W) {
            const size_t o = (size_t)(n * H + y) * W + x;
            conf[o] = best;
            idx[o] = (uint8_t)bi;
        }
    }
}

// dimg[n,y,x,0..2] = w_c * 127.5 * sum_{dy,dx} g[y-dy+8.., ...]: gather over the 17x17 pixels p whose window covers p'
template <typename T>
__global__ __launch_bounds__(256) void gabor_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ idx,
                                                        const float* __restrict__ bank, T* __restrict__ dimg,
                                                        int N, int H, int W, int C, int tiles_x, int tiles_y)
{
    __shared__ float gp[PGH * PGW];
    __shared__ uint8_t ip[PGH * PGW];
    __shared__ float wts[NF * NTAP];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int y0 = ty * TGH, x0 = tx * TGW;
    for (int i = tid; i < PGH * PGW; i += 256) {
        const int py = i / PGW, px = i - py * PGW;
        const int y = y0 + py - KR, x = x0 + px - KR;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const size_t o = (size_t)(n * H + (ok ? y : 0)) * W + (ok ? x : 0);
        gp[i] = ok ? g[o] : 0.f;
        ip[i] = ok ? idx[o] : 0;
    }
    for (int i = tid; i < NF * NTAP; i += 256) wts[i] = bank[i];
    __syncthreads();
    for (int i = tid; i < TGH * TGW; i += 256) {
        const int py = i / TGW, px = i - py * TGW;
        const int y = y0 + py, x = x0 + px;
        if (y >= H || x >= W) continue;
        // response at p used gray[p + (ky-8, kx-8)] * K[ky][kx]; so gray[p'] feeds p = p' - (ky-8, kx-8)
        float s = 0.f;
        for (int ky = 0; ky < KS; ++ky) {
            const int qy = py + KR - (ky - KR);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int qx = px + KR - (kx - KR);
                const int q = qy * PGW + qx;
                const float gv = gp[q];
                if (gv != 0.f) s += gv * wts[(int)ip[q] * NTAP + ky * KS + kx];
            }
        }
        T* o = dimg + ((size_t)(n * H + y) * W + x) * C;
        ET<T>::store1(o, s * GR * 127.5f);
        ET<T>::store1(o + 1, s * GG * 127.5f);
        ET<T>::store1(o + 2, s * GB * 127.5f);
        for (int c = 3; c < C; ++c) ET<T>::store1(o + c, 0.f);
    }
}

}  // namespace

#define MG_GABOR_CHECK(name) \
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, name ": bad dtype"); \
    MG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C >= 3, name ": bad geometry N=%d H=%d W=%d C=%d", N, H, W, C)

extern "C" int mg_gabor_argmax_fwd(const void* img, const float* bank, float* conf, uint8_t* idx, int32_t dtype,
                                   int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_GABOR_CHECK("mg_gabor_argmax_fwd");
    MG_CHECK_ARG(img && bank && conf && idx, "mg_gabor_argmax_fwd: null pointer");
    const int tx = (W + TGW - 1) / TGW, ty = (H + TGH - 1) / TGH;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16) hipLaunchKernelGGL(gabor_fwd_kernel<uint16_t>, dim3(N * tx * ty), dim3(256), 0, st, (const uint16_t*)img, bank, conf, idx, N, H, W, C, tx, ty);
    else hipLaunchKernelGGL(gabor_fwd_kernel<float>, dim3(N * tx * ty), dim3(256), 0, st, (const float*)img, bank, conf, idx, N, H, W, C, tx, ty);
    MG_CHECK_LAUNCH("mg_gabor_argmax_fwd");
    return MG_OK;
}

extern "C" int mg_gabor_argmax_bwd(const float* dconf, const uint8_t* idx, const float* bank, void* dimg, int32_t dtype,
                                   int32_t N, int32_t H, int32_t W, int32_t C, void* stream)
{
    MG_GABOR_CHECK("mg_gabor_argmax_bwd");
    MG_CHECK_ARG(dconf && idx && bank && dimg, "mg_gabor_argmax_bwd: null pointer");
    const int tx = (W + TGW - 1) / TGW, ty = (H + TGH - 1) / TGH;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16) hipLaunchKernelGGL(gabor_bwd_kernel<uint16_t>, dim3(N * tx * ty), dim3(256), 0, st, dconf, idx, bank, (uint16_t*)dimg, N, H, W, C, tx, ty);
    else hipLaunchKernelGGL(gabor_bwd_kernel<float>, dim3(N * tx * ty), dim3(256), 0, st, dconf, idx, bank, (float*)dimg, N, H, W, C, tx, ty);
    MG_CHECK_LAUNCH("mg_gabor_argmax_bwd");
    return MG_OK;
}
