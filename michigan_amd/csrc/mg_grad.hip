// mg_grad.hip -- batched "drain" of the GEMM-order weight-gradient arena into the reference-layout gradient arena.
//
// Before: every conv backward zero-filled a scratch buffer, ran its wgrad kernel (fp32 atomics, GEMM order
// [taps][rows][cols]), launched an unpack kernel into a fresh reference-layout tensor, for spectral-normed layers a dot
// + the sigma-backward kernel, and autograd then added the result into the flat gradient arena: ~375 launches of
// 5-10 us per training step (fills 95, unpack 74, dot + sn_bwd 48, adds 158; 2.9 ms of a 75 ms step, rocprofv3
// profiles/r02a_kernel_stats.csv).  Now the wgrad kernels accumulate straight into a persistent GEMM-order arena that
// stays resident (optim.FlatAdam.gemm) and two (three with spectral norm) launches per optimiser step move everything:
//   grad_sn_dot_kernel   spectral-normed slots only:  per-tile partials of s = sum(g * W_sn); grad_sn_finish_kernel adds them in a
//                        fixed order (bit-identical on every rank of a data-parallel job)
//   grad_drain_kernel    every slot:  dst[co][ci][t] += SN ? (g - s u[co] v[ci*T+t]) / sigma : g ;  gemm <- 0 ;  biases
// (reference: torch.autograd of nn.Conv2d weights + torch.nn.utils.spectral_norm, architecture.py:31-42,
//  normalization.py:28-29,94-99; optimiser arenas pix2pix_model.py:137-145).
//
// Tiling: a workgroup owns (slot, tensor, CO_PER output channels, 64 input channels, all T taps).  The GEMM image is
// read tap by tap as 256-byte rows (coalesced), transposed through LDS (pitch T|1: conflict-free for T = 1, 9, 16, 49)
// and written / read-modify-written as one contiguous 64*T-float run of the reference tensor.  HBM-bound: ~16 B per
// gradient element (+4 B of W_sn for spectral-normed layers); no MFMA.
#include "mg_common.h"

namespace {

constexpr int CO_PER = 4;                 // output channels per workgroup (sequential)
constexpr int MAXT = 49;                  // 7x7 is the largest window on the path

struct Tile { int slot, which, co0, ci0, bias; };

__device__ __forceinline__ Tile decode(const mg_grad_slot* __restrict__ tab, const int32_t* __restrict__ block_slot, int b)
{
    Tile t;
    t.slot = block_slot[b];
    const mg_grad_slot& s = tab[t.slot];
    int rel = b - (int)s.first_block;
    const int ci_blocks = (s.cin + 63) >> 6, co_groups = (s.cout + CO_PER - 1) / CO_PER;
    const int per_tensor = ci_blocks * co_groups;
    const int ntens = s.dst1 ? 2 : 1;
    t.bias = rel >= per_tensor * ntens;
    t.which = t.bias ? 0 : rel / per_tensor;
    rel -= t.which * per_tensor;
    t.co0 = (rel / ci_blocks) * CO_PER;
    t.ci0 = (rel % ci_blocks) << 6;
    return t;
}

// GEMM row of output channel co (plain: co; fused SPADE pair: [32 gamma | 32 beta] blocks)
__device__ __forceinline__ int gemm_row(int co, int which, bool two) { return two ? 64 * (co >> 5) + (co & 31) + 32 * which : co; }

// load the tile's [T][64] GEMM values of channel `co` into LDS as lds[ci_local * pitch + t]; optionally zero the source
template <bool ZERO>
__device__ __forceinline__ void load_tile(const mg_grad_slot& s, int which, int co, int ci0, float* lds, int pitch)
{
    const int T = s.taps, r = gemm_row(co, which, s.dst1 != nullptr);
    const int c = threadIdx.x & 63, ci = ci0 + c;
    for (int t = threadIdx.x >> 6; t < T; t += 4) {
        float v = 0.f;
        if (ci < s.cin) {
            float* p = s.swapped ? s.gemm + ((size_t)t * s.cols + ci) * s.rows + r      // swapped roles: [T][cin][rows]
                                 : s.gemm + ((size_t)t * s.rows + r) * s.cols + ci;
            v = *p;
            if (ZERO) *p = 0.f;
        }
        lds[c * pitch + t] = v;
    }
}

__global__ __launch_bounds__(256) void grad_sn_dot_kernel(const mg_grad_slot* __restrict__ tab, const int32_t* __restrict__ block_slot,
                                                          double* __restrict__ partial)
{
    __shared__ float lds[64 * (MAXT + 1)];
    __shared__ double red[4];
    const Tile tl = decode(tab, block_slot, blockIdx.x);
    const mg_grad_slot& s = tab[tl.slot];
    if (tl.bias || !s.w_sn) { if (threadIdx.x == 0) partial[blockIdx.x] = 0.0; return; }
    const int T = s.taps, pitch = T | 1;
    const int ncol = min(64, s.cin - tl.ci0), run = ncol * T;
    // element e of the contiguous run <-> (column c = e / T, tap t = e % T), walked without divisions in the loop
    const int c0 = (int)threadIdx.x / T, t0 = (int)threadIdx.x - c0 * T, dc = 256 / T, dt = 256 - dc * T;
    double acc = 0.0;
    for (int k = 0; k < CO_PER && tl.co0 + k < s.cout; ++k) {
        const int co = tl.co0 + k;
        __syncthreads();
        load_tile<false>(s, 0, co, tl.ci0, lds, pitch);
        __syncthreads();
        const float* w = s.w_sn + ((size_t)co * s.cin + tl.ci0) * T;
        float part = 0.f;                                        // <= ceil(64 * 49 / 256) = 13 products per thread and channel
        for (int e = threadIdx.x, c = c0, t = t0; e < run; e += 256) {
            part += lds[c * pitch + t] * w[e];
            c += dc; t += dt;
            if (t >= T) { t -= T; ++c; }
        }
        acc += (double)part;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// s[slot] = sum of the slot's per-workgroup partials in a FIXED order (one workgroup per slot): the value every rank of a
// data-parallel job computes from the same all-reduced gradients must be bit-identical, or the replicas' weights drift apart
__global__ __launch_bounds__(256) void grad_sn_finish_kernel(const mg_grad_slot* __restrict__ tab, const double* __restrict__ partial, int nblocks_total,
                                                             const int32_t* __restrict__ block_slot)
{
    __shared__ double red[256];
    const mg_grad_slot& s = tab[blockIdx.x];
    if (!s.w_sn) return;
    const int b0 = (int)s.first_block;
    double acc = 0.0;
    for (int b = b0 + threadIdx.x; b < nblocks_total && block_slot[b] == (int)blockIdx.x; b += 256) acc += partial[b];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *s.s = red[0];
}

__global__ __launch_bounds__(256) void grad_drain_kernel(const mg_grad_slot* __restrict__ tab, const int32_t* __restrict__ block_slot)
{
    __shared__ float lds[64 * (MAXT + 1)];
    const Tile tl = decode(tab, block_slot, blockIdx.x);
    const mg_grad_slot& s = tab[tl.slot];
    if (tl.bias) {
        if (!s.dbias_gemm) return;
        const bool two = s.dbias1 != nullptr;
        for (int co = threadIdx.x; co < s.cout; co += 256) {
            for (int which = 0; which < (two ? 2 : 1); ++which) {
                float* p = s.dbias_gemm + gemm_row(co, which, two);
                (which ? s.dbias1 : s.dbias0)[co] += *p;
                *p = 0.f;
            }
        }
        return;
    }
    const int T = s.taps, pitch = T | 1;
    const int ncol = min(64, s.cin - tl.ci0), run = ncol * T;
    const bool sn = s.w_sn != nullptr;
    float sv = 0.f, inv_sigma = 1.f;
    if (sn) { sv = (float)*s.s; inv_sigma = 1.f / *s.sigma; }
    float* const dst = tl.which ? s.dst1 : s.dst0;
    const int c0 = (int)threadIdx.x / T, t0 = (int)threadIdx.x - c0 * T, dc = 256 / T, dt = 256 - dc * T;
    for (int k = 0; k < CO_PER && tl.co0 + k < s.cout; ++k) {
        const int co = tl.co0 + k;
        __syncthreads();
        load_tile<true>(s, tl.which, co, tl.ci0, lds, pitch);
        __syncthreads();
        float* d = dst + ((size_t)co * s.cin + tl.ci0) * T;
        const float su = sn ? sv * s.u[co] : 0.f;
        const float* vv = sn ? s.v + (size_t)tl.ci0 * T : nullptr;
        for (int e = threadIdx.x, c = c0, t = t0; e < run; e += 256) {
            float g = lds[c * pitch + t];
            if (sn) g = (g - su * vv[e]) * inv_sigma;
            d[e] += g;
            c += dc; t += dt;
            if (t >= T) { t -= T; ++c; }
        }
    }
}

}  // namespace

extern "C" int mg_grad_drain(const mg_grad_slot* table_dev, int32_t nslots, const int32_t* block_slot_dev, int32_t nblocks,
                             double* partial, void* stream)
{
    MG_CHECK_ARG(table_dev && block_slot_dev && nslots > 0 && nblocks > 0, "mg_grad_drain: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (partial) {                                   // spectral-normed slots present: their dot products first
        hipLaunchKernelGGL(grad_sn_dot_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, table_dev, block_slot_dev, partial);
        MG_CHECK_LAUNCH("mg_grad_drain(sn dot)");
        hipLaunchKernelGGL(grad_sn_finish_kernel, dim3((unsigned)nslots), dim3(256), 0, st, table_dev, (const double*)partial, nblocks, block_slot_dev);
        MG_CHECK_LAUNCH("mg_grad_drain(sn finish)");
    }
    hipLaunchKernelGGL(grad_drain_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, table_dev, block_slot_dev);
    MG_CHECK_LAUNCH("mg_grad_drain");
    return MG_OK;
}

extern "C" int64_t mg_grad_slot_blocks(int32_t cout, int32_t cin, int32_t ntens)
{
    if (cout <= 0 || cin <= 0 || ntens < 1 || ntens > 2) return -1;
    return (int64_t)ntens * ((cin + 63) / 64) * ((cout + CO_PER - 1) / CO_PER) + 1;      // + the bias block
}
