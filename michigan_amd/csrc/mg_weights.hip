// mg_weights.hip -- batched weight preparation: every per-layer launch of the weight path folded into a few
// network-wide ones.
//
//   mg_pack_weights         ONE launch that writes any number of GEMM images (mg_pack_weight's gather: permute + zero-pad
//                           + cast + gamma|beta row interleave, optionally divided by a spectral-norm sigma) and plain
//                           W / sigma copies; a job table in device memory says what goes where.  A training step issued
//                           ~166 single-image pack launches of ~10 us (1.65 ms, profiles/r02a_kernel_stats.csv).
//   mg_sn_power_iteration   torch.nn.utils.spectral_norm's power iteration (dim 0, one iteration) for ALL spectral-normed
//                           layers of a network at once: v <- normalize(W^T u), u <- normalize(W v), sigma = u . (W v)
//                           in four launches (two streaming passes over the weights + two tiny per-layer finishes)
//                           instead of 2 rocBLAS gemv + 2 normalize launches per layer (architecture.py:39-42,
//                           normalization.py:28-29: 18 layers in G, 6 in D, twice per training step each).
// HBM-bound (weights only): K1 and K3 read every spectral-normed weight once each; no MFMA.
#include "mg_common.h"

namespace {

constexpr int PACK_PER_BLOCK = 1024;      // destination elements per workgroup (4 per thread)
constexpr int K1_ROWS = 128;              // rows of W per W^T u partial block (32: the per-layer finish walked 32 partials per column, 53 us)
constexpr int K1_COLS = 1024;             // columns per block: four per thread, 16-byte loads
constexpr int K3_ROWS = 4;                // rows of W per W v block (one wave each)

__device__ __forceinline__ int row_to_co2(int r, int cout, bool two, int& which)
{
    if (!two) { which = 0; return r < cout ? r : -1; }
    const int b = r >> 6, rem = r & 63;
    which = rem >> 5;
    const int co = b * 32 + (rem & 31);
    return co < cout ? co : -1;
}

// Tile = 4 destination rows x 64 destination columns x all T taps.  Destination element (t, r, c) of a GEMM image comes from
//   mode 0: W[co(r)][ci = c][t]      mode 1: W[co(c)][ci = r][t]       (co() = identity, or the gamma|beta row interleave)
// The reference layout has t fastest, so a gather by destination index (what the per-image pack_kernel does) touches every
// 64-byte line of W once per tap, T blocks apart: with 438 MB of generator weights that is T re-reads from HBM (the first batched
// version measured 1.5 TB/s of useful bytes).  Here a workgroup reads its sources as contiguous runs (mode 0: 64*T floats per row,
// mode 1: 4*T floats per column), transposes through LDS (pitch T|1: conflict-free for T = 1, 9, 16, 49) and writes
// 128-byte destination rows: every byte of W is read once.
constexpr int PK_ROWS = 4, PK_COLS = 64, PK_MAXT = 49;
// LDS: the 64 columns of a tile go through in sub-tiles of SC columns, SC = 64 for T <= 16 and 16 for the 7x7 window, so that the
// buffer is 17 KiB instead of the 50 KiB a full 7x7 tile needs (one 7x7 layer in the table used to cap EVERY workgroup of the
// launch at 3 per CU; with 8 per CU the 109 M generator weights re-pack in well under half the time, profiles/r02_pack_ab.txt).
constexpr int PK_LDS_FLOATS = PK_ROWS * 64 * 17;                 // >= 4 x 64 x (16 | 1) and >= 4 x 16 x (49 | 1) = 3200

__global__ __launch_bounds__(256) void pack_weights_kernel(const mg_pack_job* __restrict__ jobs, const int32_t* __restrict__ block_job)
{
    __shared__ float tile[PK_LDS_FLOATS];
    const mg_pack_job& j = jobs[block_job[blockIdx.x]];
    const int rel = (int)((int64_t)blockIdx.x - j.first_block);
    const int tid = threadIdx.x;
    if (j.mode == 2) {                                           // fp32 copy of w0 / sigma, reference layout
        const int64_t total = (int64_t)j.cout * j.cin * j.taps, base = (int64_t)rel * PACK_PER_BLOCK;
        const float sg = j.sigma ? j.sigma[0] : 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * 256 + tid;
            if (i < total) reinterpret_cast<float*>(j.dst)[i] = j.sigma ? j.w0[i] / sg : j.w0[i];     // a true division, like torch
        }
        return;
    }
    const bool two = j.w1 != nullptr;
    const int T = j.taps, pitch = T | 1;
    const int SC = T <= 16 ? 64 : 16;                            // sub-tile width (columns)
    const int cblocks = (j.cols_p + PK_COLS - 1) / PK_COLS;
    const int r0 = (rel / cblocks) * PK_ROWS, cb0 = (rel % cblocks) * PK_COLS;
    const float sg = j.sigma ? j.sigma[0] : 1.f;
    // store mapping: 4 rows x (SC/2 column pairs) x (128/SC tap phases)
    const int pairs = SC >> 1, nphase = 64 / pairs;
    const int rr_s = tid >> 6, cp = ((tid & 63) % pairs) * 2, th = (tid & 63) / pairs;
    for (int sc = 0; sc < PK_COLS; sc += SC) {
        const int c0 = cb0 + sc;
        if (c0 >= j.cols_p) break;                               // uniform
        if (sc) __syncthreads();                                 // the previous sub-tile has been stored
        // ---- load: contiguous runs of the reference-layout source --------------------------------------------------------------
        if (j.mode == 0) {
            const int ncol = max(0, min(SC, j.cin - c0));        // source columns (ci) that exist; the rest of the tile is zero padding
            for (int rr = 0; rr < PK_ROWS; ++rr) {
                int which = 0;
                const int co = (r0 + rr < j.rows_p) ? row_to_co2(r0 + rr, j.cout, two, which) : -1;
                const float* src = (co >= 0 && ncol > 0) ? (which ? j.w1 : j.w0) + ((size_t)co * j.cin + c0) * T : nullptr;
                const int run = src ? ncol * T : 0;
                int c = tid / T, t = tid - c * T;
                const int dc = 256 / T, dt = 256 - dc * T;
                for (int e = tid; e < SC * T; e += 256) {
                    tile[(rr * SC + c) * pitch + t] = e < run ? src[e] : 0.f;
                    c += dc; t += dt;
                    if (t >= T) { t -= T; ++c; }
                }
            }
        } else {
            const int nrow = min(PK_ROWS, j.cin - r0);           // source rows (ci) that exist
            const int per = PK_ROWS * T;                         // one column's contiguous run: W[co][r0 .. r0+3][0 .. T)
            for (int e = tid; e < SC * per; e += 256) {
                const int cc = e / per, rem = e - cc * per;
                const int rr = rem / T, t = rem - rr * T;
                int which = 0;
                const int co = (c0 + cc < j.cols_p) ? row_to_co2(c0 + cc, j.cout, two, which) : -1;
                float v = 0.f;
                if (co >= 0 && rr < nrow) v = (which ? j.w1 : j.w0)[((size_t)co * j.cin + r0) * T + rem];
                tile[(rr * SC + cc) * pitch + t] = v;
            }
        }
        __syncthreads();
        // ---- store: rows of consecutive destination columns (2 per thread) -------------------------------------------------------
        const int r = r0 + rr_s, c = c0 + cp;
        if (r < j.rows_p && c < j.cols_p) {                      // cols_p is even (a multiple of 8)
            for (int t = th; t < T; t += nphase) {
                float v0 = tile[(rr_s * SC + cp) * pitch + t], v1 = tile[(rr_s * SC + cp + 1) * pitch + t];
                if (j.sigma) { v0 = v0 / sg; v1 = v1 / sg; }
                const size_t o = ((size_t)t * j.rows_p + r) * j.cols_p + c;
                if (j.dtype == MG_BF16) reinterpret_cast<uint32_t*>(j.dst)[o >> 1] = f2bf2(v0, v1);
                else { float2 pv; pv.x = v0; pv.y = v1; *reinterpret_cast<float2*>(reinterpret_cast<float*>(j.dst) + o) = pv; }
            }
        }
    }
}

// ---- spectral-norm power iteration ---------------------------------------------------------------------------------------
// K1: partial[chunk][c] = sum over the chunk's K1_ROWS rows of W[r][c] * u[r]   (block = 256 columns x one row chunk)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const mg_sn_layer* __restrict__ layers, const int32_t* __restrict__ block_layer)
{
    const mg_sn_layer& L = layers[block_layer[blockIdx.x]];
    const int rel = blockIdx.x - L.first_block_k1;
    const int cblocks = (L.cols + K1_COLS - 1) / K1_COLS;
    const int chunk = rel / cblocks, c = ((rel % cblocks) * 256 + threadIdx.x) * 4;
    if (c >= L.cols) return;
    const int r0 = chunk * K1_ROWS, r1 = min(r0 + K1_ROWS, L.rows);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if ((L.cols & 3) == 0 && (((uintptr_t)L.w | (uintptr_t)L.partial) & 15) == 0) {   // rows stay 16-byte aligned: one dwordx4 load per row (4x the bytes in flight)
        const float* __restrict__ w = L.w + (size_t)r0 * L.cols + c;
#pragma unroll 8
        for (int r = r0; r < r1; ++r, w += L.cols) {
            const float4 w4 = *reinterpret_cast<const float4*>(w);
            const float u = L.u[r];
            acc[0] += w4.x * u; acc[1] += w4.y * u; acc[2] += w4.z * u; acc[3] += w4.w * u;
        }
        float4 o; o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
        *reinterpret_cast<float4*>(L.partial + (size_t)chunk * L.cols + c) = o;
        return;
    }
    for (int r = r0; r < r1; ++r) {
        const float u = L.u[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c + j < L.cols) acc[j] += L.w[(size_t)r * L.cols + c + j] * u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (c + j < L.cols) L.partial[(size_t)chunk * L.cols + c + j] = acc[j];
}

__device__ __forceinline__ float block_sum1024(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

// K2: t1 = sum of the partials (fixed order), v = t1 / max(||t1||, eps)  (one 1024-thread block per layer)
__global__ __launch_bounds__(1024) void sn_finish_v_kernel(const mg_sn_layer* __restrict__ layers, float eps)
{
    __shared__ float red[16];
    const mg_sn_layer& L = layers[blockIdx.x];
    const int chunks = (L.rows + K1_ROWS - 1) / K1_ROWS;
    float ss = 0.f;
    for (int c = threadIdx.x; c < L.cols; c += 1024) {
        float t = 0.f;
        for (int k = 0; k < chunks; ++k) t += L.partial[(size_t)k * L.cols + c];
        L.t1[c] = t;
        ss += t * t;
    }
    ss = block_sum1024(ss, red);
    const float denom = fmaxf(sqrtf(ss), eps);
    for (int c = threadIdx.x; c < L.cols; c += 1024) {
        const float q = L.t1[c] / denom;
        L.v[c] = q;
        if (L.v_copy) L.v_copy[c] = q;
    }
}

// K3: t2[r] = W[r] . v   (one wave per row)
__global__ __launch_bounds__(256) void sn_wv_kernel(const mg_sn_layer* __restrict__ layers, const int32_t* __restrict__ block_layer)
{
    const mg_sn_layer& L = layers[block_layer[blockIdx.x]];
    const int r = (blockIdx.x - L.first_block_k3) * K3_ROWS + (threadIdx.x >> 6);
    if (r >= L.rows) return;
    const float* __restrict__ w = L.w + (size_t)r * L.cols;
    float acc = 0.f;
    if ((L.cols & 3) == 0 && (((uintptr_t)L.w | (uintptr_t)L.v) & 15) == 0) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int c = (threadIdx.x & 63) * 4; c < L.cols; c += 256) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + c), v4 = *reinterpret_cast<const float4*>(L.v + c);
            a4[0] += w4.x * v4.x; a4[1] += w4.y * v4.y; a4[2] += w4.z * v4.z; a4[3] += w4.w * v4.w;
        }
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
        for (int c = threadIdx.x & 63; c < L.cols; c += 64) acc += w[c] * L.v[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) L.t2[r] = acc;
}

// K4: power iteration: u = t2 / max(||t2||, eps), sigma = u . t2; without it: sigma = u . t2 with the stored u
__global__ __launch_bounds__(1024) void sn_finish_u_kernel(const mg_sn_layer* __restrict__ layers, float eps, int power_iteration)
{
    __shared__ float red[16];
    const mg_sn_layer& L = layers[blockIdx.x];
    float dot = 0.f;
    if (power_iteration) {
        float ss = 0.f;
        for (int r = threadIdx.x; r < L.rows; r += 1024) ss += L.t2[r] * L.t2[r];
        ss = block_sum1024(ss, red);
        const float denom = fmaxf(sqrtf(ss), eps);
        for (int r = threadIdx.x; r < L.rows; r += 1024) {
            const float q = L.t2[r] / denom;
            L.u[r] = q;
            if (L.u_copy) L.u_copy[r] = q;
            dot += q * L.t2[r];
        }
    } else {
        for (int r = threadIdx.x; r < L.rows; r += 1024) dot += L.u[r] * L.t2[r];
    }
    dot = block_sum1024(dot, red);
    if (threadIdx.x == 0) L.sigma[0] = dot;
}

}  // namespace

extern "C" int64_t mg_pack_job_blocks(int32_t mode, int32_t cout, int32_t cin, int32_t taps, int32_t rows_p, int32_t cols_p)
{
    if (mode == 2) return cout > 0 && cin > 0 && taps > 0 ? ((int64_t)cout * cin * taps + PACK_PER_BLOCK - 1) / PACK_PER_BLOCK : -1;
    if ((mode != 0 && mode != 1) || rows_p <= 0 || cols_p <= 0 || (cols_p & 1) || taps <= 0 || taps > PK_MAXT) return -1;
    return (int64_t)((rows_p + PK_ROWS - 1) / PK_ROWS) * ((cols_p + PK_COLS - 1) / PK_COLS);
}

extern "C" int mg_pack_weights(const mg_pack_job* jobs_dev, int32_t njobs, const int32_t* block_job_dev, int32_t nblocks, void* stream)
{
    MG_CHECK_ARG(jobs_dev && block_job_dev && njobs > 0 && nblocks > 0, "mg_pack_weights: bad arguments");
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev, block_job_dev);
    MG_CHECK_LAUNCH("mg_pack_weights");
    return MG_OK;
}

extern "C" int64_t mg_sn_layer_blocks(int32_t rows, int32_t cols, int32_t which)
{
    if (rows <= 0 || cols <= 0) return -1;
    if (which == 0) return (int64_t)((cols + K1_COLS - 1) / K1_COLS) * ((rows + K1_ROWS - 1) / K1_ROWS);      // K1 workgroups
    if (which == 1) return (rows + K3_ROWS - 1) / K3_ROWS;                                         // K3 workgroups
    if (which == 2) return (rows + K1_ROWS - 1) / K1_ROWS;                                         // row chunks = rows of `partial`
    return -1;
}

extern "C" int mg_sn_power_iteration(const mg_sn_layer* layers_dev, int32_t nlayers, const int32_t* block_layer_k1, int32_t nblocks_k1,
                                     const int32_t* block_layer_k3, int32_t nblocks_k3, int32_t do_power_iteration, float eps, void* stream)
{
    MG_CHECK_ARG(layers_dev && nlayers > 0 && block_layer_k3 && nblocks_k3 > 0, "mg_sn_power_iteration: bad arguments");
    MG_CHECK_ARG(!do_power_iteration || (block_layer_k1 && nblocks_k1 > 0), "mg_sn_power_iteration: missing K1 table");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (do_power_iteration) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3((unsigned)nblocks_k1), dim3(256), 0, st, layers_dev, block_layer_k1);
        MG_CHECK_LAUNCH("mg_sn_power_iteration(W^T u)");
        hipLaunchKernelGGL(sn_finish_v_kernel, dim3((unsigned)nlayers), dim3(1024), 0, st, layers_dev, eps);
        MG_CHECK_LAUNCH("mg_sn_power_iteration(v)");
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((unsigned)nblocks_k3), dim3(256), 0, st, layers_dev, block_layer_k3);
    MG_CHECK_LAUNCH("mg_sn_power_iteration(W v)");
    hipLaunchKernelGGL(sn_finish_u_kernel, dim3((unsigned)nlayers), dim3(1024), 0, st, layers_dev, eps, do_power_iteration);
    MG_CHECK_LAUNCH("mg_sn_power_iteration(u, sigma)");
    return MG_OK;
}
