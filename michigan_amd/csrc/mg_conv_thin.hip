// mg_conv_thin.hip -- convolutions over an 8-channel bf16 input, forward and weight gradient.  First the 3x3 / stride-1 / same-size
// case (register-resident weights); further down the kernels for ANY window of at most 7x7 taps at stride 1 or 2 (conv_thin_taps_kernel,
// wgrad_thin_taps_kernel: weights in LDS, per-workgroup gradient slabs).  The 3x3 case: the mlp_shared convs
// of every SPADE layer on the (mask, orientation) map (reference normalization.py:94-97,111) and the first
// convs of the encoders.  K = 9 taps x 8 channels = 72, so these layers are bound by WRITING their output
// (128 channels per pixel against 8 read): the generic tap-list kernel spent its time in the per-tile ring
// prologue and in 8-byte scattered stores and reached ~2.2 TB/s of output.
//
// Layout of the work, per 256-thread workgroup (persistent over 8-row x 32-pixel output tiles):
//   * the weights live in registers for the whole launch: one tap's 8 input channels are exactly the 8
//     consecutive K values a lane owns in v_mfma_f32_32x32x16_bf16, so K step s = taps (2s, 2s+1) and the A
//     fragment is one 16-byte load of the packed image [tap][co][8]; tap 9 is a zero fragment;
//   * the (8+2) x (32+2) pixel halo of the input is staged in LDS once per tile (16 bytes per pixel); the B
//     fragment of a K step is one ds_read_b128 at the lane's pixel shifted by its tap;
//   * a wave owns 64 output channels of two (Cout <= 64) or four rows; its 32 pixels x 64 channels go through a
//     private, XOR-swizzled LDS image so that the global stores are whole 16-byte chunks, 8 consecutive lanes
//     (one 128-byte line) per pixel.
// Results are the tap-list kernel's (same products, fp32 accumulation, bias, activation, one rounding to bf16).
#include "mg_conv_common.h"
#include "mg_wgrad_common.h"

int g_mg_conv_thin = 2;            // mg_set_option(6, v): 0 = 8-channel convs stay on the tap-list kernel, 1 = only the 3x3 / stride-1 ones leave it

namespace {

constexpr int THIN_TH = 8, THIN_TW = 32;
constexpr int THIN_HW = THIN_TW + 2, THIN_HH = THIN_TH + 2;
constexpr int THIN_HALO_BYTES = THIN_HH * THIN_HW * 16;          // 5440

template <int CS>       // 64-channel halves of GEMM rows: Cout_gemm <= 64 * CS; each wave owns ONE half
__global__ __launch_bounds__(256, 4) void conv3x3_thin_kernel(const ConvK d, const int ntiles)
{
    constexpr int NB = 2;                                        // 32-channel MFMA row blocks per wave
    constexpr int CH = NB * 4;                                   // 16-byte chunks (8 channels) a wave stages per pixel
    constexpr int ROW = NB * 64;                                 // staged bytes per pixel
    constexpr int RPW = 2 * CS;                                  // output rows per wave (4 waves cover 8 rows x CS halves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;
    float* const bias_s = reinterpret_cast<float*>(smem + THIN_HALO_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int half = CS == 2 ? (wave >> 1) : 0, r0 = CS == 2 ? (wave & 1) * RPW : wave * RPW;
    const int cb = half * 64;
    unsigned char* const stage = smem + THIN_HALO_BYTES + CS * 64 * 4 + wave * (32 * ROW);

    if (tid < CS * 64) bias_s[tid] = (d.bias && tid < d.Cout_gemm) ? d.bias[tid] : 0.f;

    // A fragments (weights) and the lane's tap offsets inside the halo image
    bf16x8_t wa[5][NB];
    int boff[5];
    const uint16_t* __restrict__ Wt = reinterpret_cast<const uint16_t*>(d.wt);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = 2 * s + hi;
        const int tc = t < 9 ? t : 8;
        const int dy = (int)(short)(d.tap[tc] & 0xffff), dx = d.tap[tc] >> 16;
        boff[s] = ((1 + dy) * THIN_HW + (1 + dx) + l31) * 16;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t < 9) z = *reinterpret_cast<const bf16x8_t*>(Wt + ((size_t)t * d.CoutP + cb + nb * 32 + l31) * 8);
            wa[s][nb] = z;
        }
    }

    const float neg = d.act == MG_ACT_NONE ? 1.f : (d.act == MG_ACT_RELU ? 0.f : d.slope);
    const bool relu = d.act == MG_ACT_RELU;
    const uint16_t* __restrict__ In = reinterpret_cast<const uint16_t*>(d.in);
    unsigned char* __restrict__ Out = reinterpret_cast<unsigned char*>(d.out);
    const int tpi = d.tiles_y * d.tiles_x;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / d.tiles_x) * THIN_TH, x0 = (tr % d.tiles_x) * THIN_TW;
        __syncthreads();                                         // previous tile's halo reads are done (and bias_s is visible)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = tid + it * 256;
            if (i < THIN_HH * THIN_HW) {
                const int hy = i / THIN_HW, hx = i - hy * THIN_HW;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                uint4 v = {0u, 0u, 0u, 0u};
                if (gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win)
                    v = *reinterpret_cast<const uint4*>(In + ((size_t)(img * d.Hin + gy) * d.Win + gx) * 8);
                *reinterpret_cast<uint4*>(halo + i * 16) = v;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int mb = 0; mb < RPW; ++mb) {
            const int r = r0 + mb, y = y0 + r;
            if (y >= d.Hin) break;                               // wave-uniform
            bf16x8_t b[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) b[s] = *reinterpret_cast<const bf16x8_t*>(halo + boff[s] + r * (THIN_HW * 16));
            f32x16_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t bq = *reinterpret_cast<const f32x4_t*>(bias_s + cb + nb * 32 + g * 8 + hi * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nb][g * 4 + e] = bq[e];
                }
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s][nb], b[s], acc[nb], 0, 0, 0);
            // lane (pixel l31, half-wave hi) holds channels cb + nb*32 + g*8 + hi*4 + e: 8 bytes of chunk nb*4+g
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = f2bf2(mg_act_fast(acc[nb][g * 4 + 0], neg, relu), mg_act_fast(acc[nb][g * 4 + 1], neg, relu));
                    u.y = f2bf2(mg_act_fast(acc[nb][g * 4 + 2], neg, relu), mg_act_fast(acc[nb][g * 4 + 3], neg, relu));
                    const int c = nb * 4 + g;
                    *reinterpret_cast<uint2*>(stage + l31 * ROW + ((c ^ ((l31 >> 1) & (CH - 1))) << 4) + hi * 8) = u;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const size_t orow = (size_t)(img * d.Hout + y) * d.Wout + x0;
#pragma unroll
            for (int it = 0; it < CH / 2; ++it) {
                const int q = it * 64 + lane;
                const int p = q / CH, c = q % CH;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + p * ROW + ((c ^ ((p >> 1) & (CH - 1))) << 4));
                const int co = cb + c * 8;
                if (co < d.Cout)
                    *reinterpret_cast<uint4*>(Out + ((orow + p) * d.Cout + co) * 2) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                     // stage is rewritten by the next row
        }
    }
}

template <int CS>
int launch_thin(ConvK& k, hipStream_t st)
{
    k.tiles_y = (k.Hin + THIN_TH - 1) / THIN_TH;
    k.tiles_x = k.Win / THIN_TW;
    const long ntiles = (long)k.N * k.tiles_y * k.tiles_x;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_taps(thin): bad grid %ld", ntiles);
    const int lds = THIN_HALO_BYTES + CS * 64 * 4 + 4 * 32 * 128;
    const long grid = ntiles < 1024 ? ntiles : 1024;             // 4 workgroups per CU, each walking tiles
    hipLaunchKernelGGL(conv3x3_thin_kernel<CS>, dim3((unsigned)grid), dim3(256), lds, st, k, (int)ntiles);
    MG_CHECK_LAUNCH("mg_conv_taps(thin)");
    return MG_OK;
}


// ---- any tap window over an 8-channel input (7x7 first conv of the background encoder, the discriminator's 4x4 / stride-2 first conv,
// the appearance encoder's 3x3 / stride-2 first partial conv) ------------------------------------------------------------------------
// Same GEMM view as above -- one tap's 8 channels are the 8 K values a lane owns -- but the window no longer fits the register file
// (49 taps x 64 channels = 50 KiB), so the packed weights sit in LDS for the whole launch ([tap][co][8], a lane's A fragment is one
// conflict-free ds_read_b128) next to the tile's input halo ((8-1)*S + KH rows x (32-1)*S + KW pixels, 16 bytes each); the B fragment of
// K step s is one ds_read_b128 at the lane's pixel * S plus the byte offset of tap 2s + hi (a 64-entry table in LDS, filled from the
// kernel arguments).  The tap-list kernel gathered each pixel's K row as four separate 16-byte taps into its staging image: 230-400
// TFLOP/s and 1.1 TB/s of output on these layers (profiles/r03_conv_census.txt).  Output widths need not be multiples of the tile.
template <int CS, int S>
__global__ __launch_bounds__(256, 2) void conv_thin_taps_kernel(const ConvK d, const int ntiles, const int HH, const int HW,
                                                                const int dy0, const int dx0, const int nsteps)
{
    constexpr int NB = 2, CH = NB * 4, ROW = NB * 64, RPW = 2 * CS, RB = 2, WROWS = CS * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int half = CS == 2 ? (wave >> 1) : 0, r0 = CS == 2 ? (wave & 1) * RPW : wave * RPW;
    const int cb = half * 64;
    unsigned char* const wl = smem;
    unsigned char* const halo = wl + nsteps * 2 * WROWS * 16;
    int* const tofs = reinterpret_cast<int*>(halo + HH * HW * 16);
    float* const bias_s = reinterpret_cast<float*>(tofs + MG_MAX_TAPS);
    unsigned char* const stage = reinterpret_cast<unsigned char*>(bias_s + WROWS) + wave * (32 * ROW);

    if (tid < WROWS) bias_s[tid] = (d.bias && tid < d.Cout_gemm) ? d.bias[tid] : 0.f;
    if (tid < MG_MAX_TAPS) {                                     // d.tap[] holds byte offsets into the halo here (launch_thin_taps)
        int v = 0;
#pragma unroll
        for (int t = 0; t < MG_MAX_TAPS; ++t) v = (tid == t) ? d.tap[t] : v;
        tofs[tid] = v;
    }
    {
        const uint16_t* __restrict__ Wt = reinterpret_cast<const uint16_t*>(d.wt);
        for (int i = tid; i < nsteps * 2 * WROWS; i += 256) {
            const int t = i / WROWS, row = i % WROWS;
            uint4 v = {0u, 0u, 0u, 0u};
            if (t < d.ntaps) v = *reinterpret_cast<const uint4*>(Wt + ((size_t)t * d.CoutP + row) * 8);
            *reinterpret_cast<uint4*>(wl + i * 16) = v;
        }
    }
    const float neg = d.act == MG_ACT_NONE ? 1.f : (d.act == MG_ACT_RELU ? 0.f : d.slope);
    const bool relu = d.act == MG_ACT_RELU;
    const uint16_t* __restrict__ In = reinterpret_cast<const uint16_t*>(d.in);
    unsigned char* __restrict__ Out = reinterpret_cast<unsigned char*>(d.out);
    const int tpi = d.tiles_y * d.tiles_x;
    const unsigned char* const wa = wl + (hi * WROWS + cb + l31) * 16;       // this lane's A row of tap `hi`; +2 taps per K step

    // the next tile's halo is fetched into registers behind this tile's MFMAs and stores
    constexpr int HPT = 6;                                       // halo pieces per thread: HH * HW <= 20 * 69 = 1380
    uint4 hv[HPT];
    const int npix = HH * HW;
    auto fetch = [&](int tile) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / d.tiles_x) * THIN_TH, x0 = (tr % d.tiles_x) * THIN_TW;
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            const int hy = i / HW, hx = i - hy * HW;
            const int gy = y0 * S + dy0 + hy, gx = x0 * S + dx0 + hx;
            uint4 v = {0u, 0u, 0u, 0u};
            if (i < npix && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win)
                v = *reinterpret_cast<const uint4*>(In + ((size_t)(img * d.Hin + gy) * d.Win + gx) * 8);
            hv[q] = v;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / d.tiles_x) * THIN_TH, x0 = (tr % d.tiles_x) * THIN_TW;
        __syncthreads();                                         // previous tile's halo reads are done (first tile: weights, table, bias visible)
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            if (i < npix) *reinterpret_cast<uint4*>(halo + i * 16) = hv[q];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        // two rows of the wave advance together: one pair of A reads feeds 4 MFMAs (a row at a time the weight reads alone were two thirds
        // of the LDS traffic, ~190 of the 256 B/clk the LDS delivers with two workgroups per CU)
#pragma unroll 1
        for (int rg = 0; rg < RPW; rg += RB) {
        if (y0 + r0 + rg >= d.Hj) break;                         // wave-uniform
        const unsigned char* const bb = halo + (((r0 + rg) * S) * HW + l31 * S) * 16;
        f32x16_t acc[RB][NB];
#pragma unroll
        for (int mb = 0; mb < RB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t bq = *reinterpret_cast<const f32x4_t*>(bias_s + cb + nb * 32 + g * 8 + hi * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mb][nb][g * 4 + e] = bq[e];
                }
#pragma unroll 5
        for (int s = 0; s < nsteps; ++s) {
            const int o = tofs[2 * s + hi];
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(wa + s * (2 * WROWS * 16));
            const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(wa + s * (2 * WROWS * 16) + 32 * 16);
#pragma unroll
            for (int mb = 0; mb < RB; ++mb) {                   // rows past the image read halo rows that exist (zeros or real pixels); never stored
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(bb + o + mb * (S * HW * 16));
                acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[mb][0], 0, 0, 0);
                acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[mb][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mb = 0; mb < RB; ++mb) {
            const int y = y0 + r0 + rg + mb;
            if (y >= d.Hj) break;                                // wave-uniform
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = f2bf2(mg_act_fast(acc[mb][nb][g * 4 + 0], neg, relu), mg_act_fast(acc[mb][nb][g * 4 + 1], neg, relu));
                    u.y = f2bf2(mg_act_fast(acc[mb][nb][g * 4 + 2], neg, relu), mg_act_fast(acc[mb][nb][g * 4 + 3], neg, relu));
                    const int c = nb * 4 + g;
                    *reinterpret_cast<uint2*>(stage + l31 * ROW + ((c ^ ((l31 >> 1) & (CH - 1))) << 4) + hi * 8) = u;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const size_t orow = (size_t)(img * d.Hout + y) * d.Wout + x0;
            const int pmax = d.Wout - x0;                        // ragged last tile of a row
#pragma unroll
            for (int it = 0; it < CH / 2; ++it) {
                const int q = it * 64 + lane;
                const int p = q / CH, c = q % CH;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + p * ROW + ((c ^ ((p >> 1) & (CH - 1))) << 4));
                const int co = cb + c * 8;
                if (co < d.Cout && p < pmax)
                    *reinterpret_cast<uint4*>(Out + ((orow + p) * d.Cout + co) * 2) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                     // stage is rewritten by the next row
        }
        }
    }
}

struct ThinTapsGeom { int S, dy0, dx0, KH, KW, HH, HW, nsteps, CS, lds; long ntiles; };

bool thin_taps_geom(const ConvK& k, ThinTapsGeom& g)
{
    if (k.isy != k.isx || (k.isy != 1 && k.isy != 2) || k.ntaps > 50) return false;
    int dy0 = 127, dx0 = 127, dy1 = -128, dx1 = -128;
    for (int t = 0; t < k.ntaps; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        dy0 = dy < dy0 ? dy : dy0; dy1 = dy > dy1 ? dy : dy1; dx0 = dx < dx0 ? dx : dx0; dx1 = dx > dx1 ? dx : dx1;
    }
    g.S = k.isy; g.dy0 = dy0; g.dx0 = dx0; g.KH = dy1 - dy0 + 1; g.KW = dx1 - dx0 + 1;
    if (g.KH > 7 || g.KW > 7) return false;
    g.HH = (THIN_TH - 1) * g.S + g.KH; g.HW = (THIN_TW - 1) * g.S + g.KW;
    g.nsteps = (k.ntaps + 1) / 2;
    g.CS = k.Cout_gemm <= 64 ? 1 : 2;
    g.lds = g.nsteps * 2 * g.CS * 64 * 16 + g.HH * g.HW * 16 + MG_MAX_TAPS * 4 + g.CS * 64 * 4 + 4 * 32 * 128;
    g.ntiles = (long)k.N * ((k.Hj + THIN_TH - 1) / THIN_TH) * ((k.Wj + THIN_TW - 1) / THIN_TW);
    return g.lds <= 80 * 1024 && g.ntiles >= 64 && g.ntiles <= 0x7fffffffL;
}

template <int CS, int S>
int launch_thin_taps(ConvK& k, const ThinTapsGeom& g, hipStream_t st)
{
    k.tiles_y = (k.Hj + THIN_TH - 1) / THIN_TH;
    k.tiles_x = (k.Wj + THIN_TW - 1) / THIN_TW;
    int ofs[MG_MAX_TAPS];
    for (int t = 0; t < MG_MAX_TAPS; ++t) {
        const int tc = t < k.ntaps ? t : 0;                      // the odd tap of the last K step meets zero weights: any valid address
        const int dy = (int)(short)(k.tap[tc] & 0xffff), dx = k.tap[tc] >> 16;
        ofs[t] = ((dy - g.dy0) * g.HW + (dx - g.dx0)) * 16;
    }
    for (int t = 0; t < MG_MAX_TAPS; ++t) k.tap[t] = ofs[t];
    auto kern = conv_thin_taps_kernel<CS, S>;
    mg_raise_lds_cap(reinterpret_cast<const void*>(kern), 80 * 1024);
    const long grid = g.ntiles < 512 ? g.ntiles : 512;           // 2 workgroups per CU, each walking tiles
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), g.lds, st, k, (int)g.ntiles, g.HH, g.HW, g.dy0, g.dx0, g.nsteps);
    MG_CHECK_LAUNCH("mg_conv_taps(thin, tap list)");
    return MG_OK;
}

// ---- weight gradient of the same layers ---------------------------------------------------------------------
// dW[tap][co][ci] = sum over pixels of dY[p][co] * X[p + tap][ci] (+ dbias[co] = sum dY[p][co]): a GEMM with
// M = co, N = (tap, ci) = 72 columns (three 32-column MFMA blocks) and K = pixels, bound by reading dY once.
// A workgroup walks 4-row x 32-pixel tiles: the dY tile (pixel-major, 64-byte blocks XOR-swizzled by row) and the
// 6 x 34 halo of X are staged in LDS; both fragments are ds_read_b64_tr_b16 transposes -- for B the lane's address
// is its tap-shifted halo pixel, so no im2col image exists anywhere.  A wave owns one 32-channel block of dY
// (two waves share a block and split the K steps when Cg == 64) and keeps its 32 x 96 fp32 accumulators in
// registers across all its tiles; one pass of fp32 atomics per workgroup at the end (dW is 36 KiB).
typedef __attribute__((ext_vector_type(4))) short thin_s16x4_t;
typedef __attribute__((address_space(3))) thin_s16x4_t* thin_lds_s16x4_p;
constexpr int WTH = 4;                                           // rows per weight-gradient tile
constexpr int WHALO_BYTES = (WTH + 2) * THIN_HW * 16;            // 3264

template <int MB>       // 32-channel blocks of dY: Cg == 32 * MB, MB = 2 or 4
__global__ __launch_bounds__(256, MB == 4 ? 3 : 4) void wgrad3x3_thin_kernel(const Wg3K d, const int ntiles, const int tiles_y, const int tiles_x)
{
    constexpr int RBA = MB * 64, PPA = RBA / 16;                 // bytes / 16-byte pieces per dY pixel
    constexpr int KG = 4 / MB;                                   // waves sharing a channel block
    constexpr int APT = WTH * 32 * PPA / 256;                    // dY pieces per thread and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;
    unsigned char* const dyt = smem + WHALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mblk = wave % MB, kg = wave / MB;
    auto swz = [](int row) { return RBA == 256 ? (row & 3) : ((row >> 1) & 1); };    // in 64-byte blocks

    // transpose-read geometry (as in mg_wgrad3x3.hip): 16-lane group g2 -> columns (g2 & 1) * 16, K half g2 >> 1;
    // lane i16 -> K row i16 >> 2, columns (i16 & 3) * 4
    const int i16 = lane & 15, g2 = lane >> 4;
    const int rsub = (g2 >> 1) * 8 + (i16 >> 2);
    const int csub = ((g2 & 1) * 16 + (i16 & 3) * 4) * 2;
    int boff[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + (g2 & 1) * 16 + (i16 & 3) * 4;   // first of the lane's 4 columns: tap n / 8, channels n % 8 ..
        const int tap = (n >> 3) < 9 ? (n >> 3) : 8;             // columns 72..95 feed accumulator columns nobody stores
        boff[nb] = ((tap / 3) * THIN_HW + rsub + tap % 3) * 16 + ((n >> 2) & 1) * 8;
    }

    f32x16_t acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    float bsum = 0.f;

    const uint16_t* __restrict__ X = reinterpret_cast<const uint16_t*>(d.x);
    const uint16_t* __restrict__ DY = reinterpret_cast<const uint16_t*>(d.dy);
    const int tpi = tiles_y * tiles_x;

    // The next tile's global loads are issued BEFORE this tile's MFMAs and land in registers behind them (the loop used to be load ->
    // barrier -> compute with nothing in flight during the compute: 3.3 TB/s of dY at four workgroups per CU).
    uint4 hv = {0u, 0u, 0u, 0u};
    uint4 av[APT];
    auto fetch = [&](int tile) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / tiles_x) * WTH, x0 = (tr % tiles_x) * THIN_TW;
        if (tid < (WTH + 2) * THIN_HW) {
            const int hy = tid / THIN_HW, hx = tid - hy * THIN_HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            uint4 v = {0u, 0u, 0u, 0u};
            if (gy >= 0 && gy < d.H && gx >= 0 && gx < d.W)
                v = *reinterpret_cast<const uint4*>(X + ((size_t)(img * d.H + gy) * d.W + gx) * 8);
            hv = v;
        }
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            const int g = j * 256 + tid, prow = g / PPA, slot = g % PPA;
            const int y = y0 + (prow >> 5), x = x0 + (prow & 31);
            uint4 v = {0u, 0u, 0u, 0u};
            if (y < d.H) v = *reinterpret_cast<const uint4*>(DY + ((size_t)(img * d.H + y) * d.W + x) * d.Cg + slot * 8);
            av[j] = v;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        if (tid < (WTH + 2) * THIN_HW) *reinterpret_cast<uint4*>(halo + tid * 16) = hv;
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            const int g = j * 256 + tid, prow = g / PPA, slot = g % PPA;
            *reinterpret_cast<uint4*>(dyt + prow * RBA + ((slot ^ (swz(prow) << 2)) << 4)) = av[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int kk = 0; kk < WTH * 2 / KG; ++kk) {
            const int kstep = kk * KG + kg, r = kstep >> 1, ks = kstep & 1;
            const int pr = r * 32 + ks * 16 + rsub;
            const unsigned char* pa = dyt + pr * RBA + ((mblk * 64 + csub) ^ (swz(pr) << 6));
            const thin_s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pa));
            const thin_s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pa + 4 * RBA));
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x4_t w = __builtin_bit_cast(u32x4_t, a);
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum += __uint_as_float(w[j] << 16) + __uint_as_float(w[j] & 0xffff0000u);
            }
            const unsigned char* pb = halo + (r * THIN_HW + ks * 16) * 16;
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const thin_s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pb + boff[nb]));
                const thin_s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pb + boff[nb] + 64));
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
    }

    const int split = blockIdx.x * KG + kg;                      // deterministic mode: one slab per (workgroup, K group)
    if (d.dbias) {
        const float t = bsum + __shfl_xor(bsum, 32);             // the two K halves of the row
        if (hi == 0) wg_accum(d.dbias, d.det_stride, split, (size_t)(mblk * 32 + l31), t);
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + l31, tap = n >> 3, ci = n & 7;
        if (tap < 9) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                wg_accum(d.dw, d.det_stride, split, (size_t)(tap * d.Cg + co) * 8 + ci, acc[nb][r]);
            }
        }
    }
}


// ---- weight gradient over any tap window (the layers conv_thin_taps_kernel runs forward) ------------------------------------------------
// dW[tap][co][ci] = sum over output pixels of dY[p][co] * X[p * S + tap][ci]: M = 64 channels of dY (two 32-row blocks), N = (tap, ci) in
// 32-column blocks of four taps (13 blocks for a 7x7 window), K = pixels.  The four waves are 2 row blocks x 2 column groups of NBW
// blocks; a workgroup walks 4-row x 32-pixel tiles of dY with the matching X halo in LDS ((4-1)*S + KH rows x (32-1)*S + KW pixels) and
// keeps its accumulators in registers; both fragments are ds_read_b64_tr_b16 transposes as in wgrad3x3_thin_kernel (for B the lane's
// K row is its pixel * S shifted by the column's tap).  dW is 100 KiB for the 7x7 layer, so a pass of atomics per workgroup would cost
// more than the GEMM: every workgroup stores its partial dW into its own slab of a scratch buffer and a finishing launch adds the slabs
// in workgroup order (bit-reproducible whether or not the caller asked for deterministic gradients).
template <int NBW, int S>
__global__ __launch_bounds__(256, 2) void wgrad_thin_taps_kernel(const WgT d, const int ntiles, const int tiles_y, const int tiles_x)
{
    constexpr int RBA = 128, PPA = 8;                            // dY bytes / 16-byte pieces per pixel (Cg == 64)
    constexpr int APT = WTH * 32 * PPA / 256;                    // dY pieces per thread and tile (4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hbytes = d.HH * d.HW * 16;
    unsigned char* const halo = smem;
    unsigned char* const dyt = smem + hbytes;
    int* const tofs = reinterpret_cast<int*>(dyt + WTH * 32 * RBA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mblk = wave & 1, ng = wave >> 1;
    auto swz = [](int row) { return (row >> 1) & 1; };            // in 64-byte blocks

    if (tid < MG_MAX_TAPS) {                                     // d.tap[] = pixel offsets of the taps inside the halo
        int v = 0;
#pragma unroll
        for (int t = 0; t < MG_MAX_TAPS; ++t) v = (tid == t) ? d.tap[t] : v;
        tofs[tid] = v;
    }
    __syncthreads();
    const int i16 = lane & 15, g2 = lane >> 4;
    const int rsub = (g2 >> 1) * 8 + (i16 >> 2);
    const int csub = ((g2 & 1) * 16 + (i16 & 3) * 4) * 2;
    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int n = (ng * NBW + nb) * 32 + (g2 & 1) * 16 + (i16 & 3) * 4;   // first of the lane's 4 columns: tap n / 8, channels n % 8 ..
        const int tap = (n >> 3) < d.ntaps ? (n >> 3) : 0;       // columns past the window feed accumulator columns nobody stores
        boff[nb] = (tofs[tap] + rsub * S) * 16 + ((n >> 2) & 1) * 8;
    }

    f32x16_t acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    float bsum = 0.f;

    const uint16_t* __restrict__ X = reinterpret_cast<const uint16_t*>(d.x);
    const uint16_t* __restrict__ DY = reinterpret_cast<const uint16_t*>(d.dy);
    const int tpi = tiles_y * tiles_x;

    // the next tile's global loads are issued before this tile's MFMAs (see wgrad3x3_thin_kernel)
    constexpr int HPT = 3;                                       // halo pieces per thread: HH * HW <= 10 * 69
    uint4 hv[HPT];
    uint4 av[APT];
    const int npix = d.HH * d.HW;
    auto fetch = [&](int tile) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / tiles_x) * WTH, x0 = (tr % tiles_x) * THIN_TW;
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            const int hy = i / d.HW, hx = i - hy * d.HW;
            const int gy = y0 * S + d.dy0 + hy, gx = x0 * S + d.dx0 + hx;
            uint4 v = {0u, 0u, 0u, 0u};
            if (i < npix && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win)
                v = *reinterpret_cast<const uint4*>(X + ((size_t)(img * d.Hin + gy) * d.Win + gx) * 8);
            hv[q] = v;
        }
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            const int g = j * 256 + tid, prow = g / PPA, slot = g % PPA;
            const int y = y0 + (prow >> 5), x = x0 + (prow & 31);
            uint4 v = {0u, 0u, 0u, 0u};
            if (y < d.Hj && x < d.Wj) v = *reinterpret_cast<const uint4*>(DY + ((size_t)(img * d.Hj + y) * d.Wj + x) * 64 + slot * 8);
            av[j] = v;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            if (i < npix) *reinterpret_cast<uint4*>(halo + i * 16) = hv[q];
        }
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            const int g = j * 256 + tid, prow = g / PPA, slot = g % PPA;
            *reinterpret_cast<uint4*>(dyt + prow * RBA + ((slot ^ (swz(prow) << 2)) << 4)) = av[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 2
        for (int kstep = 0; kstep < WTH * 2; ++kstep) {
            const int r = kstep >> 1, ks = kstep & 1;
            const int pr = r * 32 + ks * 16 + rsub;
            const unsigned char* pa = dyt + pr * RBA + ((mblk * 64 + csub) ^ (swz(pr) << 6));
            const thin_s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pa));
            const thin_s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pa + 4 * RBA));
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            if (ng == 0) {                                       // wave-uniform: one column group sums the bias gradient
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x4_t w = __builtin_bit_cast(u32x4_t, a);
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum += __uint_as_float(w[j] << 16) + __uint_as_float(w[j] & 0xffff0000u);
            }
            const unsigned char* pb = halo + ((r * S) * d.HW + ks * 16 * S) * 16;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const thin_s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pb + boff[nb]));
                const thin_s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_p)(pb + boff[nb] + 64 * S));
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
    }

    float* const slab = d.ws + (size_t)blockIdx.x * d.slab;      // every element of the slab is written by exactly one lane
    if (d.has_bias && ng == 0) {
        const float t = bsum + __shfl_xor(bsum, 32);             // the two K halves of the row
        if (hi == 0) slab[(size_t)d.ntaps * 64 * 8 + mblk * 32 + l31] = t;
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int n = (ng * NBW + nb) * 32 + l31, tap = n >> 3, ci = n & 7;
        if (tap < d.ntaps) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                slab[(size_t)(tap * 64 + co) * 8 + ci] = acc[nb][r];
            }
        }
    }
}

template <int NBW, int S>
int launch_wthin_taps(const WgT& k, int grid, long ntiles, int tiles_y, int tiles_x, hipStream_t st)
{
    const int lds = k.HH * k.HW * 16 + WTH * 32 * 128 + MG_MAX_TAPS * 4;
    hipLaunchKernelGGL((wgrad_thin_taps_kernel<NBW, S>), dim3((unsigned)grid), dim3(256), lds, st, k, (int)ntiles, tiles_y, tiles_x);
    MG_CHECK_LAUNCH("mg_conv_wgrad(thin, tap list)");
    return MG_OK;
}

template <int MB>
int launch_wthin(Wg3K& k, hipStream_t st, int* nsplit, bool dry)
{
    const int tiles_y = (k.H + WTH - 1) / WTH, tiles_x = k.W / THIN_TW;
    const long ntiles = (long)k.N * tiles_y * tiles_x;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_wgrad(thin): bad grid %ld", ntiles);
    // every workgroup ends with one pass of atomics over the whole dW: at least 4 tiles each, at most 2 per CU
    long grid = (ntiles + 3) / 4;
    if (grid > 512) grid = 512;
    if (nsplit) *nsplit = (int)grid * (4 / MB);
    if (dry) return MG_OK;
    const int lds = WHALO_BYTES + WTH * 32 * MB * 64;
    hipLaunchKernelGGL(wgrad3x3_thin_kernel<MB>, dim3((unsigned)grid), dim3(256), lds, st, k, (int)ntiles, tiles_y, tiles_x);
    MG_CHECK_LAUNCH("mg_conv_wgrad(thin)");
    return MG_OK;
}

}  // namespace

// bf16, Cin == 8, the nine 3x3 taps at stride 1 onto a same-size output, plain epilogue without residual / mask / tanh
bool conv_thin_applies(const ConvK& k, int dtype, int epilogue)
{
    if (!g_mg_conv_thin || dtype != MG_BF16 || epilogue != MG_EPI_PLAIN) return false;
    if (k.Cin != 8 || k.ntaps != 9 || k.isy != 1 || k.isx != 1 || k.osy != 1 || k.osx != 1 || k.ooy != 0 || k.oox != 0) return false;
    if (k.Hj != k.Hin || k.Wj != k.Win || k.Hout != k.Hin || k.Wout != k.Win) return false;
    if (k.resid || k.x || k.act == MG_ACT_TANH) return false;
    if ((k.Win % THIN_TW) || (k.Cout & 7) || k.Cout_gemm > 128 || k.Cout_gemm < 32) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        seen |= 1u << ((dy + 1) * 3 + dx + 1);
    }
    if (seen != 0x1ffu) return false;
    return (long)k.N * ((k.Hin + THIN_TH - 1) / THIN_TH) * (k.Win / THIN_TW) >= 64;
}

// bf16, Cin == 8, any window of at most 7 x 7 taps at stride 1 or 2 onto the whole output grid (osy = osx = 1), plain epilogue
bool conv_thin_taps_applies(const ConvK& k, int dtype, int epilogue)
{
    if (g_mg_conv_thin < 2 || dtype != MG_BF16 || epilogue != MG_EPI_PLAIN) return false;
    if (k.Cin != 8 || k.osy != 1 || k.osx != 1 || k.ooy != 0 || k.oox != 0 || k.Hout != k.Hj || k.Wout != k.Wj) return false;
    if (k.resid || k.x || k.act == MG_ACT_TANH || (k.Cout & 7) || k.Cout_gemm > 128 || k.Cout_gemm < 32) return false;
    ThinTapsGeom g;
    return thin_taps_geom(k, g);
}

int launch_conv_thin_taps(ConvK& k, hipStream_t st)
{
    ThinTapsGeom g;
    if (!thin_taps_geom(k, g)) return mg_fail(MG_ERR_UNSUPPORTED, "mg_conv_taps(thin, tap list): geometry");
    if (g.CS == 1) return g.S == 1 ? launch_thin_taps<1, 1>(k, g, st) : launch_thin_taps<1, 2>(k, g, st);
    return g.S == 1 ? launch_thin_taps<2, 1>(k, g, st) : launch_thin_taps<2, 2>(k, g, st);
}

int launch_conv_thin(ConvK& k, hipStream_t st)
{
    if (k.Cout_gemm <= 64) return launch_thin<1>(k, st);
    return launch_thin<2>(k, st);
}

// Eligibility beyond these checks is decided by the caller (mg_wgrad.hip): bf16, the 9 taps of a 3x3 / pad 1 window
// in raster order, stride 1, same-size dY.
bool wgrad_thin_applies(const Wg3K& k)
{
    return g_mg_conv_thin && k.Cin == 8 && (k.Cg == 64 || k.Cg == 128) && (k.W % THIN_TW) == 0 &&
           (long)k.N * ((k.H + WTH - 1) / WTH) * (k.W / THIN_TW) >= 64;
}

int launch_wgrad_thin(Wg3K& k, hipStream_t st, int* nsplit, bool dry)
{
    return k.Cg == 64 ? launch_wthin<2>(k, st, nsplit, dry) : launch_wthin<4>(k, st, nsplit, dry);
}

// bf16, Cin == 8, Cg == 64, any window of at most 7 x 7 taps at stride 1 or 2 (not the 3x3 / stride-1 / same-size case of the kernel above)
static bool wgrad_thin_taps_geom(const mg_wgrad_desc* d, WgT& k, long& ntiles, int& tiles_y, int& tiles_x, int& nblocks)
{
    if (g_mg_conv_thin < 2 || d->dtype != MG_BF16 || d->Cin != 8 || d->Cg != 64 || d->isy != d->isx || (d->isy != 1 && d->isy != 2)) return false;
    if (d->ntaps > 52) return false;
    int dy0 = 127, dx0 = 127, dy1 = -128, dx1 = -128;
    for (int t = 0; t < d->ntaps; ++t) {
        const int dy = d->tap_dy[t], dx = d->tap_dx[t];
        dy0 = dy < dy0 ? dy : dy0; dy1 = dy > dy1 ? dy : dy1; dx0 = dx < dx0 ? dx : dx0; dx1 = dx > dx1 ? dx : dx1;
    }
    const int KH = dy1 - dy0 + 1, KW = dx1 - dx0 + 1, S = d->isy;
    if (KH > 7 || KW > 7) return false;
    k.x = d->x; k.dy = d->dy; k.ws = nullptr; k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Hj = d->Hj; k.Wj = d->Wj; k.ntaps = d->ntaps;
    k.dy0 = dy0; k.dx0 = dx0; k.HH = (WTH - 1) * S + KH; k.HW = (THIN_TW - 1) * S + KW;
    k.slab = (long)d->ntaps * 64 * 8 + (d->dbias ? 64 : 0); k.has_bias = d->dbias ? 1 : 0;
    for (int t = 0; t < MG_MAX_TAPS; ++t) {
        const int tc = t < d->ntaps ? t : 0;
        k.tap[t] = (d->tap_dy[tc] - dy0) * k.HW + (d->tap_dx[tc] - dx0);
    }
    tiles_y = (d->Hj + WTH - 1) / WTH; tiles_x = (d->Wj + THIN_TW - 1) / THIN_TW;
    ntiles = (long)d->N * tiles_y * tiles_x;
    nblocks = (d->ntaps * 8 + 31) / 32;
    return ntiles >= 64 && ntiles <= 0x7fffffffL;
}

bool wgrad_thin_taps_applies(const mg_wgrad_desc* d)
{
    WgT k; long ntiles; int ty, tx, nb;
    return wgrad_thin_taps_geom(d, k, ntiles, ty, tx, nb);
}

// dw / dbias / det_stride as in route_wgrad: det_stride != 0 -> dw IS the caller's slab workspace (dbias behind each slab's dW) and the
// caller runs the finishing pass; otherwise the slabs go to the stream's scratch buffer and the finishing pass runs here.
int launch_wgrad_thin_taps(const mg_wgrad_desc* d, hipStream_t st, float* dw, float* dbias, long det_stride, int* nsplit, bool dry)
{
    WgT k; long ntiles; int tiles_y, tiles_x, nblocks;
    if (!wgrad_thin_taps_geom(d, k, ntiles, tiles_y, tiles_x, nblocks)) return mg_fail(MG_ERR_UNSUPPORTED, "mg_conv_wgrad(thin, tap list): geometry");
    long grid = (ntiles + 7) / 8;                                // at least 8 tiles per slab
    if (grid > 512) grid = 512;
    if (nsplit) *nsplit = (int)grid;
    if (dry) return MG_OK;
    const long ndw = (long)d->ntaps * 64 * 8;
    if (det_stride) {
        if (det_stride != k.slab) return mg_fail(MG_ERR_ARG, "mg_conv_wgrad(thin, tap list): slab stride %ld, expected %ld", det_stride, k.slab);
        k.ws = dw;
    } else {
        k.ws = mg_stream_scratch(st, (size_t)grid * k.slab * sizeof(float));
        if (k.ws == nullptr) return mg_fail(MG_ERR_LAUNCH, "mg_conv_wgrad(thin, tap list): scratch allocation failed");
    }
    const int S = d->isy;
    int rc;
    if (nblocks <= 4) rc = S == 1 ? launch_wthin_taps<2, 1>(k, (int)grid, ntiles, tiles_y, tiles_x, st) : launch_wthin_taps<2, 2>(k, (int)grid, ntiles, tiles_y, tiles_x, st);
    else if (nblocks <= 8) rc = S == 1 ? launch_wthin_taps<4, 1>(k, (int)grid, ntiles, tiles_y, tiles_x, st) : launch_wthin_taps<4, 2>(k, (int)grid, ntiles, tiles_y, tiles_x, st);
    else rc = S == 1 ? launch_wthin_taps<7, 1>(k, (int)grid, ntiles, tiles_y, tiles_x, st) : launch_wthin_taps<7, 2>(k, (int)grid, ntiles, tiles_y, tiles_x, st);
    if (rc != MG_OK || det_stride) return rc;
    return launch_wgrad_det_finish(k.ws, (int)grid, k.slab, dw, ndw, dbias, d->dbias ? 64 : 0, st);
}
