// mg_conv_dot.hip -- convolutions with very few output channels and a long reduction: the 1-channel 4x4 heads of the
// two patch discriminators (reference discriminator.py:96: nn.Conv2d(nf, 1, kernel_size=4, stride=1, padding=2) on
// 512 channels, K = 16 taps x 512 = 8192).  On the tap-list MFMA kernel the single output channel occupies a 32-row
// tile (97 % padding: 2-6 TFLOP/s, 0.15-0.21 ms per launch, four launches per step) although the layer only has to
// read its input once (19-73 MB).  Here a wavefront owns an output pixel: its 64 lanes split the input channels in
// 16-byte pieces (8 bf16 / 4 fp32), the weights sit in LDS for the whole launch, every tap is one coalesced row read,
// products accumulate in fp32 per lane and a DPP/shuffle tree reduces the wave.  Bound by the L2 -> CU re-read of the
// overlapping windows (16 taps), not by MFMA -- there is nothing to feed a matrix core with.
// Results: same products and fp32 accumulation as the tap-list kernel in a different order, bias, activation, one
// rounding to the storage type.
#include "mg_conv_common.h"

int g_mg_conv_dot = 2;             // mg_set_option(8, v): 0 = few-output-channel convs stay on the tap-list kernel, 1 = only the wave-per-pixel dot kernel

namespace {

constexpr int DOT_MAXCO = 4;

template <typename T>
__global__ __launch_bounds__(256) void conv_dot_kernel(const ConvK d)
{
    constexpr int EPL = 16 / (int)sizeof(T);                   // elements per lane piece (8 bf16 / 4 fp32)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* const wsm = reinterpret_cast<T*>(smem);                  // [tap][co][Cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Cin = d.Cin, nco = d.Cout;
    {
        const T* __restrict__ Wt = reinterpret_cast<const T*>(d.wt);          // packed image [tap][CoutP][Cin]
        const int pieces = Cin / EPL, total = d.ntaps * nco * pieces;
        for (int i = tid; i < total; i += blockDim.x) {
            const int pc = i % pieces, co = (i / pieces) % nco, t = i / (pieces * nco);
            *reinterpret_cast<uint4*>(wsm + ((size_t)(t * nco + co) * Cin + pc * EPL)) =
                *reinterpret_cast<const uint4*>(Wt + ((size_t)(t * d.CoutP + co) * Cin + pc * EPL));
        }
    }
    __syncthreads();
    const T* __restrict__ In = reinterpret_cast<const T*>(d.in);
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
    const int HWj = d.Hj * d.Wj;
    for (int q = blockIdx.x * 4 + wave; q < d.ngemm; q += gridDim.x * 4) {
        const int n = q / HWj, r = q - n * HWj, jy = r / d.Wj, jx = r - jy * d.Wj;
        float acc[DOT_MAXCO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int t = 0; t < d.ntaps; ++t) {
            const int tp = d.tap[t];
            const int iy = jy * d.isy + (int)(short)(tp & 0xffff), ix = jx * d.isx + (tp >> 16);
            // taps outside the image read a valid pixel and are weighted by zero: no branch, so the unrolled loop keeps
            // several row reads in flight per wave
            const bool ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            const float m = ok ? 1.f : 0.f;
            const T* __restrict__ src = In + ((size_t)(n * d.Hin + (ok ? iy : 0)) * d.Win + (ok ? ix : 0)) * Cin;
            for (int c = lane * EPL; c < Cin; c += 64 * EPL) {
                float xv[8], wv[8];
                if constexpr (sizeof(T) == 2) {
                    const uint4 u = *reinterpret_cast<const uint4*>(src + c);
                    const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[2 * j] = __uint_as_float(uu[j] << 16); xv[2 * j + 1] = __uint_as_float(uu[j] & 0xffff0000u); }
                } else {
                    const f32x4_t u = *reinterpret_cast<const f32x4_t*>(src + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[j] = u[j];
                }
#pragma unroll
                for (int co = 0; co < DOT_MAXCO; ++co) {
                    if (co >= nco) break;
                    const T* wp = wsm + (size_t)(t * nco + co) * Cin + c;
                    if constexpr (sizeof(T) == 2) {
                        const uint4 u = *reinterpret_cast<const uint4*>(wp);
                        const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) { wv[2 * j] = __uint_as_float(uu[j] << 16); wv[2 * j + 1] = __uint_as_float(uu[j] & 0xffff0000u); }
                    } else {
                        const f32x4_t u = *reinterpret_cast<const f32x4_t*>(wp);
#pragma unroll
                        for (int j = 0; j < 4; ++j) wv[j] = u[j];
                    }
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < EPL; ++j) s = fmaf(xv[j], wv[j], s);
                    acc[co] = fmaf(m, s, acc[co]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < DOT_MAXCO; ++co) {
            float v = acc[co];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[co] = v;
        }
        if (lane == 0) {
            const size_t o = ((size_t)(n * d.Hout + jy * d.osy + d.ooy) * d.Wout + jx * d.osx + d.oox) * d.Cout;
            const T* __restrict__ Res = reinterpret_cast<const T*>(d.resid);
            for (int co = 0; co < nco; ++co) {
                float v = acc[co] + (d.bias ? d.bias[co] : 0.f);
                if (Res) v += ET<T>::load1(Res + o + co);
                ET<T>::store1(Out + o + co, mg_act(v, d.act, d.slope));
            }
        }
    }
}


// ---- few output channels over a 64-channel bf16 input (round 3) ---------------------------------------------------------------------------
// conv_img (generator.py:227: 64 -> 3, 3x3, tanh), the data gradients that end in an 8-channel network input (VGG conv1_1 onto the fake image,
// the four stride-2 parity classes of the discriminators' first conv): a 32-row MFMA tile is 75-90 % padding for them and the tap-list kernel
// re-fetched the input once per tap (1.3 TB/s of input, 34 TFLOP/s, profiles/r03_conv_census.txt).  Here the GEMM rows are the 16 rows of
// v_mfma_f32_16x16x32_bf16, a workgroup stages the input halo of an 8 x 32 tile of the output grid once (128 bytes per pixel, the eight 16-byte
// pieces XOR-swizzled with the pixel index so that 16 consecutive pixels of one K piece cover all 64 banks), the (at most 9 taps x 2 K chunks)
// weight fragments live in registers, and a K step is one ds_read_b128 + one MFMA per 16 pixels.  Bound by reading the input once.
typedef __attribute__((ext_vector_type(4))) float few_f32x4_t;
constexpr int FEW_TH = 8, FEW_TW = 32, FEW_MAXT = 9;

__global__ __launch_bounds__(256, 2) void conv_fewout_kernel(const ConvK d, const int ntiles, const int HH, const int HW, const int dy0, const int dx0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;                   // MFMA row / column index, K group (8 channels)
    const uint16_t* __restrict__ Wt = reinterpret_cast<const uint16_t*>(d.wt);
    const uint16_t* __restrict__ In = reinterpret_cast<const uint16_t*>(d.in);
    uint16_t* __restrict__ Out = reinterpret_cast<uint16_t*>(d.out);

    bf16x8_t wa[FEW_MAXT][2];
    int toff[FEW_MAXT];
#pragma unroll
    for (int t = 0; t < FEW_MAXT; ++t) {
        const int tc = t < d.ntaps ? t : 0;
        const int dy = (int)(short)(d.tap[tc] & 0xffff), dx = d.tap[tc] >> 16;
        toff[t] = (dy - dy0) * HW + (dx - dx0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t < d.ntaps) z = *reinterpret_cast<const bf16x8_t*>(Wt + ((size_t)t * d.CoutP + l15) * 64 + c * 32 + kg * 8);
            wa[t][c] = z;
        }
    }
    few_f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[i] = (kg * 4 + i) < d.Cout_gemm ? d.bias[kg * 4 + i] : 0.f;
    }
    const int tpi = d.tiles_y * d.tiles_x;
    const int npieces = HH * HW * 8;

    // halo pieces of a tile: all loads of a thread in flight at once, and the NEXT tile's loads issued before this tile's MFMAs
    constexpr int HPT = (10 * 34 * 8 + 255) / 256;               // 11 pieces per thread at most
    uint4 hv[HPT];
    auto fetch = [&](int tile) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / d.tiles_x) * FEW_TH, x0 = (tr % d.tiles_x) * FEW_TW;
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            const int px = i >> 3, piece = i & 7;
            const int hy = px / HW, hx = px - hy * HW;
            const int gy = y0 + dy0 + hy, gx = x0 + dx0 + hx;
            uint4 v = {0u, 0u, 0u, 0u};
            if (i < npieces && gy >= 0 && gy < d.Hin && gx >= 0 && gx < d.Win)
                v = *reinterpret_cast<const uint4*>(In + ((size_t)(img * d.Hin + gy) * d.Win + gx) * 64 + piece * 8);
            hv[q] = v;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / tpi, tr = tile - img * tpi;
        const int y0 = (tr / d.tiles_x) * FEW_TH, x0 = (tr % d.tiles_x) * FEW_TW;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < HPT; ++q) {
            const int i = tid + q * 256;
            const int px = i >> 3, piece = i & 7;
            if (i < npieces) *reinterpret_cast<uint4*>(smem + px * 128 + ((piece ^ (px & 7)) << 4)) = hv[q];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int r = wave * 2 + mb, jy = y0 + r;
            if (jy >= d.Hj) break;                               // wave-uniform
            few_f32x4_t acc[2] = {bias4, bias4};
#pragma unroll
            for (int t = 0; t < FEW_MAXT; ++t) {
                if (t < d.ntaps) {                               // uniform
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const int j = r * HW + cb * 16 + l15 + toff[t];
                        const unsigned char* row = smem + j * 128;
                        const int sw = j & 7;
                        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(row + ((kg ^ sw) << 4));
                        const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(row + (((4 + kg) ^ sw) << 4));
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[t][0], b0, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[t][1], b1, acc[cb], 0, 0, 0);
                    }
                }
            }
            // lane (column l15 = pixel, K group kg) holds output channels kg*4 .. kg*4+3 of its pixel
            const int oy = jy * d.osy + d.ooy;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int jx = x0 + cb * 16 + l15;
                if (jx < d.Wj && kg * 4 < d.Cout) {
                    const size_t o = ((size_t)(img * d.Hout + oy) * d.Wout + (jx * d.osx + d.oox)) * d.Cout + kg * 4;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = mg_act(acc[cb][i], d.act, d.slope);
                    if ((d.Cout & 3) == 0) {
                        uint2 u; u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(Out + o) = u;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (kg * 4 + i < d.Cout) Out[o + i] = f2bf(v[i]);
                    }
                }
            }
        }
    }
}

struct FewGeom { int dy0, dx0, HH, HW, lds; long ntiles; };

bool fewout_geom(const ConvK& k, FewGeom& g)
{
    int dy0 = 127, dx0 = 127, dy1 = -128, dx1 = -128;
    for (int t = 0; t < k.ntaps; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        dy0 = dy < dy0 ? dy : dy0; dy1 = dy > dy1 ? dy : dy1; dx0 = dx < dx0 ? dx : dx0; dx1 = dx > dx1 ? dx : dx1;
    }
    if (dy1 - dy0 > 2 || dx1 - dx0 > 2) return false;
    g.dy0 = dy0; g.dx0 = dx0; g.HH = FEW_TH + dy1 - dy0; g.HW = FEW_TW + dx1 - dx0;
    g.lds = g.HH * g.HW * 128;
    g.ntiles = (long)k.N * ((k.Hj + FEW_TH - 1) / FEW_TH) * ((k.Wj + FEW_TW - 1) / FEW_TW);
    return g.ntiles >= 64 && g.ntiles <= 0x7fffffffL;
}

}  // namespace

bool conv_dot_applies(const ConvK& k, int dtype, int epilogue)
{
    if (!g_mg_conv_dot || epilogue != MG_EPI_PLAIN) return false;
    const int esz = dtype == MG_BF16 ? 2 : 4;
    if (k.Cout < 1 || k.Cout > DOT_MAXCO || k.Cout_gemm > 32) return false;
    if (k.Cin < 128 || (k.Cin * esz) % 16 != 0) return false;                 // short reductions stay on the packed-taps MFMA path
    if (k.x != nullptr) return false;                                          // no ReLU-mask epilogue here
    return (size_t)k.ntaps * k.Cout * k.Cin * esz <= 60 * 1024;                // weights resident in LDS
}

int launch_conv_dot(ConvK& k, int dtype, hipStream_t st)
{
    const int esz = dtype == MG_BF16 ? 2 : 4;
    const size_t lds = (size_t)k.ntaps * k.Cout * k.Cin * esz;
    long nblk = ((long)k.ngemm + 3) / 4;
    if (nblk > 256 * 8) nblk = 256 * 8;                                        // persistent: 8 workgroups per CU at most
    if (dtype == MG_BF16) hipLaunchKernelGGL(conv_dot_kernel<uint16_t>, dim3((unsigned)nblk), dim3(256), lds, st, k);
    else hipLaunchKernelGGL(conv_dot_kernel<float>, dim3((unsigned)nblk), dim3(256), lds, st, k);
    MG_CHECK_LAUNCH("mg_conv_taps(dot)");
    return MG_OK;
}

// bf16, 64 input channels, at most 16 GEMM rows, at most 9 taps inside a 3x3 window at input stride 1, any output stride / offset
bool conv_fewout_applies(const ConvK& k, int dtype, int epilogue)
{
    if (g_mg_conv_dot < 2 || dtype != MG_BF16 || epilogue != MG_EPI_PLAIN) return false;
    if (k.Cin != 64 || k.Cout_gemm > 16 || k.ntaps > FEW_MAXT || k.isy != 1 || k.isx != 1) return false;
    if (k.resid || k.x) return false;
    FewGeom g;
    return fewout_geom(k, g);
}

int launch_conv_fewout(ConvK& k, hipStream_t st)
{
    FewGeom g;
    if (!fewout_geom(k, g)) return mg_fail(MG_ERR_UNSUPPORTED, "mg_conv_taps(few outputs): geometry");
    k.tiles_y = (k.Hj + FEW_TH - 1) / FEW_TH;
    k.tiles_x = (k.Wj + FEW_TW - 1) / FEW_TW;
    const long grid = g.ntiles < 512 ? g.ntiles : 512;
    hipLaunchKernelGGL(conv_fewout_kernel, dim3((unsigned)grid), dim3(256), g.lds, st, k, (int)g.ntiles, g.HH, g.HW, g.dy0, g.dx0);
    MG_CHECK_LAUNCH("mg_conv_taps(few outputs)");
    return MG_OK;
}
