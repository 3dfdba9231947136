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

int g_mg_conv_dot = 1;             // mg_set_option(8, v): 0 = few-channel long-K convs stay on the tap-list kernel

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
