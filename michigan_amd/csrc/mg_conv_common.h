// mg_conv_common.h -- pieces shared by the conv translation units (generic tap-list kernels in
// mg_conv.hip, 3x3 halo-tile kernel in mg_conv_halo.hip).  Everything here is static / inline: the
// library is built without relocatable device code, so each TU carries its own copy.
#pragma once
#include "mg_common.h"
#include <type_traits>

namespace {

constexpr int ROWB = 64;    // bytes of K per LDS row and pipeline stage
constexpr int NTHR = 256;

}  // namespace
struct ConvK {              // kernel-side view of mg_conv_desc (passed by value)
    const void* in; const void* wt; void* out;
    const float* bias; const void* resid; const void* x;
    const float* mean; const float* rstd; void* gamma_out;
    int N, Hin, Win, Cin;
    int Hout, Wout, Cout, Cout_gemm, CoutP;
    int Hj, Wj, isy, isx, osy, osx, ooy, oox;
    int ntaps, act; float slope;
    int ngemm;              // N*Hj*Wj
    int tiles_m;
    int tpc;                // taps packed into one 64-byte K chunk (tiny Cin), 1 otherwise
    int tap[MG_MAX_TAPS];   // (dy & 0xffff) | (dx << 16)
    int tiles_y, tiles_x;   // halo kernel: spatial tiles per image
};
namespace {

// LDS image shared by both pipelines: rows of 64 bytes of K, NO padding; the four 16-byte pieces of a
// row are XOR-swizzled with (row >> 2) & 3.  ds_write_b128 (8-lane groups = 2 rows x 4 pieces, bank =
// addr/4 mod 32) and ds_read_b128 (16-lane groups with rows distinct mod 16, bank = addr/4 mod 64) are
// both conflict-free with it; the previous 80-byte padded rows made every ds_write_b128 2-way
// (SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_conv.txt).
__device__ __forceinline__ int lds_off(int row, int piece) { return row * ROWB + ((piece ^ ((row >> 2) & 3)) << 4); }

template <typename T, int MT, int NT>
__device__ __forceinline__ void conv_compute(const unsigned char* As, const unsigned char* Bs, int l31, int hi,
                                             f32x16_t (&acc)[MT][NT])
{
    // As/Bs point at this wave's first row; the lane's rows are l31 + 32*t, so (row >> 2) & 3 == (l31 >> 2) & 3
    const int sw = (l31 >> 2) & 3;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[MT], b[NT];
            const int po = ((ks * 2 + hi) ^ sw) << 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const bf16x8_t*>(As + (mt * 32 + l31) * ROWB + po);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + (nt * 32 + l31) * ROWB + po);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    } else {
        // lane (row, hi) owns K elements hi*8 .. hi*8+7 of the 16-float chunk; MFMA j consumes element j
        // of both halves -- any K permutation is legal as long as A and B use the same one.
        f32x4_t a[MT][2], b[NT][2];
        const int p0 = ((hi * 2) ^ sw) << 4, p1 = ((hi * 2 + 1) ^ sw) << 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            a[mt][0] = *reinterpret_cast<const f32x4_t*>(As + (mt * 32 + l31) * ROWB + p0);
            a[mt][1] = *reinterpret_cast<const f32x4_t*>(As + (mt * 32 + l31) * ROWB + p1);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            b[nt][0] = *reinterpret_cast<const f32x4_t*>(Bs + (nt * 32 + l31) * ROWB + p0);
            b[nt][1] = *reinterpret_cast<const f32x4_t*>(Bs + (nt * 32 + l31) * ROWB + p1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j >> 2][j & 3], b[nt][j >> 2][j & 3],
                                                                       acc[mt][nt], 0, 0, 0);
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// `pixmap(p, opix)`: p = pixel index inside the workgroup's pixel tile (0 .. TN-1) -> false if the pixel does
// not exist, else opix = flat output pixel index ((n*Hout + oy)*Wout + ox).
template <typename T, int MT, int NT, int EPI, typename PixMap>
__device__ __forceinline__ void conv_epilogue(const ConvK& d, f32x16_t (&acc)[MT][NT], int m0, PixMap&& pixmap,
                                              int wm, int wn, int l31, int hi)
{
    // ---- epilogue -----------------------------------------------------------
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
    // compile-time tile indices (static_for): runtime-indexed accumulator arrays would be demoted to scratch
    static_for<0, NT>([&](auto nt_) {
        constexpr int nt = decltype(nt_)::value;
        size_t opix;
        if (!pixmap(wn * NT * 32 + nt * 32 + l31, opix)) return;

        if constexpr (EPI == MG_EPI_PLAIN) {
            const T* __restrict__ Res = reinterpret_cast<const T*>(d.resid);
            static_for<0, MT * 4>([&](auto mr_) {
                {
                    constexpr int mt = decltype(mr_)::value / 4, rq = decltype(mr_)::value % 4;
                    const int co = m0 + wm * MT * 32 + mt * 32 + rq * 8 + hi * 4;
                    if (co >= d.Cout) return;
                    f32x4_t v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][rq * 4 + j];
                    if (d.bias) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (co + j < d.Cout_gemm) ? d.bias[co + j] : 0.f;
                    }
                    const size_t o = opix * d.Cout + co;
                    if ((d.Cout & 3) == 0) {
                        if (Res) { f32x4_t rv = ET<T>::load4(Res + o);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += rv[j]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = mg_act(v[j], d.act, d.slope);
                        ET<T>::store4(Out + o, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (co + j < d.Cout) {
                                float s = v[j];
                                if (Res) s += ET<T>::load1(Res + o + j);
                                ET<T>::store1(Out + o + j, mg_act(s, d.act, d.slope));
                            }
                        }
                    }
                }
            });
        } else {
            // SPADE: acc[0] = gamma rows, acc[1] = beta rows of the same 32 output channels.
            const T* __restrict__ X = reinterpret_cast<const T*>(d.x);
            T* __restrict__ G1 = reinterpret_cast<T*>(d.gamma_out);
            const int grow = m0 + wm * 64;              // first GEMM row of this wave's [gamma|beta] block
            static_for<0, 4>([&](auto rq_) {
                constexpr int rq = decltype(rq_)::value;
                const int sub = rq * 8 + hi * 4;
                const int oc = (grow >> 1) + sub;
                if (oc >= d.Cout) return;
                f32x4_t g, bt;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[j]  = 1.f + acc[0][nt][rq * 4 + j] + (d.bias ? d.bias[grow + sub + j] : 0.f);
                    bt[j] = acc[1][nt][rq * 4 + j] + (d.bias ? d.bias[grow + 32 + sub + j] : 0.f);
                }
                const size_t o = opix * d.Cout + oc;
                if ((d.Cout & 3) == 0) {
                    const f32x4_t xv = ET<T>::load4(X + o);
                    f32x4_t hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = (xv[j] - d.mean[oc + j]) * d.rstd[oc + j];
                        hv[j] = mg_act(xh * g[j] + bt[j], d.act, d.slope);
                    }
                    ET<T>::store4(Out + o, hv);
                    if (G1) ET<T>::store4(G1 + o, g);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (oc + j < d.Cout) {
                            const float xh = (ET<T>::load1(X + o + j) - d.mean[oc + j]) * d.rstd[oc + j];
                            ET<T>::store1(Out + o + j, mg_act(xh * g[j] + bt[j], d.act, d.slope));
                            if (G1) ET<T>::store1(G1 + o + j, g[j]);
                        }
                    }
                }
            });
        }
    });
}

// zero source for out-of-image taps / tail rows: long enough to be walked chunk by chunk (<= 8 KiB of K per tap: Cin <= 4096 bf16 / 2048 f32)
static __device__ __attribute__((aligned(64))) unsigned char g_mg_zeros[8192 + 64];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


// linear pixel tile of the tap-list kernels: q = q0 + p over N*Hj*Wj, strided/offset output grid
struct LinearPixMap {
    const ConvK& d; int q0, HWj;
    __device__ __forceinline__ bool operator()(int p, size_t& opix) const {
        const int q = q0 + p;
        if (q >= d.ngemm) return false;
        const int n = q / HWj, r = q - n * HWj;
        const int jy = r / d.Wj, jx = r - jy * d.Wj;
        opix = (size_t)((n * d.Hout + jy * d.osy + d.ooy) * d.Wout + jx * d.osx + d.oox);
        return true;
    }
};

}  // namespace

int launch_conv_halo(ConvK& k, int dtype, int epilogue, hipStream_t st);   // mg_conv_halo.hip
