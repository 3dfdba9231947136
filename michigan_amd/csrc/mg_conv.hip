// mg_conv.hip -- tap-list implicit-GEMM convolution on the CDNA4 matrix cores.
//
// GEMM view (per launch):   D[co][q] = sum_{tap, ci} W[tap][co][ci] * X[pix(q)+tap][ci]
//   rows   (MFMA "i")  = output channels (A operand = packed weights)
//   cols   (MFMA "n")  = output pixels   (B operand = NHWC activations, gathered per tap)
//   K                  = ntaps * Cin, walked tap-major in 64-byte chunks
// so every lane ends up holding, for ONE pixel, quads of 4 consecutive output
// channels -> NHWC vector stores and channel-wise epilogues (bias, SPADE
// modulation with per-channel mean/rstd) without any cross-lane traffic.
//
// Both operands are staged global -> registers -> LDS (16 B pieces, rows padded
// 64 -> 80 B so the ds_read_b128 fragment reads are bank-conflict free),
// double-buffered with one barrier per K chunk; the next chunk's global loads
// are issued before the current chunk's MFMAs (latency hides under the matrix
// pipe).  bf16 uses v_mfma_f32_32x32x16_bf16, f32 uses v_mfma_f32_32x32x2_f32
// (exact fp32 fma chain) -- same tiling, same LDS image (64 B of K per row).
#include "mg_common.h"

namespace {

constexpr int ROWB = 64;    // bytes of K per LDS row and pipeline stage
constexpr int ROWS = 80;    // padded LDS row stride in bytes
constexpr int NTHR = 256;

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
    int tap[MG_MAX_TAPS];   // (dy & 0xffff) | (dx << 16)
};

template <typename T, int WM, int WN, int MT, int NT, int EPI>
__global__ __launch_bounds__(NTHR) void conv_taps_kernel(const ConvK d)
{
    constexpr bool BF = (sizeof(T) == 2);
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    constexpr int EPP = 16 / (int)sizeof(T);        // elements per 16-byte piece
    constexpr int CH  = ROWB / (int)sizeof(T);      // K elements per chunk
    constexpr int A_PT = (TM * 4 + NTHR - 1) / NTHR;
    constexpr int B_PT = (TN * 4) / NTHR;
    constexpr int STAGE = (TM + TN) * ROWS;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(EPI == MG_EPI_PLAIN || MT == 2, "SPADE epilogue needs gamma/beta tile pair");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware (bijective) block -> tile map: blocks of one XCD (b % 8) get a
    // contiguous tile range so neighbouring pixel tiles share that XCD's L2.
    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int tm = tile % d.tiles_m, tn = tile / d.tiles_m;
    const int m0 = tm * TM;
    const int q0 = tn * TN;

    const T* __restrict__ In = reinterpret_cast<const T*>(d.in);
    const T* __restrict__ Wt = reinterpret_cast<const T*>(d.wt);
    const int HWj = d.Hj * d.Wj;

    // ---- per-thread staging assignment (fixed over the K loop) ------------
    const int piece = tid & 3;
    const int srow  = tid >> 2;                       // 0..63
    int b_n[B_PT], b_y[B_PT], b_x[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
        const int q = q0 + srow + i * 64;
        if (q < d.ngemm) {
            const int n = q / HWj, r = q - n * HWj;
            const int jy = r / d.Wj, jx = r - jy * d.Wj;
            b_n[i] = n; b_y[i] = jy * d.isy; b_x[i] = jx * d.isx;
        } else { b_n[i] = 0; b_y[i] = -(1 << 20); b_x[i] = 0; }
    }

    uint4 ra[A_PT], rb[B_PT];
    const int nchunk = (d.Cin + CH - 1) / CH;
    const int nk = d.ntaps * nchunk;

    auto gload = [&](int tap, int chunk) {
        const int c = chunk * CH + piece * EPP;
        const bool cv = c < d.Cin;
        const int tp = d.tap[tap];
        const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int row = srow + i * 64;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < TM && cv)
                v = *reinterpret_cast<const uint4*>(Wt + ((size_t)(tap * d.CoutP + m0 + row) * d.Cin + c));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int iy = b_y[i] + dy, ix = b_x[i] + dx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (cv && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win)
                v = *reinterpret_cast<const uint4*>(In + ((size_t)((b_n[i] * d.Hin + iy) * d.Win + ix) * d.Cin + c));
            rb[i] = v;
        }
    };
    auto lstore = [&](int s) {
        unsigned char* base = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int row = srow + i * 64;
            if (row < TM) *reinterpret_cast<uint4*>(base + row * ROWS + piece * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int row = srow + i * 64;
            *reinterpret_cast<uint4*>(base + (TM + row) * ROWS + piece * 16) = rb[i];
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    auto compute = [&](int s) {
        const unsigned char* As = smem + s * STAGE + (wm * MT * 32 + l31) * ROWS;
        const unsigned char* Bs = smem + s * STAGE + (TM + wn * NT * 32 + l31) * ROWS;
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const bf16x8_t*>(As + mt * 32 * ROWS + ks * 32 + hi * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + nt * 32 * ROWS + ks * 32 + hi * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            // lane (row, hi) owns K elements hi*8 .. hi*8+7 of the 16-float chunk; MFMA j
            // consumes element j of both halves -- any K permutation is legal as long as
            // A and B use the same one.
            f32x4_t a[MT][2], b[NT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt][0] = *reinterpret_cast<const f32x4_t*>(As + mt * 32 * ROWS + hi * 32);
                a[mt][1] = *reinterpret_cast<const f32x4_t*>(As + mt * 32 * ROWS + hi * 32 + 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[nt][0] = *reinterpret_cast<const f32x4_t*>(Bs + nt * 32 * ROWS + hi * 32);
                b[nt][1] = *reinterpret_cast<const f32x4_t*>(Bs + nt * 32 * ROWS + hi * 32 + 16);
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
    };

    // ---- main loop: double-buffered, one barrier per chunk -----------------
    int tap = 0, chunk = 0;
    gload(0, 0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const bool more = (it + 1 < nk);
        if (more) {
            if (++chunk == nchunk) { chunk = 0; ++tap; }
            gload(tap, chunk);
        }
        compute(it & 1);
        if (more) lstore((it + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = q0 + wn * NT * 32 + nt * 32 + l31;
        if (q >= d.ngemm) continue;
        const int n = q / HWj, r = q - n * HWj;
        const int jy = r / d.Wj, jx = r - jy * d.Wj;
        const size_t opix = (size_t)((n * d.Hout + jy * d.osy + d.ooy) * d.Wout + jx * d.osx + d.oox);

        if constexpr (EPI == MG_EPI_PLAIN) {
            const T* __restrict__ Res = reinterpret_cast<const T*>(d.resid);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = m0 + wm * MT * 32 + mt * 32 + rq * 8 + hi * 4;
                    if (co >= d.Cout) continue;
                    f32x4_t v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][rq * 4 + j];
                    if (d.bias) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += d.bias[co + j];   // bias is padded to CoutP
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
            }
        } else {
            // SPADE: acc[0] = gamma rows, acc[1] = beta rows of the same 32 output channels.
            const T* __restrict__ X = reinterpret_cast<const T*>(d.x);
            T* __restrict__ G1 = reinterpret_cast<T*>(d.gamma_out);
            const int grow = m0 + wm * 64;              // first GEMM row of this wave's [gamma|beta] block
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int sub = rq * 8 + hi * 4;
                const int oc = (grow >> 1) + sub;
                if (oc >= d.Cout) continue;
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
            }
        }
    }
}

template <typename T, int WM, int WN, int MT, int NT, int EPI>
int launch_conv(ConvK& k, hipStream_t st)
{
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    k.tiles_m = (k.Cout_gemm + TM - 1) / TM;
    const int tiles_n = (k.ngemm + TN - 1) / TN;
    if (k.tiles_m * TM > k.CoutP)
        return mg_fail(MG_ERR_ARG, "mg_conv_taps: CoutP=%d too small for Cout_gemm=%d (tile %d)", k.CoutP, k.Cout_gemm, TM);
    const size_t lds = 2 * (size_t)(TM + TN) * ROWS;
    const long nblk = (long)k.tiles_m * tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_taps: bad grid %ld", nblk);
    hipLaunchKernelGGL((conv_taps_kernel<T, WM, WN, MT, NT, EPI>), dim3((unsigned)nblk), dim3(NTHR), lds, st, k);
    MG_CHECK_LAUNCH("mg_conv_taps");
    return MG_OK;
}

template <typename T>
int dispatch_conv(ConvK& k, int epilogue, hipStream_t st)
{
    if (epilogue == MG_EPI_SPADE) {
        if (k.Cout_gemm >= 128) return launch_conv<T, 2, 2, 2, 2, MG_EPI_SPADE>(k, st);
        return launch_conv<T, 1, 4, 2, 2, MG_EPI_SPADE>(k, st);
    }
    if (k.Cout_gemm > 64) return launch_conv<T, 2, 2, 2, 2, MG_EPI_PLAIN>(k, st);
    if (k.Cout_gemm > 32) return launch_conv<T, 1, 4, 2, 2, MG_EPI_PLAIN>(k, st);
    return launch_conv<T, 1, 4, 1, 2, MG_EPI_PLAIN>(k, st);
}

}  // namespace

extern "C" int mg_conv_taps(const mg_conv_desc* d, void* stream)
{
    MG_CHECK_ARG(d != nullptr, "mg_conv_taps: null descriptor");
    MG_CHECK_ARG(d->in && d->wt && d->out, "mg_conv_taps: null tensor pointer");
    MG_CHECK_ARG(d->dtype == MG_F32 || d->dtype == MG_BF16, "mg_conv_taps: bad dtype %d", d->dtype);
    MG_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= MG_MAX_TAPS, "mg_conv_taps: ntaps=%d out of range", d->ntaps);
    MG_CHECK_ARG(d->Cin > 0 && (d->Cin % 8) == 0, "mg_conv_taps: Cin=%d must be a positive multiple of 8", d->Cin);
    MG_CHECK_ARG(d->N > 0 && d->Hj > 0 && d->Wj > 0 && d->Hin > 0 && d->Win > 0, "mg_conv_taps: empty geometry");
    MG_CHECK_ARG(d->Cout > 0 && d->Cout_gemm > 0 && d->CoutP >= d->Cout_gemm && (d->CoutP % 128) == 0,
                 "mg_conv_taps: bad channel counts Cout=%d Cout_gemm=%d CoutP=%d", d->Cout, d->Cout_gemm, d->CoutP);
    MG_CHECK_ARG((long)d->N * d->Hj * d->Wj < (1L << 30), "mg_conv_taps: too many output pixels");
    MG_CHECK_ARG((d->Hj - 1) * d->osy + d->ooy < d->Hout && (d->Wj - 1) * d->osx + d->oox < d->Wout && d->ooy >= 0 && d->oox >= 0,
                 "mg_conv_taps: output grid exceeds output tensor");
    if (d->epilogue == MG_EPI_SPADE) {
        MG_CHECK_ARG(d->x && d->mean && d->rstd, "mg_conv_taps: SPADE epilogue needs x/mean/rstd");
        MG_CHECK_ARG(d->Cout_gemm == 2 * ((d->Cout + 31) / 32) * 32, "mg_conv_taps: SPADE needs Cout_gemm == 2*roundup(Cout,32)");
    } else {
        MG_CHECK_ARG(d->epilogue == MG_EPI_PLAIN, "mg_conv_taps: bad epilogue %d", d->epilogue);
        MG_CHECK_ARG(d->Cout <= d->Cout_gemm, "mg_conv_taps: Cout > Cout_gemm");
    }
    ConvK k;
    k.in = d->in; k.wt = d->wt; k.out = d->out; k.bias = d->bias; k.resid = d->resid; k.x = d->x;
    k.mean = d->mean; k.rstd = d->rstd; k.gamma_out = d->gamma_out;
    k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Cin = d->Cin;
    k.Hout = d->Hout; k.Wout = d->Wout; k.Cout = d->Cout; k.Cout_gemm = d->Cout_gemm; k.CoutP = d->CoutP;
    k.Hj = d->Hj; k.Wj = d->Wj; k.isy = d->isy; k.isx = d->isx;
    k.osy = d->osy; k.osx = d->osx; k.ooy = d->ooy; k.oox = d->oox;
    k.ntaps = d->ntaps; k.act = d->act; k.slope = d->slope;
    k.ngemm = d->N * d->Hj * d->Wj; k.tiles_m = 0;
    for (int t = 0; t < MG_MAX_TAPS; ++t)
        k.tap[t] = t < d->ntaps ? (int)((((uint32_t)(int)d->tap_dy[t]) & 0xffffu) | (((uint32_t)(int)d->tap_dx[t]) << 16)) : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == MG_BF16 ? dispatch_conv<uint16_t>(k, d->epilogue, st) : dispatch_conv<float>(k, d->epilogue, st);
}
