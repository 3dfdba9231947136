// mg_wgrad.hip -- convolution weight gradient, split-K over output pixels.
//
// GEMM view per (tap, split):  dW[tap][co][ci] += sum_q dY[q][co] * X[pix(q)+tap][ci]
//   rows (MFMA "i") = co, cols (MFMA "n") = ci, K = output pixels q.
// NHWC keeps CHANNELS contiguous, but here K runs over PIXELS, so the MFMA
// fragments (8 consecutive K per lane for bf16) are a transpose of the natural
// image.  Tiles are staged pixel-major in LDS ([KP pixels][128 channels], the
// coalesced global image) and transposed on the way out:
//   bf16: ds_read_b64_tr_b16 (gfx950 hardware transpose read; each 16-lane group
//         turns a [4 pixels][16 channels] block into 4-pixels-per-lane fragments),
//         with a ds_read_u16 gather as a selectable fallback (flags bit0 = 0);
//   f32 : v_mfma_f32_32x32x2_f32 takes one K value per lane, so the pixel-major
//         image is read directly (32 consecutive floats per half-wave).
// Partial sums are accumulated into the fp32 dW with hardware float atomics.
#include "mg_common.h"
#include "mg_wgrad_common.h"

int g_mg_wgrad3x3 = 1;     // mg_set_option(3, v): 0 = always the generic tap-per-workgroup kernel

int g_mg_wgrad_min_stages = 32;      // mg_set_option(18, v): stages (of 32 / 16 pixels) a split of the generic kernel keeps at least

namespace {

constexpr int NTHR = 256;

struct WgK {
    const void* x; const void* dy; float* dw; float* dbias;
    int N, Hin, Win, Cin, Hj, Wj, Cg, isy, isx, ntaps;
    int K;            // N*Hj*Wj
    int kper;         // pixels per split (multiple of KP)
    int tiles_m, tiles_n, splitk;
    int tpt;          // taps packed into one 128-wide N tile (Cin < 128 and 128 % Cin == 0), else 1
    long det_stride;  // deterministic mode: floats per split slab (dw / dbias point into the workspace), 0 = fp32 atomics
    int tap[MG_MAX_TAPS];
};

template <typename T, bool TR>
__global__ __launch_bounds__(NTHR) void wgrad_kernel(const WgK d)
{
    constexpr bool BF = (sizeof(T) == 2);
    constexpr int KP  = BF ? 32 : 16;                       // pixels per stage
    constexpr int EPP = 16 / (int)sizeof(T);
    constexpr int PPR = 128 * (int)sizeof(T) / 16;          // 16-byte pieces per pixel row (128 channels)
    constexpr int RPP = NTHR / PPR;                         // rows per pass
    constexpr int NPASS = KP / RPP;                         // = 2
    constexpr int RS  = BF ? 320 : 528;                     // LDS row stride in bytes
    constexpr int OPB = KP * RS;                            // one operand tile
    constexpr int STAGE = 2 * OPB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    // N tiles: tpt == 1 -> (tap, 128-channel slice); tpt > 1 -> groups of tpt taps x all Cin channels
    const int ngroups = (d.ntaps + d.tpt - 1) / d.tpt;
    const int tm = tile % d.tiles_m;  tile /= d.tiles_m;
    const int tn = tile % d.tiles_n;  tile /= d.tiles_n;
    const int tg = tile % ngroups;
    const int split = tile / ngroups;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tap0 = tg * d.tpt;
    const int kbeg = split * d.kper;
    const int kend = min(d.K, kbeg + d.kper);
    if (kbeg >= kend) return;


    const T* __restrict__ X  = reinterpret_cast<const T*>(d.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(d.dy);
    const int HWj = d.Hj * d.Wj;

    const int piece = tid % PPR, prow = tid / PPR;
    const int ca = m0 + piece * EPP;      // dy channel of this thread's piece
    const bool cav = ca < d.Cg;
    // x side: this thread's piece belongs to tap `mytap`, channels cb..cb+EPP-1 (loop invariant)
    int mytap, cb;
    if (d.tpt > 1) { const int e = piece * EPP; mytap = tap0 + e / d.Cin; cb = e % d.Cin; }
    else           { mytap = tap0; cb = n0 + piece * EPP; }
    const bool cbv = (mytap < d.ntaps) && (cb < d.Cin);
    const int tp = d.tap[mytap < d.ntaps ? mytap : 0];
    const int tdy = (int)(short)(tp & 0xffff), tdx = tp >> 16;

    uint4 ra[NPASS], rb[NPASS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int q = k0 + prow + i * RPP;
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (q < kend) {
                if (cav) va = *reinterpret_cast<const uint4*>(DY + ((size_t)q * d.Cg + ca));
                if (cbv) {
                    const int n = q / HWj, r = q - n * HWj;
                    const int jy = r / d.Wj, jx = r - jy * d.Wj;
                    const int iy = jy * d.isy + tdy, ix = jx * d.isx + tdx;
                    if ((unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win)
                        vb = *reinterpret_cast<const uint4*>(X + ((size_t)((n * d.Hin + iy) * d.Win + ix) * d.Cin + cb));
                }
            }
            ra[i] = va; rb[i] = vb;
        }
    };
    auto lstore = [&](int s) {
        unsigned char* base = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = prow + i * RPP;
            *reinterpret_cast<uint4*>(base + row * RS + piece * 16) = ra[i];
            *reinterpret_cast<uint4*>(base + OPB + row * RS + piece * 16) = rb[i];
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Fused bias gradient: the workgroups of the first N tile / tap group also column-sum the dY pieces
    // they stage anyway (dbias[co] = sum over pixels of dY), replacing a separate full pass over dY.
    const bool do_bias = d.dbias != nullptr && tg == 0 && tn == 0;
    float bsum[EPP];
#pragma unroll
    for (int j = 0; j < EPP; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const uint32_t w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            if constexpr (BF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { bsum[2 * j] += __uint_as_float(w[j] << 16); bsum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum[j] += __uint_as_float(w[j]);
            }
        }
    };

    auto compute = [&](int s) {
        const unsigned char* As = smem + s * STAGE;
        const unsigned char* Bs = As + OPB;
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t a[2], b[2];
                if constexpr (TR) {
                    // 16-lane group g2: channel block (g2&1)*16, K half g2>>1 (= hi).
                    const int i16 = lane & 15, g2 = lane >> 4;
                    const int row = ks * 16 + (g2 >> 1) * 8 + (i16 >> 2);
                    const int col = (g2 & 1) * 16 + (i16 & 3) * 4;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        typedef __attribute__((address_space(3))) s16x4_t* lp_t;
                        const unsigned char* pa = As + row * RS + (wm * 64 + t * 32 + col) * 2;
                        const unsigned char* pb = Bs + row * RS + (wn * 64 + t * 32 + col) * 2;
                        s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pa));
                        s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pa + 4 * RS));
                        s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pb));
                        s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pb + 4 * RS));
                        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
                        s16x8_t av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        s16x8_t bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        a[t] = __builtin_bit_cast(bf16x8_t, av);
                        b[t] = __builtin_bit_cast(bf16x8_t, bv);
                    }
                } else {
                    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        s16x8_t av, bv;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int row = ks * 16 + hi * 8 + j;
                            av[j] = *reinterpret_cast<const short*>(As + row * RS + (wm * 64 + t * 32 + l31) * 2);
                            bv[j] = *reinterpret_cast<const short*>(Bs + row * RS + (wn * 64 + t * 32 + l31) * 2);
                        }
                        a[t] = __builtin_bit_cast(bf16x8_t, av);
                        b[t] = __builtin_bit_cast(bf16x8_t, bv);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            // two-level sums (mg_conv_common.h, mma_f32_chunk): a panel's KP pixel terms start from zero in a temporary tile and the block sum
            // is added to the split's accumulator once -- the split's sum over up to 10^5 pixels is otherwise ONE sequential fp32 chain
            f32x16_t t[2][2];
#pragma unroll
            for (int kk = 0; kk < KP / 2; ++kk) {
                float a[2], b[2];
                const int row = kk * 2 + hi;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a[q] = *reinterpret_cast<const float*>(As + row * RS + (wm * 64 + q * 32 + l31) * 4);
                    b[q] = *reinterpret_cast<const float*>(Bs + row * RS + (wn * 64 + q * 32 + l31) * 4);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        f32x16_t c;
                        if (MG_F32_ONE_CHAIN) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                            continue;
                        }
                        if (kk == 0) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) c[e] = 0.f;
                        } else {
                            c = t[mt][nt];
                        }
                        t[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], c, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (MG_F32_ONE_CHAIN) continue;
                    acc[mt][nt] += t[mt][nt];
                    asm volatile("" : "+v"(acc[mt][nt]));            // keep the add here (see mma_f32_chunk)
                }
        }
    };

    const int nk = (kend - kbeg + KP - 1) / KP;
    gload(kbeg);
    if (do_bias) bias_acc();
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const bool more = (it + 1 < nk);
        if (more) gload(kbeg + (it + 1) * KP);
        compute(it & 1);
        if (more) { if (do_bias) bias_acc(); lstore((it + 1) & 1); }
        __syncthreads();
    }
    if (do_bias) {                       // reduce the RPP thread rows through the (idle) LDS, one atomic per channel
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int j = 0; j < EPP; ++j) red[prow * 128 + piece * EPP + j] = bsum[j];
        __syncthreads();
        if (tid < 128 && m0 + tid < d.Cg) {
            float t = 0.f;
            for (int r = 0; r < RPP; ++r) t += red[r * 128 + tid];
            wg_accum(d.dbias, d.det_stride, split, (size_t)(m0 + tid), t);
        }
    }

    // ---- accumulate the tile into dW (fp32 atomics; lanes 0..31 = 32 consecutive ci)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int nn = wn * 64 + nt * 32 + l31;            // column inside the N tile
            int tap, ci;
            if (d.tpt > 1) { tap = tap0 + nn / d.Cin; ci = nn % d.Cin; }
            else           { tap = tap0; ci = n0 + nn; }
            if (tap >= d.ntaps || ci >= d.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < d.Cg)
                    wg_accum(d.dw, d.det_stride, split, (size_t)(tap * d.Cg + co) * d.Cin + ci, acc[mt][nt][r]);
            }
        }
}

// dst[i] += sum over splits in a FIXED order: dw for i < ndw, dbias behind it.  A 256-thread block owns 32 outputs; its 8 thread rows
// take every 8th split each (in split order), then the 8 row sums are added in row order.  (One thread per output walking all splits
// was 19 blocks x 512 dependent steps for a 3x3 / 8-channel layer: 100 us behind a 50 us kernel.)
__global__ __launch_bounds__(256) void wgrad_det_finish_kernel(const float* __restrict__ ws, int nsplit, long stride, float* __restrict__ dw, long ndw,
                                                               float* __restrict__ dbias, int nbias)
{
    __shared__ float red[256];
    const long total = ndw + nbias;
    const int cl = threadIdx.x & 31, row = threadIdx.x >> 5;
    for (long i0 = blockIdx.x * 32L; i0 < total; i0 += gridDim.x * 32L) {
        const long i = i0 + cl;
        float a = 0.f;
        if (i < total) {
            const float* p = ws + i;
            int sp = row;
            for (; sp + 24 < nsplit; sp += 32) {                 // four independent loads in flight
                const float v0 = p[(size_t)sp * stride], v1 = p[(size_t)(sp + 8) * stride], v2 = p[(size_t)(sp + 16) * stride], v3 = p[(size_t)(sp + 24) * stride];
                a += v0; a += v1; a += v2; a += v3;
            }
            for (; sp < nsplit; sp += 8) a += p[(size_t)sp * stride];
        }
        __syncthreads();
        red[threadIdx.x] = a;
        __syncthreads();
        if (row == 0 && i < total) {
#pragma unroll
            for (int r = 1; r < 8; ++r) a += red[r * 32 + cl];
            if (i < ndw) dw[i] += a; else dbias[i - ndw] += a;
        }
    }
}

template <typename T, bool TR>
int launch_wgrad(WgK& k, hipStream_t st, int* nsplit = nullptr, bool dry = false)
{
    constexpr bool BF = (sizeof(T) == 2);
    constexpr int KP = BF ? 32 : 16;
    constexpr int RS = BF ? 320 : 528;
    k.tiles_m = (k.Cg + 127) / 128;
    k.tpt = (k.Cin < 128 && 128 % k.Cin == 0) ? 128 / k.Cin : 1;
    k.tiles_n = k.tpt > 1 ? 1 : (k.Cin + 127) / 128;
    const long base = (long)k.tiles_m * k.tiles_n * ((k.ntaps + k.tpt - 1) / k.tpt);
    int S = k.splitk;
    if (S <= 0) {
        S = (int)((1536 + base - 1) / base);                 // ~6 workgroups per CU in flight
        // >= 32 stages per split: every split ends with a pass of atomics over its 128 x 128 tile; with the 8-stage floor the mid-size
        // stride-2 / 1x1 layers ran 1500 workgroups of 12 stages and took 0.10 ms where 32-stage splits take 0.065 (tools/variant_sweep.py)
        int maxS = (k.K + KP * g_mg_wgrad_min_stages - 1) / (KP * g_mg_wgrad_min_stages);
        if (base * maxS < 256) {                               // ... unless that leaves CUs idle (tiny layers): down to 4 stages per split
            const int fill = (int)((256 + base - 1) / base), floor4 = (k.K + KP * 4 - 1) / (KP * 4);
            maxS = fill < floor4 ? fill : floor4;
        }
        if (S > maxS) S = maxS;
        if (S < 1) S = 1;
    }
    k.kper = ((k.K + S - 1) / S + KP - 1) / KP * KP;
    S = (k.K + k.kper - 1) / k.kper;
    k.splitk = S;
    const long nblk = base * S;
    if (nsplit) *nsplit = S;
    if (dry) return MG_OK;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_wgrad: bad grid %ld", nblk);
    const size_t lds = 2 * 2 * (size_t)KP * RS;           // (capping THIS kernel at one workgroup per CU beside another stream cost +1.6 ms per step: not done)
    hipLaunchKernelGGL((wgrad_kernel<T, TR>), dim3((unsigned)nblk), dim3(NTHR), lds, st, k);
    MG_CHECK_LAUNCH("mg_conv_wgrad");
    return MG_OK;
}

}  // namespace

int launch_wgrad_det_finish(const float* ws, int nsplit, long stride, float* dw, long ndw, float* dbias, int nbias, hipStream_t st)
{
    const long total = ndw + nbias;
    long grid = (total + 31) / 32;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(wgrad_det_finish_kernel, dim3((unsigned)grid), dim3(256), 0, st, ws, nsplit, stride, dw, ndw, dbias, nbias);
    MG_CHECK_LAUNCH("mg_conv_wgrad(deterministic finish)");
    return MG_OK;
}

namespace {

// One routing function for the launch and for the workspace query: picks the kernel exactly as the launch would; with `dry` it only
// reports the split count that kernel's launcher chooses.  det (dw_ws != nullptr): partial sums go to slabs of dw_ws.
int route_wgrad(const mg_wgrad_desc* d, hipStream_t st, float* dw, float* dbias, long det_stride, int* nsplit, bool dry)
{
    WgK k;
    k.x = d->x; k.dy = d->dy; k.dw = dw; k.dbias = dbias;
    k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Cin = d->Cin;
    k.Hj = d->Hj; k.Wj = d->Wj; k.Cg = d->Cg; k.isy = d->isy; k.isx = d->isx; k.ntaps = d->ntaps;
    k.K = d->N * d->Hj * d->Wj; k.splitk = d->splitk; k.kper = 0; k.tiles_m = k.tiles_n = 0; k.tpt = 1; k.det_stride = det_stride;
    for (int t = 0; t < MG_MAX_TAPS; ++t)
        k.tap[t] = t < d->ntaps ? (int)((((uint32_t)(int)d->tap_dy[t]) & 0xffffu) | (((uint32_t)(int)d->tap_dx[t]) << 16)) : 0;
    if (d->dtype == MG_BF16 && d->ntaps == 9 && d->isy == 1 && d->isx == 1 && d->Hin == d->Hj && d->Win == d->Wj && d->Cin == 8) {
        bool std3x3 = true;
        for (int t = 0; t < 9; ++t) std3x3 = std3x3 && d->tap_dy[t] == t / 3 - 1 && d->tap_dx[t] == t % 3 - 1;
        Wg3K k3;
        k3.x = d->x; k3.dy = d->dy; k3.dw = dw; k3.dbias = dbias;
        k3.N = d->N; k3.H = d->Hin; k3.W = d->Win; k3.Cin = d->Cin; k3.Cg = d->Cg;
        k3.nstg = 0; k3.sps = 0; k3.tiles_m = k3.tiles_n = 0; k3.splitk = 0; k3.det_stride = det_stride; k3.half_cu = 0; k3.stripe_w = 0;
        if (std3x3 && wgrad_thin_applies(k3)) return launch_wgrad_thin(k3, st, nsplit, dry);
    }
    if (wgrad_thin_taps_applies(d)) return launch_wgrad_thin_taps(d, st, dw, dbias, det_stride, nsplit, dry);
    if (g_mg_wgrad3x3 && d->dtype == MG_BF16 && (d->flags & 1) && d->ntaps == 9 && d->isy == 1 && d->isx == 1 &&
        d->Hin == d->Hj && d->Win == d->Wj && (d->Win == 16 || d->Win % 32 == 0) && (d->Hin * d->Win) % 32 == 0 &&
        d->Cin >= 64 && d->Cg >= 64) {
        bool std3x3 = true;
        for (int t = 0; t < 9; ++t) std3x3 = std3x3 && d->tap_dy[t] == t / 3 - 1 && d->tap_dx[t] == t % 3 - 1;
        if (std3x3) {
            Wg3K k3;
            k3.x = d->x; k3.dy = d->dy; k3.dw = dw; k3.dbias = dbias;
            k3.N = d->N; k3.H = d->Hin; k3.W = d->Win; k3.Cin = d->Cin; k3.Cg = d->Cg;
            k3.nstg = k.K / 32; k3.sps = 0; k3.tiles_m = k3.tiles_n = 0; k3.splitk = d->splitk; k3.det_stride = det_stride;
            k3.half_cu = (d->flags & 2) ? 1 : 0; k3.stripe_w = 0;
            return launch_wgrad3x3(k3, st, nsplit, dry);
        }
    }
    if (d->dtype == MG_BF16)
        return (d->flags & 1) ? launch_wgrad<uint16_t, true>(k, st, nsplit, dry) : launch_wgrad<uint16_t, false>(k, st, nsplit, dry);
    return launch_wgrad<float, false>(k, st, nsplit, dry);
}

int check_wgrad_desc(const mg_wgrad_desc* d, const char* who)
{
    MG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    MG_CHECK_ARG(d->x && d->dy && d->dw, "%s: null tensor pointer", who);
    MG_CHECK_ARG(d->dtype == MG_F32 || d->dtype == MG_BF16, "%s: bad dtype %d", who, d->dtype);
    MG_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= MG_MAX_TAPS, "%s: ntaps=%d out of range", who, d->ntaps);
    MG_CHECK_ARG(d->Cin > 0 && (d->Cin % 8) == 0 && d->Cg > 0 && (d->Cg % 8) == 0,
                 "%s: Cin=%d / Cg=%d must be positive multiples of 8", who, d->Cin, d->Cg);
    MG_CHECK_ARG(d->N > 0 && d->Hj > 0 && d->Wj > 0 && d->Hin > 0 && d->Win > 0, "%s: empty geometry", who);
    MG_CHECK_ARG((long)d->N * d->Hj * d->Wj < (1L << 30), "%s: too many pixels", who);
    return MG_OK;
}

}  // namespace

extern "C" int64_t mg_wgrad_det_workspace(const mg_wgrad_desc* d)
{
    if (check_wgrad_desc(d, "mg_wgrad_det_workspace") != MG_OK) return -1;
    int S = 0;
    if (route_wgrad(d, nullptr, d->dw, d->dbias, 0, &S, true) != MG_OK || S <= 0) return -1;
    const long slab = (long)d->ntaps * d->Cg * d->Cin + (d->dbias ? d->Cg : 0);
    return (int64_t)S * slab * (int64_t)sizeof(float);
}

extern "C" int mg_conv_wgrad(const mg_wgrad_desc* d, void* stream)
{
    const int rc = check_wgrad_desc(d, "mg_conv_wgrad");
    if (rc != MG_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->det_ws == nullptr) return route_wgrad(d, st, d->dw, d->dbias, 0, nullptr, false);
    // deterministic split-K: every split stores its partial tile into its own slab, a finishing launch adds the slabs in split order
    int S = 0;
    if (route_wgrad(d, st, d->dw, d->dbias, 0, &S, true) != MG_OK || S <= 0) return mg_fail(MG_ERR_ARG, "mg_conv_wgrad: cannot size the deterministic workspace");
    const long ndw = (long)d->ntaps * d->Cg * d->Cin, nbias = d->dbias ? d->Cg : 0, slab = ndw + nbias;
    MG_CHECK_ARG(d->det_ws_bytes >= (int64_t)S * slab * (int64_t)sizeof(float), "mg_conv_wgrad: deterministic workspace too small (%ld splits)", (long)S);
    float* ws = reinterpret_cast<float*>(d->det_ws);
    int S2 = 0;
    const int r2 = route_wgrad(d, st, ws, d->dbias ? ws + ndw : nullptr, slab, &S2, false);
    if (r2 != MG_OK) return r2;
    if (S2 != S) return mg_fail(MG_ERR_LAUNCH, "mg_conv_wgrad: split count changed between sizing and launch (%d vs %d)", S, S2);
    return launch_wgrad_det_finish(ws, S, slab, d->dw, ndw, d->dbias, (int)nbias, st);
}
