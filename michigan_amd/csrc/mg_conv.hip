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
// (exact fp32 fma; two-level sums, mma_f32_chunk) -- same tiling, same LDS image (64 B of K per row).

int g_mg_conv_splitk_wide = 0;   // ... also for launches of 161..320 workgroups with >= 256 K steps (mg_set_option(17, 1)): measured flat, off
int g_mg_conv_splitk = 1;        // deterministic split-K for low-resolution long-K layers (mg_set_option(5, v))
int g_mg_conv_halo_big = 1;      // 128 channels x 16x16 pixel halo tiles where the launch is big enough (mg_set_option(4, v))
int g_mg_conv_halo = 1;          // 3x3 stride-1 convs on the LDS halo-tile kernel (mg_set_option(2, v))
int g_mg_conv_bigtiles = 1;      // allow the 128x256 / 256x256 tiles (mg_set_option(1, v))
extern int g_mg_wgrad3x3;          // mg_wgrad.hip (mg_set_option(3, v))
extern int g_mg_norm_bwd_vec;      // mg_norm.hip (mg_set_option(19, v))
extern int g_mg_wgrad_min_stages;  // mg_wgrad.hip (mg_set_option(18, v))
extern int g_mg_conv_thin;         // mg_conv_thin.hip (mg_set_option(6, v))
extern int g_mg_conv_dot;          // mg_conv_dot.hip (mg_set_option(8, v))
extern int g_mg_wgrad3x3_stripe;   // mg_wgrad3x3.hip (mg_set_option(24, v))
extern int g_mg_conv_halo64;       // mg_conv_halo64.hip (mg_set_option(22, v))
extern int g_mg_conv_halo64_dbg;   // mg_conv_halo64.hip (mg_set_option(23, bits)): measurement only
int g_mg_conv_noxpre = 0;        // mg_set_option(15, 1): A/B switch, the SPADE halo kernel loads x in its epilogue instead of ahead of the main loop
int g_mg_conv_halo_ldspad = 0;  // MEASUREMENT ONLY (mg_set_option(21, bytes), MG_PROBES builds): extra dynamic LDS per halo workgroup -> fewer residents per CU
int g_mg_conv_dbg_noepi = 0;     // MEASUREMENT ONLY (mg_set_option(10, 1)): the halo kernel returns before its epilogue -- wrong results, main-loop time
int g_mg_conv_wide = 1;          // bf16 epilogues store 16 bytes per lane after a half-wave quad exchange (mg_set_option(7, v))
int g_mg_conv_pipeline = 1;      // 0 = register-staged double buffer, 1 = LDS-DMA 3-stage ring (mg_set_option(0, v))

#include "mg_conv_common.h"
int conv_halo_set_probe(unsigned long long addr);
int wgrad3x3_set_probe(unsigned long long addr);         // mg_wgrad3x3.hip (measurement build)
int g_mg_wgrad3x3_probe = 0;       // mg_set_option(12, 1): stamped build of wgrad3x3_kernel<2, 2>          // mg_conv_halo.hip (measurement builds)
static unsigned g_probe_lo = 0;
#include <map>
#include <mutex>

namespace {

template <typename T, int WM, int WN, int MT, int NT, int EPI, bool PACK>
__global__ __launch_bounds__(NTHR, 3) void conv_taps_kernel(const ConvK d)
{
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    constexpr int EPP = 16 / (int)sizeof(T);        // elements per 16-byte piece
    constexpr int CH  = ROWB / (int)sizeof(T);      // K elements per chunk
    constexpr int A_PT = (TM * 4 + NTHR - 1) / NTHR;
    constexpr int B_PT = (TN * 4) / NTHR;
    constexpr int STAGE = (TM + TN) * ROWB;
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
    const int nchunk = PACK ? 1 : (d.Cin + CH - 1) / CH;
    const int nk = PACK ? (d.ntaps + d.tpc - 1) / d.tpc : d.ntaps * nchunk;
    // PACK (Cin < one chunk): a 64-byte chunk holds tpc whole taps; this thread's piece sits in tap
    // (it * tpc + ptap) at channel pch -- 4x (bf16, Cin 8) fewer K iterations for the first-layer convs.
    const int ptap = PACK ? (piece * EPP) / d.Cin : 0;
    const int pch  = PACK ? (piece * EPP) % d.Cin : 0;

    auto gload = [&](int tap, int chunk) {
        if constexpr (PACK) { tap = tap * d.tpc + ptap; }
        const int c = PACK ? pch : chunk * CH + piece * EPP;
        const bool cv = PACK ? (tap < d.ntaps) : (c < d.Cin);
        const int tp = d.tap[cv || !PACK ? tap : 0];
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
            if (row < TM) *reinterpret_cast<uint4*>(base + lds_off(row, piece)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int row = srow + i * 64;
            *reinterpret_cast<uint4*>(base + TM * ROWB + lds_off(row, piece)) = rb[i];
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
        conv_compute<T, MT, NT>(smem + s * STAGE + (wm * MT * 32) * ROWB, smem + s * STAGE + (TM + wn * NT * 32) * ROWB,
                                l31, hi, acc);
    };

    // ---- main loop: double-buffered, one barrier per chunk -----------------
    int tap = 0, chunk = 0;
    float* const par = reinterpret_cast<float*>(smem + 2 * STAGE);        // epilogue channel parameters
    conv_stage_params<TM, EPI, NTHR>(d, m0, par, tid);
    gload(0, 0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const bool more = (it + 1 < nk);
        if (more) {
            if (++chunk == nchunk) { chunk = 0; ++tap; }     // PACK: nchunk == 1, `tap` counts chunk groups
            gload(tap, chunk);
        }
        compute(it & 1);
        if (more) lstore((it + 1) & 1);
        __syncthreads();
    }

    conv_epilogue<T, MT, NT, EPI, TM>(d, acc, m0, LinearPixMap{d, q0, HWj}, wm, wn, l31, hi, par);
}

// ---------------------------------------------------------------------------------------------
// Pipeline 2: LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring, two K chunks in flight.
//
// No staging VGPRs and no ds_write: each wave-instruction drops 64 lanes x 16 B = 16 rows x 64 B straight
// into LDS (destination = wave-uniform base + lane*16, so the XOR swizzle is applied on the SOURCE piece
// each lane fetches); out-of-image taps, tail pixels and tail channels fetch from a 64-byte block of
// zeros instead of being predicated off (a masked lane would leave stale LDS bytes).  The loads are
// inline asm so hipcc neither counts them nor drains them with vmcnt(0) before every ds_read; the
// schedule per K chunk is:  wait(vmcnt = one stage of this wave's loads still in flight) -> s_barrier
// -> issue chunk it+2 into the slot consumed at it-1 -> MFMAs on chunk it.
// ---------------------------------------------------------------------------------------------
// Tile geometries: WM x WN waves (4 or 8 per workgroup), each wave MT x NT MFMA 32x32 tiles.
//   (2,2,2,2) 128co x 128pix   (1,4,2,2) 64 x 256   (2,2,2,4) 128 x 256   (4,2,2,4) 256 x 256 (8 waves)
// The operand stream comes out of L2 at ~64 B/clk/CU and everything in flight must sit in LDS, so at a
// fixed LDS budget the only way to feed the matrix pipe faster is more FLOP per staged byte: the
// 256x256 tile needs half the bytes per FLOP of the 128x128 one.
template <typename T, int WM, int WN, int MT, int NT, int EPI, bool PACK>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? (MT * NT >= 8 ? 2 : 3) : 1)) void conv_taps_glds_kernel(const ConvK d)
{
    constexpr int NW = WM * WN;
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    constexpr int EPP = 16 / (int)sizeof(T);
    constexpr int CH  = ROWB / (int)sizeof(T);
    constexpr int A_IPS = TM / (16 * NW), B_IPS = TN / (16 * NW);   // 1 KiB wave-instructions per stage per wave
    constexpr int IPS = A_IPS + B_IPS;
    constexpr int STAGE = (TM + TN) * ROWB;
    constexpr int NS = (STAGE >= 32768) ? 4 : 3;                    // ring depth: NS-1 chunks in flight
    static_assert((NW == 4 || NW == 8) && TM % (16 * NW) == 0 && TN % (16 * NW) == 0, "tile must be whole 16-row blocks per wave");
    static_assert(EPI == MG_EPI_PLAIN || MT == 2, "SPADE epilogue needs gamma/beta tile pair");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ksp = tile / d.ntiles;        // split-K slice of this workgroup (0 when ksplit == 1)
    tile -= ksp * d.ntiles;
    const int tm = tile % d.tiles_m, tn = tile / d.tiles_m;
    const int m0 = tm * TM;
    const int q0 = tn * TN;

    const T* __restrict__ In = reinterpret_cast<const T*>(d.in);
    const T* __restrict__ Wt = reinterpret_cast<const T*>(d.wt);
    const int HWj = d.Hj * d.Wj;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // lane -> (row inside its 16-row block, LOGICAL piece it must fetch so that the physical slot
    // lane&3 holds piece ^ ((row>>2)&3); all block bases are multiples of 16 rows so (row>>2)&3 == (lane>>4)&3)
    const int lrow = lane >> 2;
    const int piece = (lane & 3) ^ ((lane >> 4) & 3);
    int b_n[B_IPS], b_y[B_IPS], b_x[B_IPS];
#pragma unroll
    for (int j = 0; j < B_IPS; ++j) {
        const int q = q0 + (wave + NW * j) * 16 + lrow;
        if (q < d.ngemm) {
            const int n = q / HWj, r = q - n * HWj;
            const int jy = r / d.Wj, jx = r - jy * d.Wj;
            b_n[j] = n; b_y[j] = jy * d.isy; b_x[j] = jx * d.isx;
        } else { b_n[j] = 0; b_y[j] = -(1 << 20); b_x[j] = 0; }
    }

    const int nchunk = PACK ? 1 : (d.Cin + CH - 1) / CH;
    const int nk_all = PACK ? (d.ntaps + d.tpc - 1) / d.tpc : d.ntaps * nchunk;
    // split-K: this workgroup walks K steps [kbeg, kbeg + nk) of the nk_all (tap, chunk) steps
    const int kper = (nk_all + d.ksplit - 1) / d.ksplit;
    const int kbeg = ksp * kper;
    const int nk = min(nk_all, kbeg + kper) - kbeg;
    const int ptap = PACK ? (piece * EPP) / d.Cin : 0;
    const int pch  = PACK ? (piece * EPP) % d.Cin : 0;

    // Source pointers are kept per row and WALKED: inside a tap the next chunk is +64 bytes for every row
    // (valid or not -- invalid rows walk the zero block), so the steady-state cost of a load is one 64-bit
    // add; tap decode, bounds checks and base addresses are redone only at tap boundaries (or every
    // iteration in PACK mode, where each chunk holds several taps).  The tap table sits in one VGPR
    // (lane t = tap t) and is read with v_readlane instead of a kernarg load + s_waitcnt per iteration.
    const int tapv = d.tap[lane];
    const unsigned char* pa[A_IPS];
    const unsigned char* pb[B_IPS];
    const bool tail_ok = (d.Cin % CH) == 0;      // otherwise the last chunk of a tap needs a channel check

    auto set_tap = [&](int tap, int chunk) {
        int ltap = tap;
        if constexpr (PACK) ltap = tap * d.tpc + ptap;
        const int c = PACK ? pch : chunk * CH + piece * EPP;
        const bool cv = PACK ? (ltap < d.ntaps) : (c < d.Cin);
        int tp;
        if constexpr (PACK) tp = d.tap[cv ? ltap : 0];
        else                tp = __builtin_amdgcn_readlane(tapv, tap);
        const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
#pragma unroll
        for (int j = 0; j < A_IPS; ++j) {
            const int blk = wave + NW * j;
            pa[j] = cv ? reinterpret_cast<const unsigned char*>(Wt + ((size_t)(ltap * d.CoutP + m0 + blk * 16 + lrow) * d.Cin + c))
                       : g_mg_zeros + (lane & 3) * 16;
        }
#pragma unroll
        for (int j = 0; j < B_IPS; ++j) {
            const int iy = b_y[j] + dy, ix = b_x[j] + dx;
            const bool ok = cv && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            pb[j] = ok ? reinterpret_cast<const unsigned char*>(In + ((size_t)((b_n[j] * d.Hin + iy) * d.Win + ix) * d.Cin + c))
                       : g_mg_zeros + (lane & 3) * 16;
        }
    };
    bool fresh = true;                       // a K slice may start in the middle of a tap: decode on the first issue
    auto issue = [&](int stage, int tap, int chunk) {
        if (PACK || chunk == 0 || fresh || (!tail_ok && chunk == nchunk - 1)) set_tap(tap, chunk);
        fresh = false;
        const unsigned sbase = lds0 + stage * STAGE;
#pragma unroll
        for (int j = 0; j < A_IPS; ++j) {
            glds16(pa[j], __builtin_amdgcn_readfirstlane(sbase + (wave + NW * j) * 1024));
            pa[j] += ROWB;
        }
#pragma unroll
        for (int j = 0; j < B_IPS; ++j) {
            glds16(pb[j], __builtin_amdgcn_readfirstlane(sbase + TM * ROWB + (wave + NW * j) * 1024));
            pb[j] += ROWB;
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // epilogue channel parameters (LDS-DMA, the oldest loads of each wave; no ds_write in this kernel)
    float* const par = reinterpret_cast<float*>(smem + NS * STAGE);
    conv_stage_params_dma<TM, EPI, NW>(d, m0, lds0 + NS * STAGE, wave, lane);
    // prologue: the first NS-1 chunks in flight
    int tap = kbeg / nchunk, chunk = kbeg - tap * nchunk;     // position of the NEXT chunk to issue
    auto advance = [&]() { if (++chunk == nchunk) { chunk = 0; ++tap; } };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) { issue(s, tap, chunk); advance(); }

    int slot = 0, islot = NS - 1;           // ring slot of chunk `it`, slot chunk it+NS-1 goes to
    for (int it = 0; it < nk; ++it) {
        // chunk `it` must have landed; younger chunks (issued later by this wave) may stay in flight
        const int younger = nk - 1 - it;
        if (NS == 4 && younger >= 2) wait_vmcnt<2 * IPS>();
        else if (younger >= 1)       wait_vmcnt<IPS>();
        else                         wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // every wave's part of chunk `it` has landed; chunk it-1 fully consumed
        if (it + NS - 1 < nk) { issue(islot, tap, chunk); advance(); }
        conv_compute<T, MT, NT>(smem + slot * STAGE + (wm * MT * 32) * ROWB,
                                smem + slot * STAGE + (TM + wn * NT * 32) * ROWB, l31, hi, acc);
        slot = (slot == NS - 1) ? 0 : slot + 1;
        islot = (islot == NS - 1) ? 0 : islot + 1;
    }

    if (d.ksplit > 1) {
        wait_vmcnt<0>();                    // (a slice without K steps still has its parameter DMA in flight)
        // partial sums of this K slice: plain fp32 stores into its own slab ws[ksp][q][co] (a slice that got no K
        // steps stores zeros); conv_splitk_finish adds the slabs in a fixed order -> bit-reproducible
        static_for<0, MT * NT>([&](auto i_) {
            constexpr int mt = decltype(i_)::value / NT, nt = decltype(i_)::value % NT;
            const int q = q0 + wn * NT * 32 + nt * 32 + l31;
            if (q >= d.ngemm) return;
            float* row = d.ws + ((size_t)ksp * d.ngemm + q) * d.Cout_gemm;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int co = m0 + wm * MT * 32 + mt * 32 + rq * 8 + hi * 4;
                if (co < d.Cout_gemm) {
                    f32x4_t v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][rq * 4 + j];
                    *reinterpret_cast<f32x4_t*>(row + co) = v;
                }
            }
        });
        return;
    }
    conv_epilogue<T, MT, NT, EPI, TM>(d, acc, m0, LinearPixMap{d, q0, HWj}, wm, wn, l31, hi, par);
}

// out[opix][co..co+3] = act(sum_s ws[s][q][co..] + bias + resid) for a split-K launch (PLAIN epilogue, Cout % 4 == 0)
template <typename T>
__global__ void conv_splitk_finish(const ConvK d)
{
    const int HWj = d.Hj * d.Wj, c4 = d.Cout / 4;
    const long n = (long)d.ngemm * c4;
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
    const T* __restrict__ Res = reinterpret_cast<const T*>(d.resid);
    const T* __restrict__ Msk = reinterpret_cast<const T*>(d.x);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % c4) * 4;
        const int q = (int)(i / c4);
        const int nimg = q / HWj, r = q - nimg * HWj;
        const int jy = r / d.Wj, jx = r - jy * d.Wj;
        const size_t o = (size_t)((nimg * d.Hout + jy * d.osy + d.ooy) * d.Wout + jx * d.osx + d.oox) * d.Cout + co;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (co < d.Cout_gemm) {
            for (int s = 0; s < d.ksplit; ++s) {
                const f32x4_t p = *reinterpret_cast<const f32x4_t*>(d.ws + ((size_t)s * d.ngemm + q) * d.Cout_gemm + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += p[j];
            }
            if (d.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += d.bias[co + j];
            }
        }
        if (Res) { const f32x4_t rv = ET<T>::load4(Res + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rv[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mg_act(v[j], d.act, d.slope);
        if (Msk) { const f32x4_t mk = ET<T>::load4(Msk + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!(mk[j] > 0.f)) v[j] *= d.mslope; }
        ET<T>::store4(Out + o, v);
    }
}

// grow-only fp32 scratch for split-K partial sums, one buffer per stream (stream order makes reuse safe)
float* splitk_workspace(hipStream_t st, size_t bytes)
{
    static std::mutex mu;
    static std::map<hipStream_t, std::pair<float*, size_t>> pool;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = pool[st];
    if (e.second < bytes) {
        if (e.first) { (void)hipStreamSynchronize(st); (void)hipFree(e.first); }
        e.first = nullptr; e.second = 0;
        const size_t want = bytes + bytes / 2;
        if (hipMalloc(reinterpret_cast<void**>(&e.first), want) != hipSuccess) { e.first = nullptr; return nullptr; }
        e.second = want;
    }
    return e.first;
}

template <typename T, int WM, int WN, int MT, int NT, int EPI, bool PACK>
int launch_conv_p(ConvK& k, hipStream_t st);

template <typename T, int WM, int WN, int MT, int NT, int EPI>
int launch_conv(ConvK& k, hipStream_t st)
{
    constexpr int CH = ROWB / (int)sizeof(T);
    k.tpc = (k.Cin < CH && CH % k.Cin == 0) ? CH / k.Cin : 1;
    if constexpr (EPI == MG_EPI_PLAIN) {
        if (k.tpc > 1) return launch_conv_p<T, WM, WN, MT, NT, EPI, true>(k, st);
    }
    k.tpc = 1;
    return launch_conv_p<T, WM, WN, MT, NT, EPI, false>(k, st);
}

template <typename T, int WM, int WN, int MT, int NT, int EPI, bool PACK>
int launch_conv_p(ConvK& k, hipStream_t st)
{
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    k.tiles_m = (k.Cout_gemm + TM - 1) / TM;
    const int tiles_n = (k.ngemm + TN - 1) / TN;
    if (k.tiles_m * TM > k.CoutP)
        return mg_fail(MG_ERR_ARG, "mg_conv_taps: CoutP=%d too small for Cout_gemm=%d (tile %d)", k.CoutP, k.Cout_gemm, TM);
    const long nblk = (long)k.tiles_m * tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_taps: bad grid %ld", nblk);
    if constexpr (TM % (16 * WM * WN) == 0) {
        if (g_mg_conv_pipeline == 1 || WM * WN != 4 || MT * NT > 4) {
            const size_t stage = (size_t)(TM + TN) * ROWB;
            const size_t ldsr = (stage >= 32768 ? 4 : 3) * stage + 2 * TM * sizeof(float);
            auto kern = conv_taps_glds_kernel<T, WM, WN, MT, NT, EPI, PACK>;
            k.ntiles = (int)nblk; k.ksplit = 1; k.ws = nullptr;
            // Low-resolution layers with a long K loop (1024-channel blocks at the 8x8..32x32 latents, the gamma/beta
            // dgrads there): a few dozen workgroups each pull hundreds of K steps through ONE CU's L2->LDS path
            // (~50 GB/s per CU, 0.33 us per step whatever the tile, ring depth or wave count) while most CUs idle.
            // Split K across workgroups: every slice stores its fp32 partial tile, a finishing pass adds the slices
            // in a fixed order (deterministic) and applies bias / residual / activation.
            if constexpr (EPI == MG_EPI_PLAIN && !PACK && WM == 2 && WN == 2 && MT == 2 && NT == 2) {
                const int nchunk = (k.Cin * (int)sizeof(T) + ROWB - 1) / ROWB, nkk = k.ntaps * nchunk;
                // (mg_set_option(17, 1) widens the rule to 161..320 workgroups with >= 256 K steps -- the 18432-deep gamma|beta data gradient at
                // 64x64 runs 256 workgroups, one per CU, at 760 TFLOP/s; two slices per tile measured 69.35 vs 69.31 ms per step: flat, left off)
                if (g_mg_conv_splitk && (nblk <= 160 || (nblk <= 320 && nkk >= 256 && g_mg_conv_splitk_wide)) && nkk >= 64 && (k.Cout & 3) == 0 && (k.Cout_gemm & 3) == 0) {
                    int S = (int)(((nblk <= 160 ? 384 : 512) + nblk - 1) / nblk);
                    if (S > nkk / 16) S = nkk / 16;
                    if (S > 16) S = 16;
                    if (S >= 2) {
                        k.ws = splitk_workspace(st, (size_t)S * k.ngemm * k.Cout_gemm * sizeof(float));
                        if (k.ws == nullptr) return mg_fail(MG_ERR_LAUNCH, "mg_conv_taps: split-K scratch allocation failed");
                        k.ksplit = S;
                        hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * S)), dim3(64 * WM * WN), ldsr, st, k);
                        const long nel = (long)k.ngemm * (k.Cout / 4);
                        const long fb = (nel + 255) / 256;
                        hipLaunchKernelGGL(conv_splitk_finish<T>, dim3((unsigned)(fb > 2048 ? 2048 : fb)), dim3(256), 0, st, k);
                        MG_CHECK_LAUNCH("mg_conv_taps(glds, split-K)");
                        return MG_OK;
                    }
                }
            }
            if (ldsr > 65536) {
                mg_raise_lds_cap(reinterpret_cast<const void*>(kern), (int)ldsr);       // once per instantiation and device
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), ldsr, st, k);
            MG_CHECK_LAUNCH("mg_conv_taps(glds)");
            return MG_OK;
        }
    }
    if constexpr (WM * WN != 4 || MT * NT > 4) return mg_fail(MG_ERR_UNSUPPORTED, "mg_conv_taps: large tiles need the LDS-DMA pipeline");
    const size_t lds = 2 * (size_t)(TM + TN) * ROWB + 2 * TM * sizeof(float);
    if constexpr (WM * WN == 4 && MT * NT <= 4) {
        hipLaunchKernelGGL((conv_taps_kernel<T, WM, WN, MT, NT, EPI, PACK>), dim3((unsigned)nblk), dim3(NTHR), lds, st, k);
        MG_CHECK_LAUNCH("mg_conv_taps");
    }
    return MG_OK;
}

template <typename T, int EPI>
int dispatch_tiles(ConvK& k, hipStream_t st)
{
    // Largest tile that still yields >= 1.5 workgroups per CU (256 CUs); big tiles halve the bytes
    // staged per FLOP, small ones keep low-resolution layers from under-filling the chip.
    auto wgs = [&](int tm, int tn) { return (long)((k.Cout_gemm + tm - 1) / tm) * ((k.ngemm + tn - 1) / tn); };
    const bool big = g_mg_conv_pipeline == 1 && g_mg_conv_bigtiles;
    // 256x256 (one 8-wave workgroup per CU) only pays when the K loop is long enough to amortise its
    // exposed prologue/epilogue: measured +10..14 % at K >= 4608, -5..-13 % at K = 1152 (SPADE convs).
    const int kchunks = k.ntaps * ((k.Cin * (int)sizeof(T) + ROWB - 1) / ROWB);
    if (big && kchunks >= 64 && k.Cout_gemm >= 256 && (k.CoutP % 256) == 0 && wgs(256, 256) >= 384)
        return launch_conv<T, 4, 2, 2, 4, EPI>(k, st);
    if (k.Cout_gemm > 64) return launch_conv<T, 2, 2, 2, 2, EPI>(k, st);
    if (EPI == MG_EPI_SPADE || k.Cout_gemm > 32) return launch_conv<T, 1, 4, 2, 2, EPI>(k, st);
    if constexpr (EPI == MG_EPI_PLAIN) return launch_conv<T, 1, 4, 1, 2, EPI>(k, st);
    return MG_ERR_UNSUPPORTED;
}

// 3x3 / stride-1 / same-size convs with whole K chunks go to the halo-tile kernel (mg_conv_halo.hip)
template <typename T>
bool halo_applies(const ConvK& k)
{
    constexpr int CH = ROWB / (int)sizeof(T);
    if (!g_mg_conv_halo || g_mg_conv_pipeline != 1) return false;
    if (k.ntaps != 9 || k.isy != 1 || k.isx != 1 || k.osy != 1 || k.osx != 1 || k.ooy != 0 || k.oox != 0) return false;
    if (k.Hj != k.Hin || k.Wj != k.Win || k.Hout != k.Hin || k.Wout != k.Win) return false;
    if (k.Cin % CH || k.Cout_gemm <= 32 || k.Hin < 8 || k.Win < 16) return false;
    if (k.Cout_gemm <= 64 && k.Hin < 16) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        seen |= 1u << ((dy + 1) * 3 + dx + 1);
    }
    if (seen != 0x1ffu) return false;
    const int th = k.Cout_gemm <= 64 ? 16 : 8, tmh = k.Cout_gemm <= 64 ? 64 : 128;
    const long wgs = (long)k.N * ((k.Hin + th - 1) / th) * ((k.Win + 15) / 16) * ((k.Cout_gemm + tmh - 1) / tmh);
    return wgs >= 384;
}

template <typename T>
int dispatch_conv(ConvK& k, int epilogue, hipStream_t st)
{
    if (conv_thin_applies(k, ET<T>::DT, epilogue)) return launch_conv_thin(k, st);
    if (conv_thin_taps_applies(k, ET<T>::DT, epilogue)) return launch_conv_thin_taps(k, st);
    if (conv_dot_applies(k, ET<T>::DT, epilogue)) return launch_conv_dot(k, ET<T>::DT, st);
    if (conv_fewout_applies(k, ET<T>::DT, epilogue)) return launch_conv_fewout(k, st);
    if (conv_halo64_applies(k, ET<T>::DT, epilogue)) return launch_conv_halo64(k, st);
    if (halo_applies<T>(k)) return launch_conv_halo(k, ET<T>::DT, epilogue, st);
    return epilogue == MG_EPI_SPADE ? dispatch_tiles<T, MG_EPI_SPADE>(k, st) : dispatch_tiles<T, MG_EPI_PLAIN>(k, st);
}

}  // namespace

float* mg_stream_scratch(hipStream_t st, size_t bytes) { return splitk_workspace(st, bytes); }

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
    MG_CHECK_ARG((long)d->Cin * (d->dtype == MG_BF16 ? 2 : 4) <= 8192, "mg_conv_taps: Cin too large (%d)", d->Cin);
    MG_CHECK_ARG((d->Hj - 1) * d->osy + d->ooy < d->Hout && (d->Wj - 1) * d->osx + d->oox < d->Wout && d->ooy >= 0 && d->oox >= 0,
                 "mg_conv_taps: output grid exceeds output tensor");
    if (d->epilogue == MG_EPI_SPADE) {
        MG_CHECK_ARG(d->x && d->mean && d->rstd, "mg_conv_taps: SPADE epilogue needs x/mean/rstd");
        MG_CHECK_ARG(d->Cout_gemm == 2 * ((d->Cout + 31) / 32) * 32, "mg_conv_taps: SPADE needs Cout_gemm == 2*roundup(Cout,32)");
        MG_CHECK_ARG(d->x_up == 0 || (d->x_up == 1 && (d->Hout % 2) == 0 && (d->Wout % 2) == 0 &&
                                      (long)d->N * d->Hout * d->Wout * d->Cout < (1L << 32)),
                     "mg_conv_taps: x_up needs even output sizes (and fewer than 2^32 elements)");
    } else {
        MG_CHECK_ARG(d->epilogue == MG_EPI_PLAIN, "mg_conv_taps: bad epilogue %d", d->epilogue);
        MG_CHECK_ARG(d->Cout <= d->Cout_gemm, "mg_conv_taps: Cout > Cout_gemm");
        MG_CHECK_ARG(d->mask_slope >= 0.f && d->mask_slope <= 1.f, "mg_conv_taps: mask_slope %g outside [0, 1]", (double)d->mask_slope);
    }
    ConvK k;
    k.in = d->in; k.wt = d->wt; k.out = d->out; k.bias = d->bias; k.resid = d->resid; k.x = d->x;
    k.mean = d->mean; k.rstd = d->rstd; k.gamma_out = d->gamma_out;
    k.N = d->N; k.Hin = d->Hin; k.Win = d->Win; k.Cin = d->Cin;
    k.Hout = d->Hout; k.Wout = d->Wout; k.Cout = d->Cout; k.Cout_gemm = d->Cout_gemm; k.CoutP = d->CoutP;
    k.Hj = d->Hj; k.Wj = d->Wj; k.isy = d->isy; k.isx = d->isx;
    k.osy = d->osy; k.osx = d->osx; k.ooy = d->ooy; k.oox = d->oox;
    k.ntaps = d->ntaps; k.act = d->act; k.slope = d->slope;
    k.x_up = d->epilogue == MG_EPI_SPADE ? d->x_up : 0;
    k.mslope = d->mask_slope;
    k.ngemm = d->N * d->Hj * d->Wj; k.tiles_m = 0; k.tpc = 1; k.tiles_y = k.tiles_x = 0;
    k.ksplit = 1; k.ntiles = 1; k.ws = nullptr;
    k.wide = ((g_mg_conv_wide && d->dtype == MG_BF16 && (d->Cout % 8) == 0) ? 1 : 0) | (g_mg_conv_dbg_noepi ? 2 : 0) | (g_mg_conv_dbg_noepi == 6 ? 4 : 0) | (g_mg_conv_noxpre ? 8 : 0);
    for (int t = 0; t < MG_MAX_TAPS; ++t)
        k.tap[t] = t < d->ntaps ? (int)((((uint32_t)(int)d->tap_dy[t]) & 0xffffu) | (((uint32_t)(int)d->tap_dx[t]) << 16)) : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == MG_BF16 ? dispatch_conv<uint16_t>(k, d->epilogue, st) : dispatch_conv<float>(k, d->epilogue, st);
}

extern "C" int mg_set_option(int32_t key, int32_t value)
{
    if (key == 0 && (value == 0 || value == 1)) { g_mg_conv_pipeline = value; return MG_OK; }
    if (key == 1 && (value == 0 || value == 1)) { g_mg_conv_bigtiles = value; return MG_OK; }
    if (key == 2 && (value == 0 || value == 1)) { g_mg_conv_halo = value; return MG_OK; }
    if (key == 3 && (value == 0 || value == 1)) { g_mg_wgrad3x3 = value; return MG_OK; }
    if (key == 4 && (value == 0 || value == 1)) { g_mg_conv_halo_big = value; return MG_OK; }
    if (key == 5 && (value == 0 || value == 1)) { g_mg_conv_splitk = value; return MG_OK; }
    if (key == 17 && (value == 0 || value == 1)) { g_mg_conv_splitk_wide = value; return MG_OK; }
    if (key == 6 && value >= 0 && value <= 2) { g_mg_conv_thin = value; return MG_OK; }
    if (key == 18 && value >= 1 && value <= 1024) { g_mg_wgrad_min_stages = value; return MG_OK; }
    if (key == 19 && (value == 0 || value == 1)) { g_mg_norm_bwd_vec = value; return MG_OK; }
    if (key == 7 && (value == 0 || value == 1)) { g_mg_conv_wide = value; return MG_OK; }
    if (key == 8 && value >= 0 && value <= 2) { g_mg_conv_dot = value; return MG_OK; }
#if MG_PROBES
    // measurement builds only (python tools/build_variant.py probes mg_conv.hip mg_conv_halo.hip mg_wgrad3x3.hip -DMG_PROBES=1): truncated /
    // stamped variants of the big halo tile and of wgrad3x3_kernel, their stamp buffer, the SPADE x prefetch off
    if (key == 15 && (value == 0 || value == 1)) { g_mg_conv_noxpre = value; return MG_OK; }
    if (key == 13) { g_probe_lo = (unsigned)value; return MG_OK; }
    if (key == 14) { const unsigned long long a = ((unsigned long long)(unsigned)value << 32) | g_probe_lo; const int r = conv_halo_set_probe(a); return r != MG_OK ? r : wgrad3x3_set_probe(a); }
    if (key == 12 && (value == 0 || value == 1)) { g_mg_wgrad3x3_probe = value; return MG_OK; }
    if (key == 10 && value >= 0 && value <= 6) { g_mg_conv_dbg_noepi = value; return MG_OK; }
    if (key == 21 && value >= 0 && value <= 80 * 1024) { g_mg_conv_halo_ldspad = value; return MG_OK; }
#endif
    if (key == 22 && (value == 0 || value == 1)) { g_mg_conv_halo64 = value; return MG_OK; }
    if (key == 23 && value >= 0 && value <= 7) { g_mg_conv_halo64_dbg = value; return MG_OK; }
    if (key == 24 && value >= 0 && value <= 4096 && (value % 32) == 0) { g_mg_wgrad3x3_stripe = value; return MG_OK; }
    return mg_fail(MG_ERR_ARG, "mg_set_option: unknown key/value %d/%d", key, value);
}
