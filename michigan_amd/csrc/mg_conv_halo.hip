// mg_conv_halo.hip -- 3x3 / stride-1 "same" convolution with an LDS-staged input halo tile.
//
// The generic tap-list kernel re-fetches the activation tile once per tap (9x for a 3x3 conv): at
// 128x128 tiles its operand stream (16 KiB per 1.05 MFLOP out of L2) is what bounds it at ~30 % MFMA
// utilisation (profiles/r01_pmc_conv_spade_shape.txt).  Here a workgroup owns an 8x16 pixel rectangle of
// one image and stages, per 64-byte channel chunk, its (8+2)x(16+2) input patch ONCE; the nine taps are
// then nine shifted views of that patch in LDS (row = (py+1+dy)*18 + px+1+dx), so only the weights
// (8 KiB per tap) keep streaming: 83.5 KiB instead of 144 KiB per chunk, and gamma/beta/conv weights of a
// channel tile are reused across the whole rectangle.  SPADE's fused gamma/beta conv, conv_0/conv_1, the VGG
// tower and their stride-1 dgrads all take this path (taps in {-1,0,1}^2, Cin a multiple of one chunk).
//
// LDS: weight ring 3 x 8 KiB (LDS-DMA, two taps in flight) + patch double buffer 2 x 12 KiB = 48 KiB ->
// 3 workgroups per CU.  One s_barrier per tap; vmcnt immediates are static per tap position because the
// nine taps of a chunk are fully unrolled.
#include "mg_conv_common.h"

namespace {

constexpr int TW = 16, PW = TW + 2;

// Two geometries (4 waves, each 64 channels x 64 pixels = 2x2 MFMA tiles):
//   WM=2: 128 channels x  8x16 pixels, patch 10x18 = 180 rows (12 DMA blocks), weights 8 KiB per tap
//   WM=1:  64 channels x 16x16 pixels, patch 18x18 = 324 rows (24 DMA blocks), weights 4 KiB per tap
//          (the 64-channel layers at 512^2 -- up_3, VGG conv1_2 -- stage 3.2x fewer bytes per FLOP than on
//          the generic 64x256 tile)
template <int WM> struct HaloGeom {
    static constexpr int WN = 4 / WM;
    static constexpr int TM = WM * 64;
    static constexpr int TH = WN * 64 / TW;                    // 8 or 16
    static constexpr int PROWS = (TH + 2) * PW;
    static constexpr int PBLK = ((PROWS + 15) / 16 + 3) / 4 * 4;   // whole blocks per wave
    static constexpr int PSTAGE = PBLK * 1024;
    static constexpr int ASTAGE = TM * ROWB;
    static constexpr int A_IPS = TM / 64, P_IPS = PBLK / 4;
    static constexpr int LDS = 3 * ASTAGE + 2 * PSTAGE;
};

template <typename T, int EPI, int WM>
__global__ __launch_bounds__(NTHR) void conv3x3_halo_kernel(const ConvK d)
{
    using G = HaloGeom<WM>;
    constexpr int MT = 2, NT = 2, WN = G::WN;
    constexpr int TH = G::TH, PROWS = G::PROWS, PSTAGE = G::PSTAGE, ASTAGE = G::ASTAGE, TM_H = G::TM;
    constexpr int A_IPS = G::A_IPS, P_IPS = G::P_IPS;
    constexpr int EPP = 16 / (int)sizeof(T);
    constexpr int CH  = ROWB / (int)sizeof(T);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const patches = smem + 3 * ASTAGE;

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
    const int tm = tile % d.tiles_m;  tile /= d.tiles_m;
    const int tx = tile % d.tiles_x;  tile /= d.tiles_x;
    const int ty = tile % d.tiles_y;
    const int img = tile / d.tiles_y;
    const int m0 = tm * TM_H, y0 = ty * TH, x0 = tx * TW;

    const T* __restrict__ In = reinterpret_cast<const T*>(d.in);
    const T* __restrict__ Wt = reinterpret_cast<const T*>(d.wt);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int lrow = lane >> 2;
    const int piece = (lane & 3) ^ ((lane >> 4) & 3);          // logical piece this lane fetches (source-side swizzle)
    const unsigned char* const zsrc = g_mg_zeros + (lane & 3) * 16;

    // patch source pointers: fixed pixels, walked +64 B per channel chunk
    const unsigned char* pp[P_IPS];
#pragma unroll
    for (int j = 0; j < P_IPS; ++j) {
        const int r = (wave + 4 * j) * 16 + lrow;
        const int pr = r / PW, pc = r - pr * PW;
        const int iy = y0 + pr - 1, ix = x0 + pc - 1;
        const bool ok = r < PROWS && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        pp[j] = ok ? reinterpret_cast<const unsigned char*>(In + ((size_t)((img * d.Hin + iy) * d.Win + ix) * d.Cin + piece * EPP)) : zsrc;
    }
    // weight source pointers for (tap 0, chunk 0)
    const unsigned char* pa0[A_IPS];
#pragma unroll
    for (int j = 0; j < A_IPS; ++j)
        pa0[j] = reinterpret_cast<const unsigned char*>(Wt + ((size_t)(m0 + (wave + 4 * j) * 16 + lrow) * d.Cin + piece * EPP));
    const size_t tapstride = (size_t)d.CoutP * d.Cin * sizeof(T);

    const int tapv = d.tap[lane];
    const int nchunk = d.Cin / CH;

    auto issue_patch = [&](int buf) {
        const unsigned base = lds0 + 3 * ASTAGE + buf * PSTAGE;
#pragma unroll
        for (int j = 0; j < P_IPS; ++j) {
            glds16(pp[j], __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
            pp[j] += ROWB;
        }
    };
    auto issue_a = [&](int slot, int tap, int chunk) {
        const unsigned base = lds0 + slot * ASTAGE;
        const size_t off = (size_t)tap * tapstride + (size_t)chunk * ROWB;
#pragma unroll
        for (int j = 0; j < A_IPS; ++j)
            glds16(pa0[j] + off, __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // lane's pixel inside the 8x16 rectangle, per MFMA column tile
    int prow0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = wn * 64 + nt * 32 + l31;
        prow0[nt] = ((p >> 4) + 1) * PW + (p & 15) + 1;      // TW == 16
    }
    const int swa = (l31 >> 2) & 3;

    auto compute = [&](int slot, int buf, int dy, int dx) {
        const unsigned char* As = ring + slot * ASTAGE + (wm * 64 + l31) * ROWB;
        const unsigned char* Pb = patches + buf * PSTAGE;
        int brow[NT], bsw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { brow[nt] = prow0[nt] + dy * PW + dx; bsw[nt] = (brow[nt] >> 2) & 3; }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const bf16x8_t*>(As + mt * 32 * ROWB + (((ks * 2 + hi) ^ swa) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const bf16x8_t*>(Pb + brow[nt] * ROWB + (((ks * 2 + hi) ^ bsw[nt]) << 4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            f32x4_t a[MT][2], b[NT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt][0] = *reinterpret_cast<const f32x4_t*>(As + mt * 32 * ROWB + (((hi * 2) ^ swa) << 4));
                a[mt][1] = *reinterpret_cast<const f32x4_t*>(As + mt * 32 * ROWB + (((hi * 2 + 1) ^ swa) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[nt][0] = *reinterpret_cast<const f32x4_t*>(Pb + brow[nt] * ROWB + (((hi * 2) ^ bsw[nt]) << 4));
                b[nt][1] = *reinterpret_cast<const f32x4_t*>(Pb + brow[nt] * ROWB + (((hi * 2 + 1) ^ bsw[nt]) << 4));
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

    // prologue: patch of chunk 0, weights of taps 0 and 1
    issue_patch(0);
    issue_a(0, 0, 0);
    issue_a(1, 1, 0);

    for (int c = 0; c < nchunk; ++c) {
        const bool next_chunk = (c + 1 < nchunk);
        static_for<0, 9>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            // loads younger than the weights of this tap: next tap's weights (2 per wave) and, right after
            // a chunk started, the next chunk's patch (3 per wave) -- see the issue order below
            if constexpr (t == 1 || t == 2) { if (next_chunk) wait_vmcnt<A_IPS + P_IPS>(); else wait_vmcnt<A_IPS>(); }
            else if constexpr (t == 8)      { if (next_chunk) wait_vmcnt<A_IPS>(); else wait_vmcnt<0>(); }
            else                            wait_vmcnt<A_IPS>();
            __builtin_amdgcn_s_barrier();
            // weights two taps ahead into the ring slot consumed at the previous tap
            if constexpr (t < 7) issue_a((t + 2) % 3, t + 2, c);
            else { if (next_chunk) issue_a((t + 2) % 3, t - 7, c + 1); }
            if constexpr (t == 0) { if (next_chunk) issue_patch((c + 1) & 1); }
            const int tp = __builtin_amdgcn_readlane(tapv, t);
            compute(t % 3, c & 1, (int)(short)(tp & 0xffff), tp >> 16);
        });
    }

    auto pixmap = [&](int p, size_t& opix) -> bool {
        const int y = y0 + (p >> 4), x = x0 + (p & 15);
        if (y >= d.Hout || x >= d.Wout) return false;
        opix = (size_t)((img * d.Hout + y) * d.Wout + x);
        return true;
    };
    conv_epilogue<T, MT, NT, EPI>(d, acc, m0, pixmap, wm, wn, l31, hi);
}

template <typename T, int EPI, int WM>
int launch_halo_g(ConvK& k, hipStream_t st)
{
    using G = HaloGeom<WM>;
    k.tiles_m = (k.Cout_gemm + G::TM - 1) / G::TM;
    k.tiles_y = (k.Hin + G::TH - 1) / G::TH;
    k.tiles_x = (k.Win + TW - 1) / TW;
    const long nblk = (long)k.N * k.tiles_y * k.tiles_x * k.tiles_m;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_taps(halo): bad grid %ld", nblk);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, EPI, WM>), dim3((unsigned)nblk), dim3(NTHR), G::LDS, st, k);
    MG_CHECK_LAUNCH("mg_conv_taps(halo)");
    return MG_OK;
}

template <typename T, int EPI>
int launch_halo(ConvK& k, hipStream_t st)
{
    if (k.Cout_gemm <= 64) return launch_halo_g<T, EPI, 1>(k, st);
    return launch_halo_g<T, EPI, 2>(k, st);
}

}  // namespace

int launch_conv_halo(ConvK& k, int dtype, int epilogue, hipStream_t st)
{
    if (dtype == MG_BF16)
        return epilogue == MG_EPI_SPADE ? launch_halo<uint16_t, MG_EPI_SPADE>(k, st) : launch_halo<uint16_t, MG_EPI_PLAIN>(k, st);
    return epilogue == MG_EPI_SPADE ? launch_halo<float, MG_EPI_SPADE>(k, st) : launch_halo<float, MG_EPI_PLAIN>(k, st);
}
