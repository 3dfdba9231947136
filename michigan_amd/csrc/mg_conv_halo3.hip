// mg_conv_halo3.hip -- the 3x3 / stride-1 halo-tile convolution built for THREE workgroups per CU (bf16).
//
// Why a second halo kernel.  The 128-accumulator tile of mg_conv_halo.hip holds two workgroups per CU (254 registers, 73 KiB LDS).  A
// K = 1152 workgroup (SPADE's gamma|beta conv, conv_0 / conv_1 of the 128-channel blocks) spends 28 - 33 % of its life outside the K loop
// (prologue 2.3 - 5.6 us, epilogue 4.2 - 6.4 us, dispatch gap 1.2 us of 32 - 36 us: profiles/r02_halo_probe.txt), and while one resident
// is out there the other's lone wave per SIMD reaches only ~55 % of the matrix pipe: it pays the issue of its own LDS-DMA instructions
// and of the partner's epilogue traffic in one in-order stream.  SQ_VALU_MFMA_BUSY_CYCLES 60 % forward (profiles/r04_pmc_halo.txt).
// Rounds 2 - 4 tried to shorten or hide those phases inside the two-resident structure (persistence x3, deeper rings, priorities,
// pipelined fragments, DMA placement, de-phased residents): all flat.  This kernel changes the structure instead: a THIRD resident, so
// that two waves per SIMD are still in their K loops while one workgroup is in its prologue / epilogue.
//
// What had to give to fit three residents (168 registers, 53.3 KiB LDS):
//   * wave tile 64 channels x 96 pixels (96 accumulator registers; workgroup = 128 channels x 12x16 pixels);
//   * NO address tables: the patch lives in LDS PIECE-MAJOR -- plane j holds the j-th 16-byte piece of every patch pixel
//     ([4 planes][252 pixels][16 B], one 4 KiB plane = 4 LDS-DMA instructions whose 64 lanes gather 16 bytes from 64 pixels) -- so the
//     address of (pixel + tap shift, K piece) is LINEAR in tap, column tile, K step and buffer: one VGPR per lane plus ds_read_b128
//     immediates (the XOR-swizzled 64-byte rows of mg_conv_halo.hip need 9 x NT registers and a v_xor per second K step).  Consecutive
//     pixels are consecutive 16-byte words: the 16 lanes of a ds_read_b128 group see distinct pixel indices mod 16 under the same
//     odd-row rotation as the big tile, so every read is conflict-free;
//   * one walking 64-bit source pointer per lane for the whole patch (its 4 pieces are +16 B apart), weights as a wave-uniform scalar
//     base + 2 constant lane offsets;
//   * a two-slot weight ring (one tap computing, one fetching; 16 KiB) + the double-buffered patch (2 x 16 KiB) + parameters = 49 KiB;
//     with three waves per SIMD a tap lasts >= 2 x 384 pipe cycles, enough for the next slab to land;
//   * SPADE's x quads are loaded in the epilogue (the 32 registers of the prefetch do not exist here); the third resident hides it.
// Arithmetic: identical instruction sequence per output element (chunk-major, taps in descriptor order, two K steps, same epilogue
// code) -> outputs are BITWISE equal to mg_conv_halo.hip's (tests/test_gpu_kernels.py::test_halo3_is_bit_identical...).
#include "mg_conv_common.h"

int g_mg_conv_halo3 = 0;           // mg_set_option(20, v): 0 = off, 1 = replaces the 128 x 16x16 tile where that one is chosen, 2 = every eligible bf16 launch

namespace {

constexpr int H3_NT = 3, H3_TH = 12, H3_TW = 16, H3_PW = H3_TW + 2;
constexpr int H3_PPIX = (H3_TH + 2) * H3_PW;                 // 252 patch pixels
constexpr int H3_PLANE = 4096;                               // one 16-byte piece of every patch pixel (252 x 16 B -> 4 DMA blocks of 1 KiB)
constexpr int H3_PSTAGE = 4 * H3_PLANE;                      // one 64-byte channel chunk of the patch
constexpr int H3_ASTAGE = 128 * ROWB;                        // one tap's weight slab: 128 rows x 64 B
constexpr int H3_PATCH0 = 2 * H3_ASTAGE;
constexpr int H3_PAR = H3_PATCH0 + 2 * H3_PSTAGE;
constexpr int H3_LDS = H3_PAR + 2 * 128 * 4;                 // 50176 B -> three workgroups per CU
static_assert(H3_PPIX <= 256 && 3 * H3_LDS <= 160 * 1024, "patch = one pixel per thread; three residents");

// Epilogue of the three-resident kernel: the lean bodies of conv_epilogue_fast (mg_conv_common.h) -- same helpers, same operation order
// per element, hence the same bits -- restructured for 168 registers: one (row tile | channel half, column tile) block at a time, the
// per-channel parameters re-read from LDS and the auxiliary quads (residual / mask / SPADE x) fetched as raw bf16 inside the block, so
// that nothing but the accumulators is live across blocks (the shared code requests every auxiliary quad of a tile up front and keeps
// the parameters of a channel half in registers: 180+ live registers against the accumulators of this tile -- 40 spilled dwords).
// The launcher only sends launches here whose epilogue is one of these bodies (halo3 eligibility, mg_conv.hip).
__device__ __forceinline__ f32x4_t h3_widen(uint2 r)
{
    f32x4_t v = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    return v;
}

template <int ACT, int AUX, typename PixMap>
__device__ __forceinline__ void halo3_epilogue_plain(const ConvK& d, f32x16_t (&acc)[2][H3_NT], int m0, PixMap&& pixmap,
                                                     int wm, int wn, int l31, int hi, const float* par)
{
    constexpr int NT = H3_NT;
    uint16_t* __restrict__ Out = reinterpret_cast<uint16_t*>(d.out);
    const uint16_t* __restrict__ Aux = reinterpret_cast<const uint16_t*>(AUX == 1 ? d.resid : d.x);
    const float neg = d.act == MG_ACT_NONE ? 1.f : (d.act == MG_ACT_RELU ? 0.f : d.slope);
    static_for<0, 2>([&](auto mt_) {
        constexpr int mt = decltype(mt_)::value;
        const int lr = wm * 64 + mt * 32 + hi * 4;
        static_for<0, NT>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            size_t opix, upix = 0;
            const bool pok = pixmap(wn * NT * 32 + nt * 32 + l31, opix, upix) && !(d.wide & 4);
            opix = pok ? opix * d.Cout : 0;
            uint2 aux[4] = {};
            if constexpr (AUX != 0) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = m0 + lr + rq * 8;
                    aux[rq] = *reinterpret_cast<const uint2*>(Aux + opix + (co < d.Cout ? co : 0));
                }
            }
            mg_pk2 v[8];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(par + lr + rq * 8);
                f32x4_t ax = {};
                if constexpr (AUX != 0) ax = h3_widen(aux[rq]);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    mg_pk2 t = mg_pk(acc[mt][nt][rq * 4 + 2 * h2], acc[mt][nt][rq * 4 + 2 * h2 + 1]) + mg_pk(bias4[2 * h2], bias4[2 * h2 + 1]);
                    if constexpr (AUX == 1) t += mg_pk(ax[2 * h2], ax[2 * h2 + 1]);
                    t = mg_act2<ACT>(t, neg);
                    if constexpr (AUX == 2) {
                        const mg_pk2 tm = t * d.mslope;
                        t[0] = ax[2 * h2] > 0.f ? t[0] : tm[0];
                        t[1] = ax[2 * h2 + 1] > 0.f ? t[1] : tm[1];
                    }
                    v[rq * 2 + h2] = t;
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const uint4 w = mg_pair_swap(mg_pack_bf16x2x2(v[4 * k2], v[4 * k2 + 1]), mg_pack_bf16x2x2(v[4 * k2 + 2], v[4 * k2 + 3]));
                const int co = m0 + wm * 64 + mt * 32 + k2 * 16 + hi * 8;
                if (pok && co < d.Cout) *reinterpret_cast<uint4*>(Out + opix + co) = w;
            }
            __builtin_amdgcn_sched_barrier(0);                 // the next block's loads stay behind this block's stores: short live ranges
        });
    });
}

template <int ACT, typename PixMap>
__device__ __forceinline__ void halo3_epilogue_spade(const ConvK& d, f32x16_t (&acc)[2][H3_NT], int m0, PixMap&& pixmap,
                                                     int wm, int wn, int l31, int hi, const float* par)
{
    constexpr int NT = H3_NT;
    uint16_t* __restrict__ Out = reinterpret_cast<uint16_t*>(d.out);
    uint16_t* __restrict__ G1 = reinterpret_cast<uint16_t*>(d.gamma_out);
    const uint16_t* __restrict__ X = reinterpret_cast<const uint16_t*>(d.x);
    const float neg = d.act == MG_ACT_NONE ? 1.f : (d.act == MG_ACT_RELU ? 0.f : d.slope);
    const int lrow = wm * 64;
    static_for<0, 2>([&](auto h_) {
        constexpr int h = decltype(h_)::value;
        const int ocw = ((m0 + lrow) >> 1) + h * 16 + hi * 8;
        static_for<0, NT>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            size_t opix, upix = 0;
            const bool pok = pixmap(wn * NT * 32 + nt * 32 + l31, opix, upix) && !(d.wide & 4);
            const unsigned xoff = pok ? (unsigned)((d.x_up ? upix : opix) * d.Cout) : 0u;
            opix = pok ? opix * d.Cout : 0;
            uint2 xr[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int occ = ((m0 + lrow) >> 1) + (h * 2 + q) * 8 + hi * 4;
                xr[q] = *reinterpret_cast<const uint2*>(X + xoff + (occ < d.Cout ? occ : 0));
            }
            mg_pk2 g[2][2], hv[2][2];
            const mg_pk2 one = mg_pk(1.f, 1.f);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sub = (h * 2 + q) * 8 + hi * 4;
                const f32x4_t bg = *reinterpret_cast<const f32x4_t*>(par + lrow + sub);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(par + lrow + 32 + sub);
                const f32x4_t mean4 = *reinterpret_cast<const f32x4_t*>(par + 128 + (lrow >> 1) + sub);
                const f32x4_t rstd4 = *reinterpret_cast<const f32x4_t*>(par + 128 + 64 + (lrow >> 1) + sub);
                const f32x4_t xv = h3_widen(xr[q]);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int e = (h * 2 + q) * 4 + 2 * h2;
                    g[q][h2] = (one + mg_pk(acc[0][nt][e], acc[0][nt][e + 1])) + mg_pk(bg[2 * h2], bg[2 * h2 + 1]);
                    const mg_pk2 bt = mg_pk(acc[1][nt][e], acc[1][nt][e + 1]) + mg_pk(bb[2 * h2], bb[2 * h2 + 1]);
                    const mg_pk2 xh = (mg_pk(xv[2 * h2], xv[2 * h2 + 1]) - mg_pk(mean4[2 * h2], mean4[2 * h2 + 1])) * mg_pk(rstd4[2 * h2], rstd4[2 * h2 + 1]);
                    hv[q][h2] = mg_act2<ACT>(mg_fma2(xh, g[q][h2], bt), neg);
                }
            }
            const uint4 hw = mg_pair_swap(mg_pack_bf16x2x2(hv[0][0], hv[0][1]), mg_pack_bf16x2x2(hv[1][0], hv[1][1]));
            uint4 gw = hw;
            if (G1) gw = mg_pair_swap(mg_pack_bf16x2x2(g[0][0], g[0][1]), mg_pack_bf16x2x2(g[1][0], g[1][1]));
            if (pok && ocw < d.Cout) {
                const size_t o = opix + ocw;
                *reinterpret_cast<uint4*>(Out + o) = hw;
                if (G1) *reinterpret_cast<uint4*>(G1 + o) = gw;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

template <int EPI, typename PixMap>
__device__ __forceinline__ void halo3_epilogue(const ConvK& d, f32x16_t (&acc)[2][H3_NT], int m0, PixMap&& pixmap,
                                               int wm, int wn, int l31, int hi, const float* par)
{
    using I0 = std::integral_constant<int, 0>;
    if constexpr (EPI == MG_EPI_PLAIN) {
        const int ak = d.resid ? 1 : (d.x ? 2 : 0);
        if (ak == 1)      halo3_epilogue_plain<0, 1>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else if (ak == 2) halo3_epilogue_plain<0, 2>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else if (d.act == MG_ACT_NONE) halo3_epilogue_plain<0, 0>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else if (d.act == MG_ACT_RELU) halo3_epilogue_plain<1, 0>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else                           halo3_epilogue_plain<2, 0>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
    } else {
        if (d.act == MG_ACT_NONE)      halo3_epilogue_spade<0>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else if (d.act == MG_ACT_RELU) halo3_epilogue_spade<1>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
        else                           halo3_epilogue_spade<2>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
    }
}

template <int EPI, bool FLIP>
__global__ __launch_bounds__(NTHR, 3) void conv3x3_halo3_kernel(const ConvK d)
{
    using T = uint16_t;
    constexpr int NT = H3_NT, MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const int m0 = tm * 128, y0 = ty * H3_TH, x0 = tx * H3_TW;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // patch: thread q fetches the four 16-byte pieces of patch pixel q (row q / 18, column q % 18), one per plane
    const unsigned char* pp;
    {
        const int q = tid;
        const int pr = q / H3_PW, pc = q - pr * H3_PW;
        const int iy = y0 + pr - 1, ix = x0 + pc - 1;
        const bool ok = q < H3_PPIX && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        pp = ok ? reinterpret_cast<const unsigned char*>(d.in) + (size_t)((img * d.Hin + iy) * d.Win + ix) * d.Cin * sizeof(T) : g_mg_zeros;
    }
    // weights: wave-uniform walking base + constant per-lane offsets (source-side XOR swizzle of the 64-byte rows)
    const unsigned char* wbase;
    {
        // (a 64-bit m0 * Cin product is computed on the VALU and the "s" operand of the DMA statement then gets a VGPR pair: keep it scalar)
        const unsigned long long w0 = (unsigned long long)(size_t)d.wt + (unsigned)m0 * (unsigned)d.Cin * (unsigned)sizeof(T);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)w0), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(w0 >> 32));
        wbase = reinterpret_cast<const unsigned char*>(((unsigned long long)hi32 << 32) | lo);
    }
    unsigned woff[2];
    {
        const int lrow = lane >> 2, piece = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) woff[j] = (unsigned)(((wave + 4 * j) * 16 + lrow) * d.Cin * (int)sizeof(T) + piece * 16);
    }
    const int tapstride = d.CoutP * d.Cin * (int)sizeof(T);       // 32-bit products stay on the scalar unit (CoutP * Cin * 2 <= 2^27 by the mg_conv_taps checks)
    const int chunkwrap = ROWB - 8 * tapstride;
    const int nchunk = d.Cin / 32;

    auto issue_patch = [&](int buf) {
        const unsigned base = lds0 + H3_PATCH0 + buf * H3_PSTAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(pp + j * 16, __builtin_amdgcn_readfirstlane(base + j * H3_PLANE));
        pp += ROWB;
    };
    auto issue_a = [&](int slot, bool last_tap) {
        const unsigned base = lds0 + slot * H3_ASTAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16_s(wbase, woff[j], __builtin_amdgcn_readfirstlane(base + (wave + 4 * j) * 1024));
        wbase += last_tap ? chunkwrap : tapstride;
    };

    f32x16_t acc[MT][NT];

    // operand addresses: A = row wm*64 + l31 (+32 per mt) of the slot, K piece hi (second K step: ^ 32 bytes);
    // B = plane hi (+2 per K step), pixel (wn*6 + py) * 18 + px of the tap-(-1,-1) view; everything else is an immediate
    const int aoff0 = (wm * 64 + l31) * ROWB + ((hi ^ ((l31 >> 2) & 3)) << 4);
    const int aoff1 = aoff0 ^ 32;
    const int py = l31 >> 4, px = (l31 - 2 * py) & 15;
    const int boff = H3_PATCH0 + hi * H3_PLANE + ((wn * (NT * 2) + py) * H3_PW + px) * 16;

    auto compute = [&](auto t_, auto slot_, auto buf_, auto first_) {
        constexpr int t = decltype(t_)::value, slot = decltype(slot_)::value, buf = decltype(buf_)::value;
        constexpr bool FIRST = decltype(first_)::value;
        constexpr int dy = FLIP ? 1 - t / 3 : t / 3 - 1, dx = FLIP ? 1 - t % 3 : t % 3 - 1;
        const unsigned char* As = smem + slot * H3_ASTAGE;
        const unsigned char* Bs = smem + buf * H3_PSTAGE + ((dy + 1) * H3_PW + dx + 1) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const bf16x8_t*>(As + mt * 32 * ROWB + (ks ? aoff1 : aoff0));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + boff + ks * 2 * H3_PLANE + nt * 2 * H3_PW * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (FIRST && ks == 0) {
                        const f32x16_t zero = {};
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], zero, 0, 0, 0);
                    } else
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    };

    auto pixmap = [&](int p, size_t& opix, size_t& upix) -> bool {
        const int pyy = p >> 4;
        const int y = y0 + pyy, x = x0 + (((p & 15) - 2 * (pyy & 1)) & 15);
        if (y >= d.Hout || x >= d.Wout) return false;
        opix = (size_t)((img * d.Hout + y) * d.Wout + x);
        upix = (size_t)((img * (d.Hout >> 1) + (y >> 1)) * (d.Wout >> 1) + (x >> 1));
        return true;
    };

    float* const par = reinterpret_cast<float*>(smem + H3_PAR);
    conv_stage_params_dma<128, EPI, 4>(d, m0, lds0 + H3_PAR, wave, lane);
    issue_patch(0);
    issue_a(0, false);

    // flat tap index T = 9 c + t: weights of T in ring slot T & 1 = (t + c) & 1, patch of chunk c in buffer c & 1
    auto chunk = [&](int c, auto par_, auto first_) {
        constexpr int P = decltype(par_)::value;
        const bool next_chunk = (c + 1 < nchunk);
        static_for<0, 9>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            // loads younger than this tap's weights: only the next chunk's patch, issued behind tap 1's weights during tap 0
            if constexpr (t == 1) { if (next_chunk) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (t < 8) issue_a((t + 1 + P) & 1, t == 7);
            else { if (next_chunk) issue_a((t + 1 + P) & 1, false); }
            if constexpr (t == 0) { if (next_chunk) issue_patch(P ^ 1); }
            compute(t_, std::integral_constant<int, (t + P) & 1>{}, par_, std::bool_constant<decltype(first_)::value && t == 0>{});
        });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    chunk(0, I0{}, std::true_type{});
    for (int c = 1; c < nchunk; c += 2) {
        chunk(c, I1{}, std::false_type{});
        if (c + 1 < nchunk) chunk(c + 1, I0{}, std::false_type{});
    }

    halo3_epilogue<EPI>(d, acc, m0, pixmap, wm, wn, l31, hi, par);
}

template <int EPI>
int launch_halo3(ConvK& k, bool flip, hipStream_t st)
{
    k.tiles_m = (k.Cout_gemm + 127) / 128;
    k.tiles_y = (k.Hin + H3_TH - 1) / H3_TH;
    k.tiles_x = (k.Win + H3_TW - 1) / H3_TW;
    const long nblk = (long)k.N * k.tiles_y * k.tiles_x * k.tiles_m;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_taps(halo3): bad grid %ld", nblk);
    if (flip) hipLaunchKernelGGL((conv3x3_halo3_kernel<EPI, true>), dim3((unsigned)nblk), dim3(NTHR), H3_LDS, st, k);
    else      hipLaunchKernelGGL((conv3x3_halo3_kernel<EPI, false>), dim3((unsigned)nblk), dim3(NTHR), H3_LDS, st, k);
    MG_CHECK_LAUNCH("mg_conv_taps(halo3)");
    return MG_OK;
}

}  // namespace

// 0 = not this kernel's case; 1 = taps in forward order ((t / 3 - 1, t % 3 - 1)); 2 = the data gradient's mirrored order
int conv_halo3_taporder(const ConvK& k)
{
    bool fwd = true, rev = true;
    for (int t = 0; t < 9; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        fwd = fwd && dy == t / 3 - 1 && dx == t % 3 - 1;
        rev = rev && dy == 1 - t / 3 && dx == 1 - t % 3;
    }
    return fwd ? 1 : (rev ? 2 : 0);
}

// the epilogue bodies this kernel carries (the lean cases of conv_epilogue_fast): channel counts in quads, wide stores, no tanh;
// PLAIN: {no aux} x {none, relu, lrelu in [0, 1]} or {residual | mask} x {none};  SPADE: none / relu / lrelu in [0, 1]
bool conv_halo3_epilogue_ok(const ConvK& k, int epilogue)
{
    if (((k.Cout | k.Cout_gemm) & 3) != 0 || k.act == MG_ACT_TANH || !(k.wide & 1)) return false;
    const int ck = (k.act == MG_ACT_LRELU && !(k.slope >= 0.f && k.slope <= 1.f)) ? 3 : k.act;
    if (epilogue == MG_EPI_SPADE) return ck <= 2;
    const int ak = (k.resid && k.x) ? 3 : (k.resid ? 1 : (k.x ? 2 : 0));
    return (ak == 0 && ck <= 2) || (ak <= 2 && ck == 0);
}

int launch_conv_halo3(ConvK& k, int epilogue, int taporder, hipStream_t st)
{
    return epilogue == MG_EPI_SPADE ? launch_halo3<MG_EPI_SPADE>(k, taporder == 2, st) : launch_halo3<MG_EPI_PLAIN>(k, taporder == 2, st);
}
