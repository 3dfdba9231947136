// mg_conv_halo64.hip -- 3x3 / stride-1 "same" convolution over a 64-CHANNEL input (bf16): the HBM-shaped member of the halo family.
//
// The 64 -> 64 layers at 512^2 (up_3.conv_1, VGG conv1_2, their data gradients) and 64 -> 128 (the data gradient of up_3.conv_0) move
// 0.5 - 0.8 GB per launch for 155 - 310 GFLOP: arithmetic intensity 192 - 288 flop/B, below the 312 flop/B ridge -- they are bandwidth
// bound, and the shipped <WM=1,NT=2> geometry (a 16x16-pixel tile per workgroup, K = 576: 18 tap bodies between a prologue and an
// epilogue, weights re-streamed by every workgroup) ran them at 2.8 - 3.4 TB/s and 0.26 of the MFMA peak: under both roofs
// (profiles/r05_conv_census.txt).  Here (VERDICT r5 "next" 4):
//
//   * ONE 8-wave workgroup per CU (two waves per SIMD, 256 registers each): a wave owns 64 pixels x 32 of the workgroup's 64 output channels, and
//     the weight fragments of SIX of its nine taps -- 6 x (64-channel K) x 32 rows = 96 registers -- stay in registers for the life of the
//     workgroup; the other three taps' 64 x 64 weights sit in LDS (24 KiB, loaded once).  No weight stream, no ring, no per-tap barrier: per K step a
//     wave issues 2 ds_read_b128 (its two pixel fragments; +1 for an LDS tap) for 2 MFMAs.  (Round 6 built the one-wave-per-SIMD form first -- all 72
//     fragments of 64 rows in 288 registers: 196 us on 64 -> 64 @ 8x512^2, a lone wave serialises K loop, epilogue and memory waits; this form: 180.)
//   * a workgroup walks a STRIP of 16x16-pixel tiles along x (a whole image row band, or a power-of-two fraction of it when the batch is
//     small).  The (16+2)^2-pixel x 128-byte input patch of tile i+1 is fetched by LDS-DMA into the second buffer while tile i computes:
//     ONE s_barrier per tile.  Row pieces are XOR-swizzled with the patch pixel index (source side) so that the shifted views of the nine
//     taps are conflict-free ds_read_b128 (same odd-row rotation of the pixel <-> lane map as mg_conv_halo.hip).
//   * residual / mask quads of a tile travel by LDS-DMA into the wave's own 4 KiB at the top of the tile (no registers across the K loop, no
//     barrier); the epilogue (bias, residual, activation, ReLU / LeakyReLU mask of a data gradient, half-wave quad exchange, 16-byte stores)
//     issues its stores unconditionally -- the kernel is only chosen for whole tiles (H, W multiples of 16; Cout a multiple of 64).
//   * strips are handed to workgroups so that one XCD works on neighbouring row bands of one image at the same time (the halo rows two
//     strips share are in that XCD's L2) and the output-channel blocks of one strip run next to each other (64 -> 128).
//   Measured (tools/bench_halo64.py, profiles/r06_halo64.txt): 64 -> 64 @ 8x512^2 180 / 244 / 239 us (bias+ReLU / residual / mask) against 216 / 292 /
//   293; batch 4: 78 / 105 / 111 against 120 / 172 / 175; 64 -> 128 masked data gradient 461 against 504.  tools/dbg_halo64.py: with stores, DMA and
//   K loop switched off in turn the K loops alone are 95 us, reads + writes alone 118 us (4.5 TB/s; a device copy of the same bytes: 99 us), the
//   fixed part 31 us -- what is left above max(compute, memory) is their imperfect overlap behind one barrier per tile.
//
// Same K order as the shipped kernel (64-byte chunk -> tap -> 16-channel K step), so the outputs are BITWISE those of
// conv3x3_halo_kernel<bf16, PLAIN, 1, 2> (tests/test_gpu_kernels.py::test_halo64_matches_the_shipped_halo_kernel_bitwise).
// mg_set_option(22, 0) sends these launches back to it.
#include "mg_conv_common.h"

int g_mg_conv_halo64 = 1;          // mg_set_option(22, v)
int g_mg_conv_halo64_dbg = 0;      // MEASUREMENT ONLY (mg_set_option(23, bits), wrong results): 1 = no stores, 2 = no patch DMA after the first tile, 4 = no K loop

namespace {

constexpr int H64_PW = 18;                       // patch width / height in pixels
constexpr int H64_ROWS = H64_PW * H64_PW;        // 324 patch pixels, 128 bytes each
constexpr int H64_PINSTR = 44;                   // LDS-DMA wave-instructions per patch (8 rows of 128 B each; 41 needed, 11 per wave)
constexpr int H64_THREADS = 512;                 // 8 waves: (4 groups of 64 pixels) x (2 halves of the 64 output channels); two waves per SIMD
constexpr int H64_PIPW = (H64_PINSTR + 7) / 8;
constexpr int H64_PBUF = H64_PINSTR * 1024;      // bytes per patch buffer
constexpr int H64_PAR = 2 * H64_PBUF;            // 64 floats of bias
constexpr int H64_AUX = H64_PAR + 256;           // residual / mask quads of the tile in flight: 4 KiB per wave, [8-channel granule][column tile nt][pixel l31] x 16 B
constexpr int H64_LTMAX = 4;                     // at most this many taps' weights are read from LDS (8 KiB each)
constexpr int H64_LDS = H64_AUX + 8 * 4096 + H64_LTMAX * 8192;
constexpr int H64_W8 = H64_AUX + 8 * 4096;       // the ninth tap's 64 x 64 weights (8 KiB): rows of 128 B, 16-byte pieces XOR-swizzled with (row >> 1) & 7
// taps whose A fragments live in registers (16 registers each): 6 of 9 (with 8 the allocator spilled 16 - 23 registers at two waves per SIMD, and a
// scratch reload in front of the next tile's DMA issue waits for the epilogue's stores); the other three taps' weights are read from LDS
template <int AUX> struct H64Taps { static constexpr int RT = 6, LT = 9 - RT; };
                       // taps whose A fragments live in registers (8 x 16 = 128 of a wave's 256); the ninth's come from LDS
constexpr int H64_NST = 8;                       // 16-byte-per-lane stores per wave and tile

struct Halo64Args {
    int units;              // workgroups = strips * mtiles * nseg
    int mtiles;             // Cout / 64
    int nseg;               // x segments per strip
    int tiles_per_seg;      // 16x16 tiles a workgroup walks
    int bands;              // H / 16
    int dbg;                // measurement bits (g_mg_conv_halo64_dbg)
};

template <int ACT, int AUX>   // ACT: 0 none, 1 relu, 2 lrelu (0 <= slope <= 1).  AUX: 0 none, 1 residual add, 2 data-gradient mask
__global__ __launch_bounds__(H64_THREADS, 2) void conv3x3_halo64_kernel(const ConvK d, const Halo64Args g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [patch 0][patch 1][bias][aux][tap 8 weights]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 3, mw = wave >> 2;                   // the wave's 64 pixels (tile rows 4 pw .. 4 pw + 3) and its 32 of the 64 output channels
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int H64_RT = H64Taps<AUX>::RT, H64_LT = H64Taps<AUX>::LT;

    int unit = blockIdx.x;
    if ((g.units & 7) == 0) unit = (blockIdx.x & 7) * (g.units >> 3) + (blockIdx.x >> 3);     // neighbouring units on one XCD (block b runs on XCD b % 8)
    const int m = unit % g.mtiles;  unit /= g.mtiles;
    const int seg = unit % g.nseg;  unit /= g.nseg;
    const int band = unit % g.bands;
    const int img = unit / g.bands;
    const int m0 = m * 64 + mw * 32, y0 = band * 16, xs = seg * g.tiles_per_seg * 16;
    const int W = d.Win, H = d.Hin;

    const uint16_t* __restrict__ In = reinterpret_cast<const uint16_t*>(d.in);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- weights: 32 A fragments per lane, [chunk c][tap t < 8][K step ks]; lane (l31, hi) holds row l31 of the wave's 32, K values hi*8 .. hi*8+7 ----
    bf16x8_t a[2][H64_RT][2];
    {
        const uint16_t* __restrict__ Wt = reinterpret_cast<const uint16_t*>(d.wt);
#pragma unroll
        for (int t = 0; t < H64_RT; ++t) {
            const uint16_t* row = Wt + ((size_t)t * d.CoutP + m0 + l31) * 64 + hi * 8;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    a[c][t][ks] = *reinterpret_cast<const bf16x8_t*>(row + c * 32 + ks * 16);
        }
    }
    // the last three taps' 64 rows each -> LDS (one LDS-DMA instruction per wave and tap: 8 rows of 128 B; lane -> row wave*8 + (lane >> 3), slot lane & 7)
#pragma unroll
    for (int lt = 0; lt < H64_LT; ++lt) {
        const int row = wave * 8 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
        glds16(reinterpret_cast<const uint16_t*>(d.wt) + ((size_t)(H64_RT + lt) * d.CoutP + m * 64 + row) * 64 + piece * 8,
               __builtin_amdgcn_readfirstlane(lds0 + H64_W8 + lt * 8192 + wave * 1024));
    }
    const int a8off = H64_W8 + (mw * 32 + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);      // ^ ((c*4 + ks*2) << 4)
    // bias of the workgroup's 64 rows -> LDS (LDS-DMA: the kernel contains no ds_write, see conv_stage_params_dma)
    if (wave == 0) {
        const void* src = (d.bias && m * 64 + lane < d.Cout_gemm) ? static_cast<const void*>(d.bias + m * 64 + lane) : static_cast<const void*>(g_mg_zeros + lane * 4);
        glds4(src, __builtin_amdgcn_readfirstlane(lds0 + H64_PAR));
    }
    const float* const par = reinterpret_cast<const float*>(smem + H64_PAR) + mw * 32;

    // ---- patch DMA: wave-instruction j of this wave fills rows blk*8 .. blk*8+7 (blk = wave + 8j < 44); lane -> (row q, 16-byte slot s) ----
    // slot s of row q holds channel piece s ^ ((q >> 1) & 7)  (source-side swizzle)
    const unsigned char* const zsrc = g_mg_zeros + (lane & 7) * 16;
    // byte offset of the lane's source piece in tile 0 of the strip, from In (signed: the left halo column of the first pixel is at -128; a multiple
    // of 16), with the lane's class in the low two bits: 0 always valid, 1 left halo column, 2 right halo column, 3 never (row out of the image / past the patch)
    int poff[H64_PIPW];
#pragma unroll
    for (int j = 0; j < H64_PIPW; ++j) {
        const int q = (wave + 8 * j) * 8 + (lane >> 3), s = lane & 7;
        const int pr = q / H64_PW, pc = q - pr * H64_PW;
        const int iy = y0 + pr - 1, ix = xs + pc - 1;
        const int piece = s ^ ((q >> 1) & 7);
        const bool rowok = q < H64_ROWS && (unsigned)iy < (unsigned)H;
        poff[j] = rowok ? ((int)((((long)(img * H + iy) * W + ix) * 64 + piece * 8) * 2) | (pc == 0 ? 1 : (pc == H64_PW - 1 ? 2 : 0))) : 3;
    }
    auto issue_patch = [&](int i) {                          // tile i of the strip -> buffer i & 1
        const unsigned base = lds0 + (i & 1) * H64_PBUF;
        const int x0 = xs + i * 16;
        const bool lok = x0 > 0, rok = x0 + 16 < W;
#pragma unroll
        for (int j = 0; j < H64_PIPW; ++j) {
            if (wave + 8 * j >= H64_PINSTR) continue;          // (wave-uniform) blocks past the buffer
            const int cls = poff[j] & 3;
            const bool ok = cls == 0 || (cls == 1 && lok) || (cls == 2 && rok);
            const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(In) + ((long)(poff[j] & ~3) + (long)i * (16 * 64 * 2)) : zsrc;
            glds16(src, __builtin_amdgcn_readfirstlane(base + (wave + 8 * j) * 1024));
        }
    };

    // ---- B fragment addresses: lane's pixel (row l31 >> 4, x = (l31 - 2 * row) & 15) of the wave's two 32-pixel column tiles, per tap ----
    // address = patch row q * 128 + ((piece ^ ((q >> 1) & 7)) << 4), piece = c*4 + ks*2 + hi: the (c, ks) part is an XOR with a constant
    int bbase[9][2];
    {
        const int py = l31 >> 4, px = (l31 - 2 * py) & 15;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tp = d.tap[t];
            const int dy = (int)(short)(tp & 0xffff), dx = tp >> 16;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int q = (pw * 4 + nt * 2 + py + 1 + dy) * H64_PW + px + 1 + dx;
                bbase[t][nt] = q * 128 + ((hi ^ ((q >> 1) & 7)) << 4);
            }
        }
    }
    // ---- output / auxiliary element offsets of the lane's two pixels (tile 0) at the wave's first channel, advanced by 16 pixels per tile ----
    unsigned ooff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = pw * 64 + nt * 32 + l31, py = p >> 4;
        const int y = y0 + py, x = xs + (((p & 15) - 2 * (py & 1)) & 15);
        ooff[nt] = (unsigned)(((img * d.Hout + y) * d.Wout + x) * d.Cout + m0);
    }
    uint16_t* __restrict__ Out = reinterpret_cast<uint16_t*>(d.out);
    const uint16_t* __restrict__ Aux = reinterpret_cast<const uint16_t*>(AUX == 1 ? d.resid : d.x);
    const float neg = ACT == 2 ? d.slope : 0.f;

    issue_patch(0);
    wait_vmcnt<0>();

    for (int i = 0; i < g.tiles_per_seg; ++i) {
        __builtin_amdgcn_s_barrier();                          // every wave's share of patch i has landed; nobody reads buffer (i + 1) & 1 any more
        if (i + 1 < g.tiles_per_seg && !(g.dbg & 2)) issue_patch(i + 1);
        // residual / mask quads of this tile: LDS-DMA into the wave's own 4 KiB (no registers held across the K loop, no barrier: a wave reads
        // back only what it fetched, behind its own vmcnt(0)).  Instruction gq moves the 8-channel granule gq of the wave's 32 channels for its 64
        // pixels: lane (l31, hi) fetches pixel l31 of column tile nt = hi -- whose offset it already has as ooff[hi].
        if constexpr (AUX != 0) {
            const unsigned abase = lds0 + H64_AUX + wave * 4096;
            const uint16_t* const asrc = Aux + (hi ? ooff[1] : ooff[0]);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) glds16(asrc + gq * 8, __builtin_amdgcn_readfirstlane(abase + gq * 1024));
        }
        f32x16_t acc[2];
        const unsigned char* const Ps = smem + (i & 1) * H64_PBUF;
        if (g.dbg & 4) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        } else {
        // 36 K steps in the shipped kernel's order (chunk c -> tap t -> K step ks): two pixel-fragment reads (+ one weight-fragment read for the
        // LDS taps) and two MFMAs each, scheduled by the compiler; the SIMD's second wave (the other 32 output channels of the same pixels) covers
        // the LDS round trips.  (Hand double-buffering the fragments one K step ahead with sched_group_barrier measured SLOWER -- 202 vs 180 us on
        // 64 -> 64 @ 8x512^2 -- and a body ahead spilled: profiles/r06_halo64.txt.)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t b[2], at;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        b[nt] = *reinterpret_cast<const bf16x8_t*>(Ps + (bbase[t][nt] ^ ((c * 4 + ks * 2) << 4)));
                    if (t < H64_RT) at = a[c][t < H64_RT ? t : 0][ks];
                    else at = *reinterpret_cast<const bf16x8_t*>(smem + (t - H64_RT) * 8192 + (a8off ^ ((c * 4 + ks * 2) << 4)));
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (c == 0 && t == 0 && ks == 0) {
                            const f32x16_t zero = {};
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, b[nt], zero, 0, 0, 0);
                        } else
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, b[nt], acc[nt], 0, 0, 0);
                    }
                }
        }
        wait_vmcnt<0>();                                       // patch i + 1 and this tile's auxiliary quads (the previous tile's stores are long gone)

        // ---- epilogue: the lean PLAIN body of conv_epilogue_fast (same operation order), stores unconditional ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {                  // the two channel quads (rq = 2 k2, 2 k2 + 1) behind one 16-byte store
                mg_pk2 v[4];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const int rq = 2 * k2 + r2;
                    const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(par + hi * 4 + rq * 8);
                    f32x4_t ax = {};
                    if constexpr (AUX != 0) {
                        const uint2 r = *reinterpret_cast<const uint2*>(smem + H64_AUX + wave * 4096 + rq * 1024 + (nt * 32 + l31) * 16 + hi * 8);
                        ax[0] = __uint_as_float(r.x << 16); ax[1] = __uint_as_float(r.x & 0xffff0000u);
                        ax[2] = __uint_as_float(r.y << 16); ax[3] = __uint_as_float(r.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        mg_pk2 tt = mg_pk(acc[nt][rq * 4 + 2 * h2], acc[nt][rq * 4 + 2 * h2 + 1]) + mg_pk(bias4[2 * h2], bias4[2 * h2 + 1]);
                        if constexpr (AUX == 1) tt += mg_pk(ax[2 * h2], ax[2 * h2 + 1]);
                        tt = mg_act2<ACT>(tt, neg);
                        if constexpr (AUX == 2) {
                            const mg_pk2 tm = tt * d.mslope;
                            tt[0] = ax[2 * h2] > 0.f ? tt[0] : tm[0];
                            tt[1] = ax[2 * h2 + 1] > 0.f ? tt[1] : tm[1];
                        }
                        v[r2 * 2 + h2] = tt;
                    }
                }
                const uint4 w = mg_pair_swap(mg_pack_bf16x2x2(v[0], v[1]), mg_pack_bf16x2x2(v[2], v[3]));
                if (!(g.dbg & 1)) *reinterpret_cast<uint4*>(Out + ooff[nt] + k2 * 16 + hi * 8) = w;
                else if (w.x == 0x12345678u) *reinterpret_cast<uint4*>(Out) = w;
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ooff[nt] += 16 * d.Cout;
    }
}

template <int ACT, int AUX>
int launch64(const ConvK& k, const Halo64Args& g, hipStream_t st)
{
    auto kern = conv3x3_halo64_kernel<ACT, AUX>;
    mg_raise_lds_cap(reinterpret_cast<const void*>(kern), H64_LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)g.units), dim3(H64_THREADS), H64_LDS, st, k, g);
    MG_CHECK_LAUNCH("mg_conv_taps(halo64)");
    return MG_OK;
}

int aux_kind(const ConvK& k) { return k.resid ? (k.x ? 3 : 1) : (k.x ? 2 : 0); }
int act_kind(const ConvK& k) { return (k.act == MG_ACT_LRELU && !(k.slope >= 0.f && k.slope <= 1.f)) ? 3 : k.act; }

}  // namespace

// bf16, PLAIN, 3x3 / stride 1 / same size over exactly 64 input channels onto a multiple of 64 output channels, whole 16x16 tiles,
// one of the lean epilogue cases of conv_epilogue_fast ({no aux} x {none, relu, lrelu}, {residual | mask} x {none}), 16-byte stores on
bool conv_halo64_applies(const ConvK& k, int dtype, int epilogue)
{
    if (!g_mg_conv_halo64 || dtype != MG_BF16 || epilogue != MG_EPI_PLAIN) return false;
    if (k.Cin != 64 || k.Cout != k.Cout_gemm || (k.Cout % 64) != 0 || k.CoutP < k.Cout) return false;
    if (k.ntaps != 9 || k.isy != 1 || k.isx != 1 || k.osy != 1 || k.osx != 1 || k.ooy != 0 || k.oox != 0) return false;
    if (k.Hj != k.Hin || k.Wj != k.Win || k.Hout != k.Hin || k.Wout != k.Win) return false;
    if ((k.Hin % 16) || (k.Win % 16) || k.Hin < 32 || k.Win < 32) return false;
    if (!(k.wide & 1) || (k.wide & ~1)) return false;                       // 16-byte stores; no measurement mode
    const int ak = aux_kind(k), ck = act_kind(k);
    if (!((ak == 0 && ck <= 2) || (ak <= 2 && ck == 0))) return false;
    if ((long)k.N * k.Hin * k.Win * 64 * 2 >= (1L << 31) || (long)k.N * k.Hin * k.Win * k.Cout >= (1L << 32)) return false;     // 32-bit byte / element offsets
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int dy = (int)(short)(k.tap[t] & 0xffff), dx = k.tap[t] >> 16;
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        seen |= 1u << ((dy + 1) * 3 + dx + 1);
    }
    if (seen != 0x1ffu) return false;
    // enough 16x16 tiles that every CU's workgroup walks at least four (the weights' 72 KiB per workgroup have to amortise)
    if (k.Cout > 128) return false;
    return (long)k.N * (k.Hin / 16) * (k.Win / 16) * (k.Cout / 64) >= 1024;
}

int launch_conv_halo64(ConvK& k, hipStream_t st)
{
    Halo64Args g;
    g.mtiles = k.Cout / 64;
    g.bands = k.Hin / 16;
    const int tiles_x = k.Win / 16;
    const long strips = (long)k.N * g.bands * g.mtiles;
    int nseg = 1;                                             // cut the strips until there are >= 256 workgroups (one per CU), at least 4 tiles each
    while (strips * nseg < 256 && (tiles_x % (nseg * 2)) == 0 && tiles_x / (nseg * 2) >= 4) nseg *= 2;
    g.nseg = nseg;
    g.tiles_per_seg = tiles_x / nseg;
    g.units = (int)(strips * nseg);
    g.dbg = g_mg_conv_halo64_dbg;
    const int ak = aux_kind(k), ck = act_kind(k);
    if (ak == 1) return launch64<0, 1>(k, g, st);
    if (ak == 2) return launch64<0, 2>(k, g, st);
    if (ck == 0) return launch64<0, 0>(k, g, st);
    if (ck == 1) return launch64<1, 0>(k, g, st);
    return launch64<2, 0>(k, g, st);
}
