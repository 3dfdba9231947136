// mg_conv_common.h -- pieces shared by the conv translation units (generic tap-list kernels in
// mg_conv.hip, 3x3 halo-tile kernel in mg_conv_halo.hip).  Everything here is static / inline: the
// library is built without relocatable device code, so each TU carries its own copy.
#pragma once
#ifndef MG_PROBES
#define MG_PROBES 0        // 1: measurement builds (stamped / truncated kernel variants + their mg_set_option keys); tools/build_variant.py
#endif
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
    int ksplit, ntiles;     // generic LDS-DMA kernel: split-K factor (1 = off) and tiles per K slice
    float* ws;              // split-K: fp32 partial sums [ksplit][ngemm][Cout_gemm]
    int wide;               // bf16 epilogue: lanes l and l+32 exchange quads so that every lane stores 16 contiguous bytes
    int x_up;               // SPADE: x is the half-resolution source of a nearest 2x upsample (read at (y >> 1, x >> 1))
    float mslope;           // PLAIN with a mask x: masked-off values are multiplied by this (0 = ReLU mask, s = LeakyReLU(s) backward)
};
namespace {

// LDS image shared by both pipelines: rows of 64 bytes of K, NO padding; the four 16-byte pieces of a
// row are XOR-swizzled with (row >> 2) & 3.  ds_write_b128 (8-lane groups = 2 rows x 4 pieces, bank =
// addr/4 mod 32) and ds_read_b128 (16-lane groups with rows distinct mod 16, bank = addr/4 mod 64) are
// both conflict-free with it; the previous 80-byte padded rows made every ds_write_b128 2-way
// (SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_conv.txt).
__device__ __forceinline__ int lds_off(int row, int piece) { return row * ROWB + ((piece ^ ((row >> 2) & 3)) << 4); }


// fp32: the 16 K values of one 64-byte chunk, for every 32x32 tile of a wave, as TWO-LEVEL sums.  v_mfma_f32_32x32x2_f32 adds two K terms
// to its accumulator per issue; chaining every issue of a layer into the same accumulator makes each output ONE sequential fp32 chain of
// K = taps x Cin terms (1152 ... 9216 on the hot path), whose rounding error grows like sqrt(K): 4.4e-7 ... 1.2e-6 relative against
// 1.3e-7 ... 3.1e-7 for a CPU kernel that keeps 16 lanes of partial sums (tools/fp32_chain_error.py) -- and every (Leaky)ReLU whose
// pre-activation changes sign under that error moves a gradient by its full magnitude (tools/grad_probe.py).  Here the
// chunk's 8 issues start from ZERO in a temporary tile and the finished 16-term block sum is added to the layer accumulator once:
// blocks of 16, then a chain of K / 16 block sums -- 1.7e-7 ... 4.4e-7, a CPU kernel's error to within 1.4x.  GT = 2 temporaries in flight keep
// dependent issues GT MFMAs apart; 16 v_add_f32 per tile and chunk ride in the shadow of 8 x 64 MFMA cycles.
template <int MT, int NT>
__device__ __forceinline__ void mma_f32_chunk(const f32x4_t (&a)[MT][2], const f32x4_t (&b)[NT][2], f32x16_t (&acc)[MT][NT])
{
#if MG_F32_ONE_CHAIN                                              // A/B build (tools/build_variant.py ... -DMG_F32_ONE_CHAIN=1): the single chain of rounds 1-3
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j >> 2][j & 3], b[nt][j >> 2][j & 3], acc[mt][nt], 0, 0, 0);
    return;
#endif
    constexpr int TILES = MT * NT, GT = TILES >= 2 ? 2 : TILES;
    static_assert(TILES % GT == 0, "tile groups");
#pragma unroll
    for (int g0 = 0; g0 < TILES; g0 += GT) {
        f32x16_t t[GT];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                const int mt = (g0 + g) / NT, nt = (g0 + g) % NT;
                f32x16_t c;
                if (j == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) c[e] = 0.f;
                } else {
                    c = t[g];
                }
                t[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j >> 2][j & 3], b[nt][j >> 2][j & 3], c, 0, 0, 0);
            }
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int mt = (g0 + g) / NT, nt = (g0 + g) % NT;
            acc[mt][nt] += t[g];
            // pin the add HERE: nothing needs the layer accumulator before the epilogue, so the compiler otherwise sinks every chunk's
            // adds down there and keeps all the block sums alive until then (9 taps x 64 registers of them: 750 spills)
            asm volatile("" : "+v"(acc[mt][nt]));
        }
    }
}

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
        mma_f32_chunk<MT, NT>(a, b, acc);
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Per-channel epilogue parameters of a workgroup's TM GEMM rows, staged once into LDS (2*TM floats at `par`):
//   PLAIN: par[r] = bias[m0 + r] (0 where there is no bias / past Cout_gemm)
//   SPADE: par[r] = bias[m0 + r] (interleaved gamma|beta rows), par[TM + c] = mean, par[TM + TM/2 + c] = rstd of
//          output channel (m0 >> 1) + c
// Call it BEFORE the first LDS-DMA load is issued (a ds_write issued while DMA loads were in flight left
// DMA'd rows corrupted, tools/stress_conv.py) and before a barrier that precedes the epilogue (every K loop has one).
template <int TM, int EPI, int NTHREADS>
__device__ __forceinline__ void conv_stage_params(const ConvK& d, int m0, float* par, int tid)
{
    static_assert(TM <= NTHREADS && (TM & (TM - 1)) == 0, "one pass, every wave takes part");
    {
        const int r = tid & (TM - 1);                // waves past TM rewrite the same values: all waves run the same code
        par[r] = (d.bias && m0 + r < d.Cout_gemm) ? d.bias[m0 + r] : 0.f;
    }
    if constexpr (EPI == MG_EPI_SPADE) {
        {
            const int c = tid & (TM / 2 - 1);
            const int oc = (m0 >> 1) + c;
            par[TM + c] = oc < d.Cout ? d.mean[oc] : 0.f;
            par[TM + TM / 2 + c] = oc < d.Cout ? d.rstd[oc] : 0.f;
        }
    }
}

// branch-free activation for the batched epilogue: NONE / RELU / LRELU are all  max(v, 0) + neg * min(v, 0)  with
// neg = 1 / 0 / slope (TANH takes the generic path).  It has to be arithmetic: written as a select on a runtime `relu`
// flag the compiler emitted two uniform branches PER ELEMENT (128 elements per lane in the big halo tile), and the
// epilogue of a 128-channel 3x3 conv then took 55 % as long as its whole K loop (tools/probe_halo.py stamps,
// profiles/r02_halo_probe.txt).
__device__ __forceinline__ float mg_act_fast(float v, float neg, bool /*relu*/)
{
    return fmaxf(v, 0.f) + neg * fminf(v, 0.f);
}

// Wide bf16 stores.  In the MFMA accumulator layout lane l (pixel l & 31, hi = l >> 5) holds channel quads
// base + rq*8 + hi*4: an 8-byte store per quad makes every store instruction drop 16-byte pieces (lanes l, l+32) at the
// pixel pitch, and rocprofv3 WRITE_SIZE showed 1.45x (plain) to 2.4x (SPADE, two output arrays) of the algorithmic
// bytes leaving the L2 for such kernels while fully coalesced kernels write exactly their bytes.  Exchanging quad
// rq = 2k+1 of the low half-wave with quad 2k of the high half-wave (v_permlane32_swap, gfx950) gives each lane the 8
// consecutive channels base + k*16 + hi*8: one 16-byte store per lane, 32 contiguous bytes per pixel per instruction.
// Must be executed by all 64 lanes (no divergent predicate around it).
__device__ __forceinline__ uint2 mg_pack_bf16x4(f32x4_t v) { uint2 u; u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]); return u; }
__device__ __forceinline__ uint4 mg_pair_swap(uint2 even, uint2 odd)
{
    const auto a = __builtin_amdgcn_permlane32_swap(even.x, odd.x, false, false);   // a[0]: lo keeps even, hi gets lo's odd; a[1]: lo gets hi's even, hi keeps odd
    const auto b = __builtin_amdgcn_permlane32_swap(even.y, odd.y, false, false);
    uint4 r; r.x = a[0]; r.y = b[0]; r.z = a[1]; r.w = b[1];
    return r;
}

// Batched epilogue (the common case: channel counts that are multiples of 4, no tanh).  Per-channel parameters
// come from LDS (conv_stage_params); the only global loads left are the residual / SPADE x quads, issued as a
// batch before their first use, so a workgroup pays one memory round trip per batch instead of one per
// 4-channel group (the generic code below waits after every load; at 16-32 groups per lane that was longer than
// the whole K loop of the 128-channel SPADE convs).  Addresses of non-existing pixels / channels are clamped
// to a valid element and only the stores are predicated.  `wrow` = first GEMM row of the wave inside the tile.
// Lean epilogue arithmetic.  An epilogue wave shares its SIMD with another workgroup's MFMA stream and gets roughly one VALU
// issue per 10 cycles there (s_memrealtime stamps: a 16-value block of ~150 VALU took 0.78 us, profiles/r02_halo_probe.txt), so
// the epilogue is priced in VALU instructions: the lean paths below use packed fp32 math (v_pk_add/mul/fma_f32 on register
// pairs) and one specialised body per uniform case instead of per-element selects.  ACT: 0 none, 1 relu, 2 lrelu with slope in [0, 1].
// MG_EPI_SCALAR=1 (A/B build, tools/ab_epi_scalar.sh): the same lean bodies on scalar v_fma_f32 / v_mul_f32 / v_max_f32 instead of the
// packed forms -- /opt/skills/guides/MI355X_MICROARCH.md prices a v_pk_* beside an MFMA stream at +22...26 cycles over two scalar FMAs.
// Same operation order and roundings either way (a + b == fma(a, 1, b)).
#ifndef MG_EPI_SCALAR
#define MG_EPI_SCALAR 0
#endif
#if MG_EPI_SCALAR
struct mg_pk2 {
    float v[2];
    __device__ __forceinline__ float& operator[](int i) { return v[i]; }
    __device__ __forceinline__ float operator[](int i) const { return v[i]; }
};
__device__ __forceinline__ float mg_opaque(float x) { asm volatile("" : "+v"(x)); return x; }    // keeps the SLP vectoriser from re-packing the pair
__device__ __forceinline__ mg_pk2 operator+(mg_pk2 a, mg_pk2 b) { return {{__builtin_fmaf(a[0], 1.f, b[0]), mg_opaque(__builtin_fmaf(a[1], 1.f, b[1]))}}; }
__device__ __forceinline__ mg_pk2 operator-(mg_pk2 a, mg_pk2 b) { return {{__builtin_fmaf(b[0], -1.f, a[0]), mg_opaque(__builtin_fmaf(b[1], -1.f, a[1]))}}; }
__device__ __forceinline__ mg_pk2 operator*(mg_pk2 a, mg_pk2 b) { return {{a[0] * b[0], mg_opaque(a[1] * b[1])}}; }
__device__ __forceinline__ mg_pk2 operator*(mg_pk2 a, float b) { return {{a[0] * b, mg_opaque(a[1] * b)}}; }
__device__ __forceinline__ mg_pk2& operator+=(mg_pk2& a, mg_pk2 b) { a = a + b; return a; }
__device__ __forceinline__ mg_pk2 mg_fma2(mg_pk2 a, mg_pk2 b, mg_pk2 c) { return {{__builtin_fmaf(a[0], b[0], c[0]), mg_opaque(__builtin_fmaf(a[1], b[1], c[1]))}}; }
__device__ __forceinline__ mg_pk2 mg_max2(mg_pk2 a, mg_pk2 b) { return {{fmaxf(a[0], b[0]), mg_opaque(fmaxf(a[1], b[1]))}}; }
__device__ __forceinline__ mg_pk2 mg_pk(float a, float b) { return {{a, b}}; }
#else
typedef mg_f32x2_t mg_pk2;
__device__ __forceinline__ mg_pk2 mg_fma2(mg_pk2 a, mg_pk2 b, mg_pk2 c) { return a * b + c; }
__device__ __forceinline__ mg_pk2 mg_max2(mg_pk2 a, mg_pk2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ mg_pk2 mg_pk(float a, float b) { const mg_pk2 v = {a, b}; return v; }
#endif
template <int ACT>
__device__ __forceinline__ mg_pk2 mg_act2(mg_pk2 v, float neg)
{
    if constexpr (ACT == 0) return v;
    else if constexpr (ACT == 1) { const mg_pk2 z = mg_pk(0.f, 0.f); return mg_max2(v, z); }
    else return mg_max2(v, v * neg);           // v > 0 ? v : v * slope   for 0 <= slope <= 1
}
__device__ __forceinline__ uint2 mg_pack_bf16x2x2(mg_pk2 a, mg_pk2 b) { uint2 u; u.x = f2bf2(a[0], a[1]); u.y = f2bf2(b[0], b[1]); return u; }

struct EpiNoMark { __device__ __forceinline__ void operator()(int) const {} };      // measurement hook of the probe builds (mg_conv_halo.hip)

// `xpre` (SPADE, bf16): the lane's 2*NT*2 x quads as raw 8-byte loads [half][nt][q], fetched by the caller BEFORE its main
// loop (mg_conv_halo.hip) -- or nullptr: load them here.
template <typename T, int MT, int NT, int EPI, int TM, typename PixMap, typename Mark = EpiNoMark>
__device__ __forceinline__ void conv_epilogue_fast(const ConvK& d, f32x16_t (&acc)[MT][NT], int m0, PixMap&& pixmap,
                                                   int wm, int wn, int l31, int hi, const float* par, Mark&& mark,
                                                   const uint2 (&xpre)[2 * NT * 2], bool have_xpre)
{
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
    const float neg = d.act == MG_ACT_NONE ? 1.f : (d.act == MG_ACT_RELU ? 0.f : d.slope);
    const bool relu = d.act == MG_ACT_RELU;
    size_t opix[NT];
    unsigned xoff[NT];                 // SPADE: element offset of the pixel's x row (the half-resolution source when d.x_up)
    bool pok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        size_t upix = 0;
        pok[nt] = pixmap(wn * NT * 32 + nt * 32 + l31, opix[nt], upix) && !(d.wide & 4);      // bit 2: measurement builds, stores off
        if constexpr (EPI == MG_EPI_SPADE) xoff[nt] = pok[nt] ? (unsigned)((d.x_up ? upix : opix[nt]) * d.Cout) : 0u;
        opix[nt] = pok[nt] ? opix[nt] * d.Cout : 0;
    }

    mark(0);
    if constexpr (EPI == MG_EPI_PLAIN) {
        const T* __restrict__ Res = reinterpret_cast<const T*>(d.resid);
        const T* __restrict__ Msk = reinterpret_cast<const T*>(d.x);          // optional ReLU-output mask (dgrad)
        static_for<0, MT>([&](auto mt_) {
            constexpr int mt = decltype(mt_)::value;
            mark(1 + mt * 5);
            const int lr = wm * MT * 32 + mt * 32 + hi * 4;       // + rq*8: this lane's GEMM rows inside the tile
            f32x4_t bias4[4];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) bias4[rq] = *reinterpret_cast<const f32x4_t*>(par + lr + rq * 8);
            // residual / ReLU-mask quads of ALL column tiles of this row tile are requested before the first use (one memory round
            // trip per 32 rows instead of one per 32x32 tile).  A launch has a residual (forward) or a mask (data gradient), not
            // both; the rare both-case keeps the mask loads at their use.
            const T* __restrict__ Aux = Res ? Res : Msk;
            f32x4_t aux[NT][4];
            if (Aux) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = m0 + lr + rq * 8;
                        aux[nt][rq] = ET<T>::load4(Aux + opix[nt] + (co < d.Cout ? co : 0));
                    }
            }
            const bool both = Res && Msk;
            if constexpr (sizeof(T) == 2) {
                // lean bodies (bf16, wide stores): {no aux} x {none, relu, lrelu}, {residual, mask} x {none}; anything else below
                const int ak = both ? 3 : (Res ? 1 : (Msk ? 2 : 0));
                const int ck = (d.act == MG_ACT_LRELU && !(d.slope >= 0.f && d.slope <= 1.f)) ? 3 : d.act;
                if ((d.wide & 1) && ((ak == 0 && ck <= 2) || (ak <= 2 && ck == 0))) {
                    auto body = [&](auto act_, auto aux_) {
                        constexpr int ACT = decltype(act_)::value, AUX = decltype(aux_)::value;
                        static_for<0, NT>([&](auto nt_) {
                            constexpr int nt = decltype(nt_)::value;
                            mark(2 + mt * 5 + nt);
                            mg_pk2 v[8];
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    mg_pk2 t = mg_pk(acc[mt][nt][rq * 4 + 2 * h2], acc[mt][nt][rq * 4 + 2 * h2 + 1])
                                                 + mg_pk(bias4[rq][2 * h2], bias4[rq][2 * h2 + 1]);
                                    if constexpr (AUX == 1) t += mg_pk(aux[nt][rq][2 * h2], aux[nt][rq][2 * h2 + 1]);
                                    t = mg_act2<ACT>(t, neg);
                                    if constexpr (AUX == 2) {
                                        const mg_pk2 tm = t * d.mslope;
                                        t[0] = aux[nt][rq][2 * h2] > 0.f ? t[0] : tm[0];
                                        t[1] = aux[nt][rq][2 * h2 + 1] > 0.f ? t[1] : tm[1];
                                    }
                                    v[rq * 2 + h2] = t;
                                }
#pragma unroll
                            for (int k2 = 0; k2 < 2; ++k2) {
                                const uint4 w = mg_pair_swap(mg_pack_bf16x2x2(v[4 * k2], v[4 * k2 + 1]), mg_pack_bf16x2x2(v[4 * k2 + 2], v[4 * k2 + 3]));
                                const int co = m0 + wm * MT * 32 + mt * 32 + k2 * 16 + hi * 8;
                                if (pok[nt] && co < d.Cout) *reinterpret_cast<uint4*>(Out + opix[nt] + co) = w;
                            }
                        });
                    };
                    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                    if (ak == 1) body(I0{}, I1{});
                    else if (ak == 2) body(I0{}, I2{});
                    else if (ck == 0) body(I0{}, I0{});
                    else if (ck == 1) body(I1{}, I0{});
                    else body(I2{}, I0{});
                    return;
                }
            }
            static_for<0, NT>([&](auto nt_) {
                constexpr int nt = decltype(nt_)::value;
                mark(2 + mt * 5 + nt);
                f32x4_t rv[4], mk[4];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) { rv[rq] = aux[nt][rq]; mk[rq] = aux[nt][rq]; }
                if (both) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = m0 + lr + rq * 8;
                        mk[rq] = ET<T>::load4(Msk + opix[nt] + (co < d.Cout ? co : 0));
                    }
                }
                f32x4_t v[4];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = acc[mt][nt][rq * 4 + j] + bias4[rq][j];
                        if (Res) t += rv[rq][j];
                        t = mg_act_fast(t, neg, relu);
                        v[rq][j] = (Msk && !(mk[rq][j] > 0.f)) ? t * d.mslope : t;
                    }
                }
                bool wide = false;
                if constexpr (sizeof(T) == 2) wide = (d.wide & 1) != 0;
                if (wide) {
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            const uint4 w = mg_pair_swap(mg_pack_bf16x4(v[2 * k2]), mg_pack_bf16x4(v[2 * k2 + 1]));
                            const int co = m0 + wm * MT * 32 + mt * 32 + k2 * 16 + hi * 8;
                            if (pok[nt] && co < d.Cout) *reinterpret_cast<uint4*>(Out + opix[nt] + co) = w;
                        }
                    }
                } else {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = m0 + lr + rq * 8;
                        if (pok[nt] && co < d.Cout) ET<T>::store4(Out + opix[nt] + co, v[rq]);
                    }
                }
            });
        });
    } else {
        // SPADE: acc[0] = gamma rows, acc[1] = beta rows of the same 32 output channels; two batches of 2 groups
        const T* __restrict__ X = reinterpret_cast<const T*>(d.x);
        T* __restrict__ G1 = reinterpret_cast<T*>(d.gamma_out);
        const int lrow = wm * 64;                    // first GEMM row of this wave's [gamma|beta] block in the tile
        // all x quads of both halves are requested before the first one is used: ONE memory round trip per tile instead of two
        // (the main loop's operand-address and fragment registers are dead here, so the 16 quads fit; on the K = 1152 layers the
        // epilogue is 18-22 % of the launch, profiles/r02_halo_ring_ab.txt)
        f32x4_t xall[2][NT][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int occ = ((m0 + lrow) >> 1) + (hh * 2 + q) * 8 + hi * 4;
                    if constexpr (sizeof(T) == 2) {
                        if (have_xpre) {
                            const uint2 r = xpre[(hh * NT + nt) * 2 + q];
                            f32x4_t v = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
                            xall[hh][nt][q] = v;
                            continue;
                        }
                    }
                    xall[hh][nt][q] = ET<T>::load4(X + xoff[nt] + (occ < d.Cout ? occ : 0));
                }
        static_for<0, 2>([&](auto h_) {
            constexpr int h = decltype(h_)::value;
            int oc[2];
            f32x4_t xv[NT][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) oc[q] = ((m0 + lrow) >> 1) + (h * 2 + q) * 8 + hi * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 2; ++q) xv[nt][q] = xall[h][nt][q];
            bool wide = false;
            if constexpr (sizeof(T) == 2) wide = (d.wide & 1) != 0;
            if (wide) {
                if constexpr (sizeof(T) == 2) {
                    f32x4_t bg[2], bb[2], mean4[2], rstd4[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int sub = (h * 2 + q) * 8 + hi * 4;
                        bg[q] = *reinterpret_cast<const f32x4_t*>(par + lrow + sub);
                        bb[q] = *reinterpret_cast<const f32x4_t*>(par + lrow + 32 + sub);
                        mean4[q] = *reinterpret_cast<const f32x4_t*>(par + TM + (lrow >> 1) + sub);
                        rstd4[q] = *reinterpret_cast<const f32x4_t*>(par + TM + TM / 2 + (lrow >> 1) + sub);
                    }
                    const int ocw = ((m0 + lrow) >> 1) + h * 16 + hi * 8;          // this lane's 8 consecutive channels after the exchange
                    const int ck = (d.act == MG_ACT_LRELU && !(d.slope >= 0.f && d.slope <= 1.f)) ? 3 : d.act;
                    // lean bodies: packed fp32 math, one per activation (see mg_act2); same operation order as the generic body below
                    auto body = [&](auto act_) {
                        constexpr int ACT = decltype(act_)::value;
                        static_for<0, NT>([&](auto nt_) {
                            constexpr int nt = decltype(nt_)::value;
                            mg_pk2 g[2][2], hv[2][2];
                            const mg_pk2 one = mg_pk(1.f, 1.f);
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const int e = (h * 2 + q) * 4 + 2 * h2;
                                    g[q][h2] = (one + mg_pk(acc[0][nt][e], acc[0][nt][e + 1])) + mg_pk(bg[q][2 * h2], bg[q][2 * h2 + 1]);
                                    const mg_pk2 bt = mg_pk(acc[1][nt][e], acc[1][nt][e + 1]) + mg_pk(bb[q][2 * h2], bb[q][2 * h2 + 1]);
                                    const mg_pk2 xh = (mg_pk(xv[nt][q][2 * h2], xv[nt][q][2 * h2 + 1]) - mg_pk(mean4[q][2 * h2], mean4[q][2 * h2 + 1]))
                                                        * mg_pk(rstd4[q][2 * h2], rstd4[q][2 * h2 + 1]);
                                    hv[q][h2] = mg_act2<ACT>(mg_fma2(xh, g[q][h2], bt), neg);
                                }
                            const uint4 hw = mg_pair_swap(mg_pack_bf16x2x2(hv[0][0], hv[0][1]), mg_pack_bf16x2x2(hv[1][0], hv[1][1]));
                            uint4 gw = hw;
                            if (G1) gw = mg_pair_swap(mg_pack_bf16x2x2(g[0][0], g[0][1]), mg_pack_bf16x2x2(g[1][0], g[1][1]));
                            if (pok[nt] && ocw < d.Cout) {
                                const size_t o = opix[nt] + ocw;
                                *reinterpret_cast<uint4*>(Out + o) = hw;
                                if (G1) *reinterpret_cast<uint4*>(G1 + o) = gw;
                            }
                        });
                    };
                    if (ck == 0) body(std::integral_constant<int, 0>{});
                    else if (ck == 1) body(std::integral_constant<int, 1>{});
                    else if (ck == 2) body(std::integral_constant<int, 2>{});
                    else
                    static_for<0, NT>([&](auto nt_) {
                        constexpr int nt = decltype(nt_)::value;
                        f32x4_t g[2], hv[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int rq = h * 2 + q;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                g[q][j] = 1.f + acc[0][nt][rq * 4 + j] + bg[q][j];
                                const float bt = acc[1][nt][rq * 4 + j] + bb[q][j];
                                const float xh = (xv[nt][q][j] - mean4[q][j]) * rstd4[q][j];
                                hv[q][j] = mg_act_fast(xh * g[q][j] + bt, neg, relu);
                            }
                        }
                        const uint4 hw = mg_pair_swap(mg_pack_bf16x4(hv[0]), mg_pack_bf16x4(hv[1]));
                        uint4 gw = hw;
                        if (G1) gw = mg_pair_swap(mg_pack_bf16x4(g[0]), mg_pack_bf16x4(g[1]));     // G1 is uniform: no divergence around the exchange
                        if (pok[nt] && ocw < d.Cout) {
                            const size_t o = opix[nt] + ocw;
                            *reinterpret_cast<uint4*>(Out + o) = hw;
                            if (G1) *reinterpret_cast<uint4*>(G1 + o) = gw;
                        }
                    });
                }
            } else
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sub = (h * 2 + q) * 8 + hi * 4;
                const f32x4_t bg = *reinterpret_cast<const f32x4_t*>(par + lrow + sub);
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(par + lrow + 32 + sub);
                const f32x4_t mean4 = *reinterpret_cast<const f32x4_t*>(par + TM + (lrow >> 1) + sub);
                const f32x4_t rstd4 = *reinterpret_cast<const f32x4_t*>(par + TM + TM / 2 + (lrow >> 1) + sub);
                static_for<0, NT>([&](auto nt_) {
                    constexpr int nt = decltype(nt_)::value;
                    const int rq = h * 2 + q;
                    f32x4_t g, hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        g[j] = 1.f + acc[0][nt][rq * 4 + j] + bg[j];
                        const float bt = acc[1][nt][rq * 4 + j] + bb[j];
                        const float xh = (xv[nt][q][j] - mean4[j]) * rstd4[j];
                        hv[j] = mg_act_fast(xh * g[j] + bt, neg, relu);
                    }
                    if (pok[nt] && oc[q] < d.Cout) {
                        const size_t o = opix[nt] + oc[q];
                        ET<T>::store4(Out + o, hv);
                        if (G1) ET<T>::store4(G1 + o, g);
                    }
                });
            }
        });
    }
}

// `pixmap(p, opix, upix)`: p = pixel index inside the workgroup's pixel tile (0 .. TN-1) -> false if the pixel does
// not exist, else opix = flat output pixel index ((n*Hout + oy)*Wout + ox) and upix = the index of pixel (oy >> 1, ox >> 1)
// in a half-resolution [N][Hout/2][Wout/2] tensor (used when d.x_up).
template <typename T, int MT, int NT, int EPI, int TM, typename PixMap, typename Mark = EpiNoMark>
__device__ __forceinline__ void conv_epilogue(const ConvK& d, f32x16_t (&acc)[MT][NT], int m0, PixMap&& pixmap,
                                              int wm, int wn, int l31, int hi, const float* par, Mark&& mark,
                                              const uint2 (&xpre)[2 * NT * 2], bool have_xpre)
{
    if (((d.Cout | d.Cout_gemm) & 3) == 0 && d.act != MG_ACT_TANH) {
        conv_epilogue_fast<T, MT, NT, EPI, TM>(d, acc, m0, pixmap, wm, wn, l31, hi, par, mark, xpre, have_xpre);
        return;
    }
    // ---- epilogue -----------------------------------------------------------
    T* __restrict__ Out = reinterpret_cast<T*>(d.out);
    // compile-time tile indices (static_for): runtime-indexed accumulator arrays would be demoted to scratch
    static_for<0, NT>([&](auto nt_) {
        constexpr int nt = decltype(nt_)::value;
        size_t opix, upix = 0;
        if (!pixmap(wn * NT * 32 + nt * 32 + l31, opix, upix)) return;
        const size_t xo = (d.x_up ? upix : opix) * d.Cout;     // SPADE: where this pixel's x row starts

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
                    const T* __restrict__ Msk = reinterpret_cast<const T*>(d.x);
                    if ((d.Cout & 3) == 0) {
                        if (Res) { f32x4_t rv = ET<T>::load4(Res + o);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += rv[j]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = mg_act(v[j], d.act, d.slope);
                        if (Msk) { const f32x4_t mk = ET<T>::load4(Msk + o);
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (!(mk[j] > 0.f)) v[j] *= d.mslope; }
                        ET<T>::store4(Out + o, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (co + j < d.Cout) {
                                float s = v[j];
                                if (Res) s += ET<T>::load1(Res + o + j);
                                s = mg_act(s, d.act, d.slope);
                                if (Msk && !(ET<T>::load1(Msk + o + j) > 0.f)) s *= d.mslope;
                                ET<T>::store1(Out + o + j, s);
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
                    const f32x4_t xv = ET<T>::load4(X + xo + oc);
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
                            const float xh = (ET<T>::load1(X + xo + oc + j) - d.mean[oc + j]) * d.rstd[oc + j];
                            ET<T>::store1(Out + o + j, mg_act(xh * g[j] + bt[j], d.act, d.slope));
                            if (G1) ET<T>::store1(G1 + o + j, g[j]);
                        }
                    }
                }
            });
        }
    });
}

template <typename T, int MT, int NT, int EPI, int TM, typename PixMap>
__device__ __forceinline__ void conv_epilogue(const ConvK& d, f32x16_t (&acc)[MT][NT], int m0, PixMap&& pixmap,
                                              int wm, int wn, int l31, int hi, const float* par)
{
    const uint2 none[2 * NT * 2] = {};
    conv_epilogue<T, MT, NT, EPI, TM>(d, acc, m0, pixmap, wm, wn, l31, hi, par, EpiNoMark(), none, false);
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

// global_load_lds with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst)      // 4 bytes per lane, lane i -> lds_dst + 4*i
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// conv_stage_params for the LDS-DMA kernels: the same 2*TM-float image, fetched with global_load_lds_dword so
// that the kernel contains no ds_write at all.  (With a ds_write of these parameters in the prologue the halo
// kernel produced rare wrong tiles -- tools/stress_conv.py, tests/test_gpu_kernels.py::test_conv_race_screen --
// although every DMA'd byte was covered by a vmcnt wait and a barrier; without DS stores it is clean.)
// Issue it before the prologue's operand loads: being the oldest loads of the wave, every later counted vmcnt
// wait covers them.  `par_lds` is the LDS byte address of the image.
template <int TM, int EPI, int NW>
__device__ __forceinline__ void conv_stage_params_dma(const ConvK& d, int m0, unsigned par_lds, int wave, int lane)
{
    constexpr int NI = (EPI == MG_EPI_SPADE ? 2 * TM : TM) / 64;      // wave-instructions of 64 floats
    const unsigned char* const z = g_mg_zeros + lane * 4;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        if (k % NW != wave) continue;
        const int i = k * 64 + lane;
        const void* src = z;
        if (i < TM) {
            if (d.bias && m0 + i < d.Cout_gemm) src = d.bias + m0 + i;
        } else {
            const int c = (i - TM) % (TM / 2), oc = (m0 >> 1) + c;
            if (oc < d.Cout) src = ((i - TM) < TM / 2 ? d.mean : d.rstd) + oc;
        }
        glds4(src, __builtin_amdgcn_readfirstlane(par_lds + k * 256));
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


// linear pixel tile of the tap-list kernels: q = q0 + p over N*Hj*Wj, strided/offset output grid
struct LinearPixMap {
    const ConvK& d; int q0, HWj;
    __device__ __forceinline__ bool operator()(int p, size_t& opix, size_t& upix) const {
        const int q = q0 + p;
        if (q >= d.ngemm) return false;
        const int n = q / HWj, r = q - n * HWj;
        const int jy = r / d.Wj, jx = r - jy * d.Wj;
        const int oy = jy * d.osy + d.ooy, ox = jx * d.osx + d.oox;
        opix = (size_t)((n * d.Hout + oy) * d.Wout + ox);
        upix = (size_t)((n * (d.Hout >> 1) + (oy >> 1)) * (d.Wout >> 1) + (ox >> 1));
        return true;
    }
};

}  // namespace

int launch_conv_halo(ConvK& k, int dtype, int epilogue, hipStream_t st);   // mg_conv_halo.hip
bool conv_halo64_applies(const ConvK& k, int dtype, int epilogue);          // mg_conv_halo64.hip: 3x3 over exactly 64 input channels, weights in registers
int launch_conv_halo64(ConvK& k, hipStream_t st);
bool conv_thin_applies(const ConvK& k, int dtype, int epilogue);            // mg_conv_thin.hip
int launch_conv_thin(ConvK& k, hipStream_t st);
bool conv_thin_taps_applies(const ConvK& k, int dtype, int epilogue);       // mg_conv_thin.hip: any <= 7x7 window, stride 1 | 2
int launch_conv_thin_taps(ConvK& k, hipStream_t st);
bool conv_dot_applies(const ConvK& k, int dtype, int epilogue);             // mg_conv_dot.hip
int launch_conv_dot(ConvK& k, int dtype, hipStream_t st);
bool conv_fewout_applies(const ConvK& k, int dtype, int epilogue);          // mg_conv_dot.hip: <= 16 GEMM rows over 64 input channels
int launch_conv_fewout(ConvK& k, hipStream_t st);
