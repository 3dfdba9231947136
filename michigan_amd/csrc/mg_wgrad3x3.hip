// mg_wgrad3x3.hip -- weight gradient of the 3x3 / stride 1 / pad 1 convolutions (bf16), the bulk of the
// generator's wgrad time.  Same GEMM view as mg_wgrad.hip (rows = co, cols = ci, K = output pixels), but
//
//   * one workgroup owns a KERNEL ROW (ky) and accumulates its three taps (kx = 0,1,2) at once: the
//     dY strip is staged once and the X strip once (+1 pixel of halo each side) for three taps, so the
//     bytes staged per FLOP are a third of the tap-per-workgroup kernel's;
//   * operands go global -> LDS with global_load_lds_dwordx4 into a 4-stage ring (no staging VGPRs, no
//     ds_write: the generic kernel spends more LDS cycles writing its tiles than reading them);
//   * LDS rows are pixel-major and unpadded ([pixel][TM or TN channels]); the 64-byte blocks of a row are
//     XOR-swizzled with the row number so that the four rows a ds_read_b64_tr_b16 touches cover all 64
//     banks, whatever the tap shift.  The swizzle is applied on the SOURCE piece each lane fetches.
//
// A stage is 32 consecutive output pixels (one 32-pixel row segment, or two 16-pixel rows when W == 16)
// = two MFMA K-steps; its X strip holds RPS x (SEG + 2) pixels of image row y + ky - 1.
// Partial sums go to the fp32 dW with hardware float atomics (split-K over pixel ranges).
#include "mg_conv_common.h"
#include "mg_wgrad_common.h"

extern int g_mg_wgrad3x3_probe;    // mg_conv.hip, mg_set_option(12, v): 1 = launch the stamped build of wgrad3x3_kernel<2, 2>
int g_mg_wgrad3x3_stripe = 64;     // mg_set_option(24, v): pixel width of the column stripes the stages walk (0 = plain raster order over whole image rows)

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;

// Measurement build (mg_set_option(12, 1) + the stamp buffer of mg_set_option(13 / 14)): s_memrealtime stamps around the vmcnt wait,
// the barrier, the DMA issue (address arithmetic included) and the MFMA block of every stage; tools/probe_wgrad3x3.py.
__device__ unsigned long long* g_mg_wg3_probe_out = nullptr;
__device__ __forceinline__ unsigned long long wg3_stamp()
{
    unsigned long long v;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory");
    return v;
}

template <int MT, int NT, bool W16, bool PROBE = false>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const Wg3K d)
{
    constexpr int TM = 64 * MT, TN = 64 * NT;
    constexpr int RBA = TM * 2, RBB = TN * 2;                 // bytes per LDS row (one pixel)
    constexpr int PPA = RBA / 16, PPB = RBB / 16;             // 16-byte pieces per row
    constexpr int A_ROWS = 32, B_ROWS = (NT == 2) ? 36 : 40;  // B: whole 1 KiB wave blocks
    constexpr int A_BYTES = A_ROWS * RBA, B_BYTES = B_ROWS * RBB;
    constexpr int A_IPS = A_BYTES / 4096;                     // wave-instructions per stage per wave
    constexpr int B_IPS = (B_BYTES + 4095) / 4096;
    constexpr int IPS = A_IPS + B_IPS;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NS = 4;
    static_assert(A_BYTES % 4096 == 0 && B_BYTES % 1024 == 0, "stage must be whole wave blocks");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NS stages + 4 x 1 KiB dump blocks

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
    const int tn = tile % d.tiles_n;  tile /= d.tiles_n;
    const int ky = tile % 3;
    const int split = tile / 3;
    const int m0 = tm * TM, n0 = tn * TN;
    const int sbeg = split * d.sps;
    const int send = min(d.nstg, sbeg + d.sps);
    const int nk = send - sbeg;
    if (nk <= 0) return;

    const int H = d.H, W = d.W;
    constexpr int SEG = W16 ? 16 : 32, SEGP = SEG + 2;
    constexpr int PR = (32 / SEG) * SEGP;                     // live rows of the X strip (34 or 36)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned dump = lds0 + NS * STAGE + wave * 1024;
    const unsigned char* zsrc = g_mg_zeros + (lane & 3) * 16;

    auto swz_a = [](int row) { return RBA == 256 ? (row & 3) : ((row >> 1) & 1); };   // in 64-byte blocks
    auto swz_b = [](int row) { return RBB == 256 ? (row & 3) : ((row >> 1) & 1); };

    // Addresses = wave-uniform stage base (walked, scalar registers) + a per-thread 32-bit byte offset that never
    // changes: only the offsets and the strip coordinates stay in VGPRs (the accumulators need 192 of the 256).
    // Channels past Cg / Cin (ragged last tile) fetch the tile's first channel instead: they only feed dW rows /
    // columns that the epilogue never writes.  Out-of-image X pixels must be zeros and fetch the zero block.
    int aoff[A_IPS];
#pragma unroll
    for (int j = 0; j < A_IPS; ++j) {
        const int g = j * 256 + tid, row = g / PPA, slot = g % PPA;
        int ch = m0 + ((slot ^ (swz_a(row) << 2)) << 3);
        if (ch >= d.Cg) ch = m0;
        aoff[j] = (row * d.Cg + ch) * 2;
    }
    int boff[B_IPS], brc[B_IPS];                              // brc = r | (c + 1) << 4, or -1 for a dead row
#pragma unroll
    for (int j = 0; j < B_IPS; ++j) {
        const int g = j * 256 + tid, row = g / PPB, slot = g % PPB;
        int ch = n0 + ((slot ^ (swz_b(row) << 2)) << 3);
        if (ch >= d.Cin) ch = n0;
        const int r = (W16 && row >= SEGP) ? 1 : 0, c = row - r * SEGP - 1;
        boff[j] = (((r + ky - 1) * W + c) * d.Cin + ch) * 2;
        brc[j] = row < PR ? (r | (c + 1) << 4) : -1;
    }
    // Stage order.  W16: two 16-pixel rows per stage, raster order.  Otherwise an image is walked in COLUMN STRIPES of d.stripe_w pixels, each
    // stripe top to bottom: the three kernel-row workgroups of a pixel range read X rows y - 1, y, y + 1 for output row y, i.e. every X row
    // three times, one image row apart -- with whole-width rows (16 stages at W = 512, ~17 MB of other traffic through the XCD's 4 MiB L2 in
    // between) every re-read went back to the fabric (TCC hit rate 52 %, 1.5x the algorithmic bytes, profiles/r05_pmc_halo.txt); with 64-pixel
    // stripes the re-read comes two stages later and hits (round 6).  stripe_w = W is the old order.
    const unsigned char* const dy0 = reinterpret_cast<const unsigned char*>(d.dy);
    const unsigned char* const x0p = reinterpret_cast<const unsigned char*>(d.x);
    const unsigned char* abase; const unsigned char* xbase;
    int sy, sx, sg_next = sbeg;                               // image row / column of the next stage to issue
    int simg = 0, sstripe = 0;                                // (stripe order) image and stripe of the next stage
    const int SWp = W16 ? W : d.stripe_w, nsegs = SWp / 32, nstripes = W / SWp;
    if (W16) {
        abase = dy0 + (size_t)sbeg * 32 * d.Cg * 2;
        xbase = x0p + (size_t)sbeg * 32 * d.Cin * 2;
        const int rem = (sbeg * 32) % (H * W); sy = rem / W; sx = rem - sy * W;
    } else {
        const int spi = H * (W / 32);                         // stages per image
        simg = sbeg / spi;
        int r = sbeg - simg * spi;
        sstripe = r / (H * nsegs);  r -= sstripe * (H * nsegs);
        sy = r / nsegs;
        sx = sstripe * SWp + (r - sy * nsegs) * 32;
        const size_t p = ((size_t)simg * H + sy) * W + sx;
        abase = dy0 + p * d.Cg * 2;
        xbase = x0p + p * d.Cin * 2;
    }
    sy = __builtin_amdgcn_readfirstlane(sy); sx = __builtin_amdgcn_readfirstlane(sx);     // wave-uniform: the bounds tests of issue() stay scalar
    simg = __builtin_amdgcn_readfirstlane(simg); sstripe = __builtin_amdgcn_readfirstlane(sstripe);
    const int a_step = 32 * d.Cg * 2, x_step = 32 * d.Cin * 2;

    // The DMA issue sits beside the other workgroup's MFMA stream, where a wave gets roughly one VALU issue per 10 cycles: with the
    // bounds tests, selects and 64-bit adds done per lane it was 0.36 us of a 1.23 us stage (tools/probe_wgrad3x3.py).  Validity of a lane
    // only depends on its CLASS (left halo column / right halo column / strip row) and on the wave-uniform strip position, so the lane
    // classes become 64-bit ballots in scalar registers once, the per-stage test is scalar arithmetic, and a lane pays two adds and two
    // v_cndmask (scalar mask operand) per X load; the dY loads take a scalar base + a constant per-lane offset (no VALU at all).
    unsigned long long m_live[B_IPS], m_left[B_IPS], m_right[B_IPS], m_r1[B_IPS];
#pragma unroll
    for (int j = 0; j < B_IPS; ++j) {
        const bool lv = brc[j] >= 0;
        const int c = (brc[j] >> 4) - 1, r = brc[j] & 15;
        m_live[j]  = __builtin_amdgcn_ballot_w64(lv);
        m_left[j]  = __builtin_amdgcn_ballot_w64(lv && c == -1);
        m_right[j] = __builtin_amdgcn_ballot_w64(lv && c == SEG);
        m_r1[j]    = __builtin_amdgcn_ballot_w64(lv && r == 1);
    }
    const unsigned long long zaddr = (unsigned long long)(size_t)zsrc;
    const unsigned zlo = (unsigned)zaddr, zhi = (unsigned)(zaddr >> 32);

    auto issue = [&](int stage) {
        const bool real = sg_next < send;
        const unsigned sbase = lds0 + stage * STAGE;
        if (real) {
#pragma unroll
            for (int j = 0; j < A_IPS; ++j)
                glds16_s(abase, (unsigned)aoff[j], __builtin_amdgcn_readfirstlane(sbase + j * 4096 + wave * 1024));
        } else {
#pragma unroll
            for (int j = 0; j < A_IPS; ++j)
                glds16(zsrc, __builtin_amdgcn_readfirstlane(sbase + j * 4096 + wave * 1024));
        }
        const int sxs = __builtin_amdgcn_readfirstlane(sx), sys = __builtin_amdgcn_readfirstlane(sy), kys = __builtin_amdgcn_readfirstlane(ky);
        const bool lok = sxs > 0, rok = sxs + SEG < W;
        const bool y0 = (unsigned)(sys + kys - 1) < (unsigned)H, y1 = (unsigned)(sys + kys) < (unsigned)H;
#pragma unroll
        for (int j = 0; j < B_IPS; ++j) {
            unsigned long long v = real ? m_live[j] : 0ull;
            if (!lok) v &= ~m_left[j];
            if (!rok) v &= ~m_right[j];
            if (W16) { if (!y0) v &= m_r1[j]; if (!y1) v &= ~m_r1[j]; }
            else if (!y0) v = 0ull;
            v = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);   // a scalar register pair for the mask operand
            asm("" : "+s"(v));                                            // ... and never an immediate
            const unsigned long long pa = (unsigned long long)(size_t)(xbase + boff[j]);
            unsigned plo, phi;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(plo) : "v"(zlo), "v"((unsigned)pa), "s"(v));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(phi) : "v"(zhi), "v"((unsigned)(pa >> 32)), "s"(v));
            const bool live = (j * 4096 + wave * 1024) < B_BYTES;          // this wave's 1 KiB block lies inside the strip
            glds16(reinterpret_cast<const void*>((size_t)(((unsigned long long)phi << 32) | plo)),
                   __builtin_amdgcn_readfirstlane(live ? sbase + A_BYTES + j * 4096 + wave * 1024 : dump));
        }
        ++sg_next;
        if (W16) {
            abase += a_step; xbase += x_step;
            sy += 2; if (sy >= H) sy = 0;
        } else {
            sx += 32;
            if (sx < (sstripe + 1) * SWp) { abase += a_step; xbase += x_step; }
            else {                                            // end of the stripe's row: next row of the stripe, next stripe, next image
                sx = sstripe * SWp; ++sy;
                if (sy >= H) { sy = 0; ++sstripe; if (sstripe >= nstripes) { sstripe = 0; ++simg; } sx = sstripe * SWp; }
                const size_t p = ((size_t)simg * H + sy) * W + sx;
                abase = dy0 + p * d.Cg * 2;
                xbase = x0p + p * d.Cin * 2;
            }
        }
    };

    f32x16_t acc[3][MT][NT];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kx][mt][nt][r] = 0.f;

    // fused bias gradient (centre kernel row, first N tile, waves wn == 0): each lane already holds 8 K values
    // of dY row co = lane & 31 in its A fragment; their running sum costs MT registers.
    const bool do_bias = d.dbias != nullptr && ky == 1 && tn == 0 && wn == 0;
    float bsum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bsum[mt] = 0.f;

    // lane geometry of the transpose reads (same as mg_wgrad.hip): 16-lane group g2 -> channel block
    // (g2 & 1) * 16, K half g2 >> 1; lane i16 -> K row i16 >> 2, channels (i16 & 3) * 4
    const int i16 = lane & 15, g2 = lane >> 4;
    const int rsub = (g2 >> 1) * 8 + (i16 >> 2);
    const int csub = ((g2 & 1) * 16 + (i16 & 3) * 4) * 2;
    constexpr int ksb = W16 ? SEGP : 16;                      // X strip rows between the two K-steps

    auto compute = [&](int slot) {
        const unsigned char* As = smem + slot * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[MT];
            const int ra = ks * 16 + rsub;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int col = ((wm * MT + mt) * 64 + csub) ^ (swz_a(ra) << 6);
                const unsigned char* p = As + ra * RBA + col;
                s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p));
                s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p + 4 * RBA));
                a[mt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                if (do_bias) {
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                    const u32x4_t w = __builtin_bit_cast(u32x4_t, a[mt]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bsum[mt] += __uint_as_float(w[j] << 16) + __uint_as_float(w[j] & 0xffff0000u);
                }
            }
            static_for<0, 3>([&](auto kx_) {
                constexpr int kx = decltype(kx_)::value;
                bf16x8_t b[NT];
                const int rb = ks * ksb + kx + rsub;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = ((wn * NT + nt) * 64 + csub) ^ (swz_b(rb) << 6);
                    const unsigned char* p = Bs + rb * RBB + col;
                    s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p));
                    s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p + 4 * RBB));
                    b[nt] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[kx][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[kx][mt][nt], 0, 0, 0);
            });
        }
    };

    // prologue: NS-1 stages in flight (stages past `send` fetch zeros so that the vmcnt bookkeeping is uniform)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    int slot = 0, islot = NS - 1;
    unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_mfma = 0, t_begin = 0;
    if constexpr (PROBE) t_begin = wg3_stamp();
    for (int it = 0; it < nk; ++it) {
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if constexpr (PROBE) s0 = wg3_stamp();
        wait_vmcnt<(NS - 2) * IPS>();
        if constexpr (PROBE) s1 = wg3_stamp();
        __builtin_amdgcn_s_barrier();
        if constexpr (PROBE) s2 = wg3_stamp();
        issue(islot);
        if constexpr (PROBE) s3 = wg3_stamp();
        compute(slot);
        if constexpr (PROBE) { const unsigned long long s4 = wg3_stamp(); t_wait += s1 - s0; t_bar += s2 - s1; t_issue += s3 - s2; t_mfma += s4 - s3; }
        slot = (slot == NS - 1) ? 0 : slot + 1;
        islot = (islot == NS - 1) ? 0 : islot + 1;
    }
    wait_vmcnt<0>();
    unsigned long long t_loop_end = 0;
    if constexpr (PROBE) t_loop_end = wg3_stamp();

    if (do_bias) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float t = bsum[mt] + __shfl_xor(bsum[mt], 32);          // the two K halves of the row
            const int co = m0 + (wm * MT + mt) * 32 + l31;
            if (hi == 0 && co < d.Cg) wg_accum(d.dbias, d.det_stride, split, (size_t)co, t);
        }
    }

    static_for<0, 3 * MT * NT>([&](auto i_) {
        constexpr int kx = decltype(i_)::value / (MT * NT), mt = (decltype(i_)::value / NT) % MT, nt = decltype(i_)::value % NT;
        const int ci = n0 + (wn * NT + nt) * 32 + l31;
        if (ci >= d.Cin) return;
        const int tap = ky * 3 + kx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < d.Cg)
                wg_accum(d.dw, d.det_stride, split, (size_t)(tap * d.Cg + co) * d.Cin + ci, acc[kx][mt][nt][r]);
        }
    });
    if constexpr (PROBE) {
        const unsigned long long t_end = wg3_stamp();
        if (lane == 0 && g_mg_wg3_probe_out) {
            unsigned long long* o = g_mg_wg3_probe_out + ((size_t)blockIdx.x * 4 + wave) * 8;
            o[0] = t_wait; o[1] = t_bar; o[2] = t_issue; o[3] = t_mfma; o[4] = t_loop_end - t_begin; o[5] = t_end - t_loop_end; o[6] = (unsigned long long)nk; o[7] = t_begin;
        }
    }
}


template <int MT, int NT, bool W16>
int launch3(Wg3K& k, hipStream_t st, int* nsplit, bool dry)
{
    constexpr int TM = 64 * MT, TN = 64 * NT;
    constexpr int STAGE = 32 * TM * 2 + ((NT == 2) ? 36 : 40) * TN * 2;
    constexpr size_t LDS = 4 * (size_t)STAGE + 4096;
    k.tiles_m = (k.Cg + TM - 1) / TM;
    k.tiles_n = (k.Cin + TN - 1) / TN;
    k.stripe_w = (!W16 && g_mg_wgrad3x3_stripe >= 32 && (g_mg_wgrad3x3_stripe % 32) == 0 && k.W > g_mg_wgrad3x3_stripe && (k.W % g_mg_wgrad3x3_stripe) == 0) ? g_mg_wgrad3x3_stripe : k.W;
    const long base = 3L * k.tiles_m * k.tiles_n;
    // Split-K: every split adds one pass of fp32 atomics over the whole dW (~1.5 TB/s), while the main loop is
    // already near its rate with ~1.5 workgroups per CU (tools/wgrad_split_sweep.py): use ~384 workgroups, more
    // (up to ~1536) only while a split keeps >= 192 stages so that the atomics stay small against it.
    int S = k.splitk;
    if (S <= 0) {
        const int lo = (int)((384 + base - 1) / base), hi = (int)((1536 + base - 1) / base);
        S = k.nstg / 192 < hi ? k.nstg / 192 : hi;
        if (S < lo) S = lo;
        if (S > k.nstg / 16) S = k.nstg / 16;                // >= 16 stages per split
        if (S < 1) S = 1;
    }
    if (S > k.nstg) S = k.nstg;
    k.sps = (k.nstg + S - 1) / S;
    S = (k.nstg + k.sps - 1) / k.sps;
    const long nblk = base * S;
    if (nsplit) *nsplit = S;
    if (dry) return MG_OK;
    if (nblk <= 0 || nblk > 0x7fffffffL) return mg_fail(MG_ERR_ARG, "mg_conv_wgrad: bad grid %ld", nblk);
    auto kern = wgrad3x3_kernel<MT, NT, W16>;
    // flags bit 1 (a launch that runs BESIDE another stream's kernels: ops.sink_wgrad): request more than half a CU's LDS, i.e. ONE workgroup of
    // this kernel per CU -- the other half of every CU (registers, LDS, wave slots) stays with the main stream.  63.9 -> 63.3 ms per step
    // on top of the side stream itself (profiles/r05_side_stream_ab.txt); in stream order it only halves the kernel's occupancy: off there.
    constexpr size_t LDS_HALF_CU = LDS > 84 * 1024 ? LDS : 84 * 1024;
    const size_t lds_req = k.half_cu ? LDS_HALF_CU : LDS;
    mg_raise_lds_cap(reinterpret_cast<const void*>(kern), (int)LDS_HALF_CU);
#if MG_PROBES
    if constexpr (MT == 2 && NT == 2 && !W16) {
        if (g_mg_wgrad3x3_probe) {
            auto pk = wgrad3x3_kernel<MT, NT, W16, true>;
            mg_raise_lds_cap(reinterpret_cast<const void*>(pk), (int)LDS);
            hipLaunchKernelGGL(pk, dim3((unsigned)nblk), dim3(256), LDS, st, k);
            MG_CHECK_LAUNCH("mg_conv_wgrad(3x3 probe)");
            return MG_OK;
        }
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_req, st, k);
    MG_CHECK_LAUNCH("mg_conv_wgrad(3x3)");
    return MG_OK;
}

}  // namespace

// Eligibility is decided by the caller (mg_wgrad.hip): bf16, the 9 taps of a 3x3 / pad 1 window in raster
// order, stride 1, Hin == Hj, Win == Wj, W == 16 or W % 32 == 0, H*W % 32 == 0, Cin >= 64 and Cg >= 64.
int wgrad3x3_set_probe(unsigned long long addr)
{
    unsigned long long* p = reinterpret_cast<unsigned long long*>(addr);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mg_wg3_probe_out), &p, sizeof(p)) == hipSuccess ? MG_OK : MG_ERR_ARG;
}

int launch_wgrad3x3(Wg3K& k, hipStream_t st, int* nsplit, bool dry)
{
    const bool m2 = k.Cg > 64, n2 = k.Cin > 64;
    if (k.W == 16) {
        if (m2 && n2) return launch3<2, 2, true>(k, st, nsplit, dry);
        if (m2)       return launch3<2, 1, true>(k, st, nsplit, dry);
        if (n2)       return launch3<1, 2, true>(k, st, nsplit, dry);
        return launch3<1, 1, true>(k, st, nsplit, dry);
    }
    if (m2 && n2) return launch3<2, 2, false>(k, st, nsplit, dry);
    if (m2)       return launch3<2, 1, false>(k, st, nsplit, dry);
    if (n2)       return launch3<1, 2, false>(k, st, nsplit, dry);
    return launch3<1, 1, false>(k, st, nsplit, dry);
}
