// mg_attention.hip -- the self-attention of the frozen orientation in-painting net (reference generator.py:467-485,
// SURVEY section 8f rank 3):   out[n][i][:] = sum_j softmax_j(q[n][i] . k[n][j]) v[n][j][:]   (no 1/sqrt(d) scale)
// over L = H*W positions (4096 at the net's 64x64 working resolution), d_qk = 64, d_v = 256.
//
// Flash-style: the [L, L] score matrix never exists.  A workgroup (4 waves) owns 128 queries, a wave 32 of them; the
// keys / values stream through LDS in tiles of KVB positions (register-staged: the next tile's global loads are issued
// before this tile's MFMAs and written to the other LDS buffer behind them -- one barrier per tile).
//
// MFMA roles are SWAPPED so that every reduction over keys is lane-local:
//   S^T = K . Q^T      A = K tile rows (from LDS), B = Q^T (registers, loaded once)   -> lane (query = lane & 31) holds 16 keys
//                      per 32-key tile in its accumulator, the other 16 in lane + 32
//   O^T = V^T . P^T    B = P^T = the exponentiated accumulators themselves (no cross-lane traffic: the K order of an MFMA is
//                      free as long as A and B agree, so the A operand is gathered in the accumulator's key order),
//                      A = V^T: bf16 through ds_read_b64_tr_b16 (V stays position-major in LDS, 64-byte blocks XOR-swizzled by
//                      the key so that a half-wave's four key rows land in four different bank quarters), fp32 through
//                      ds_read_b32 (32 lanes = 32 consecutive channels).
// Online softmax per query column: running max m (synchronised between the two half-waves that share a query), running
// partial row sum l (added across the halves at the end), O rescaled only when some lane's max moved (wave-uniform branch).
// bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulation everywhere, P enters the second product as a hi + lo pair of bf16 values;
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma, two-level sums over the keys: the parity configuration).
#include "mg_common.h"

namespace {

constexpr int DQK = 64, DV = 256, BQ = 128, NTHR_A = 256;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

template <typename T> struct AT;
template <> struct AT<uint16_t> {
    static constexpr int KVB = 64;                                    // keys per LDS tile
    static constexpr int KROW = DQK * 2, VROW = DV * 2;               // bytes per key row
    __device__ static __forceinline__ int kswz(int row) { return (row >> 1) & 7; }      // 8 pieces per row, 2 rows per bank row
};
template <> struct AT<float> {
    static constexpr int KVB = 32;
    static constexpr int KROW = DQK * 4, VROW = DV * 4;
    __device__ static __forceinline__ int kswz(int row) { return row & 15; }            // 16 pieces per row, 1 row per bank row
};

// byte offset of 16-byte piece `pc` of key row `row` inside the V tile
template <typename T> __device__ __forceinline__ int v_piece_off(int row, int pc)
{
    if constexpr (sizeof(T) == 2) return row * AT<T>::VROW + ((((pc >> 2) ^ (row & 7)) << 6) | ((pc & 3) << 4));
    else                          return row * AT<T>::VROW + (pc << 4);
}

struct AttnArgs {
    const void* q; const void* k; const void* v; void* out;
    int N, L;
    long ldq, ldk, ldv, ldo;        // row pitches in elements
};

template <typename T>
__global__ __launch_bounds__(NTHR_A, 1) void self_attention_kernel(const AttnArgs a)
{
    using A = AT<T>;
    constexpr int KVB = A::KVB, KT = KVB / 32;
    constexpr int KBYTES = KVB * A::KROW, VBYTES = KVB * A::VROW, STAGE = KBYTES + VBYTES;
    constexpr int KP = A::KROW / 16, VP = A::VROW / 16;               // 16-byte pieces per row
    constexpr int NK = KVB * KP / NTHR_A, NV = KVB * VP / NTHR_A;     // pieces per thread and tile
    constexpr bool BF = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n = blockIdx.y;
    const int L = a.L;
    const T* __restrict__ Q = reinterpret_cast<const T*>(a.q) + (size_t)n * L * a.ldq;
    const T* __restrict__ K = reinterpret_cast<const T*>(a.k) + (size_t)n * L * a.ldk;
    const T* __restrict__ V = reinterpret_cast<const T*>(a.v) + (size_t)n * L * a.ldv;
    T* __restrict__ O = reinterpret_cast<T*>(a.out) + (size_t)n * L * a.ldo;

    // ---- this lane's query row as the B operand of S^T = K Q^T (K order = the lane's 16-byte pieces)
    const int qrow = blockIdx.x * BQ + wave * 32 + l31;
    const int qld = qrow < L ? qrow : L - 1;
    uint4 qf[BF ? 4 : 8];
#pragma unroll
    for (int c = 0; c < (BF ? 4 : 8); ++c)
        qf[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(Q + (size_t)qld * a.ldq) + (2 * c + hi) * 16);

    // ---- staging registers of the next K / V tile
    uint4 rk[NK], rv[NV];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int p = tid + i * NTHR_A, row = p / KP, pc = p % KP;
            uint4 t = make_uint4(0, 0, 0, 0);
            if (key0 + row < L) t = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(K + (size_t)(key0 + row) * a.ldk) + pc * 16);
            rk[i] = t;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int p = tid + i * NTHR_A, row = p / VP, pc = p % VP;
            uint4 t = make_uint4(0, 0, 0, 0);
            if (key0 + row < L) t = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(V + (size_t)(key0 + row) * a.ldv) + pc * 16);
            rv[i] = t;
        }
    };
    auto lstore = [&](int s) {
        unsigned char* kb = smem + s * STAGE;
        unsigned char* vb = kb + KBYTES;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int p = tid + i * NTHR_A, row = p / KP, pc = p % KP;
            *reinterpret_cast<uint4*>(kb + row * A::KROW + ((pc ^ A::kswz(row)) << 4)) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int p = tid + i * NTHR_A, row = p / VP, pc = p % VP;
            *reinterpret_cast<uint4*>(vb + v_piece_off<T>(row, pc)) = rv[i];
        }
    };

    f32x16_t acc[DV / 32];
#pragma unroll
    for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const int ntiles = (L + KVB - 1) / KVB;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) gload((t + 1) * KVB);
        const unsigned char* kb = smem + (t & 1) * STAGE;
        const unsigned char* vb = kb + KBYTES;

        // ---- S^T tile(s): keys x this wave's 32 queries
        f32x16_t s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int key = kt * 32 + l31;
            const unsigned char* krow = kb + key * A::KROW;
            if constexpr (BF) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint4 ka = *reinterpret_cast<const uint4*>(krow + (((2 * ks + hi) ^ A::kswz(key)) << 4));
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka), __builtin_bit_cast(bf16x8_t, qf[ks]), s[kt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint4 ka = *reinterpret_cast<const uint4*>(krow + (((2 * c + hi) ^ A::kswz(key)) << 4));
                    const uint32_t kw[4] = {ka.x, ka.y, ka.z, ka.w}, qw[4] = {qf[c].x, qf[c].y, qf[c].z, qf[c].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kw[e]), __uint_as_float(qw[e]), s[kt], 0, 0, 0);
                }
            }
        }
        if (t * KVB + KVB > L) {                                       // ragged last tile: keys past L do not exist
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KVB + kt * 32 + 8 * (r >> 2) + (r & 3) + 4 * hi >= L) s[kt][r] = NEG_BIG;
        }

        // ---- online softmax over this lane's keys (+ the other half-wave's through one exchange)
        float mloc = s[0][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kt][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        if (__any(m_new > m_run)) {                                    // wave-uniform: rare after the first tiles
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            m_run = m_new;
        }
        // (s - m) first: exact near the maximum, where the probabilities are not negligible (the reference subtracts the row
        // maximum before exp as well: torch.softmax); v_exp_f32 directly -- arguments are <= 0, a flushed denormal is a zero weight
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f((s[kt][r] - m_run) * LOG2E); s[kt][r] = p; l_run += p; }

        // ---- O^T += V^T P^T
        if constexpr (BF) {
            const int i16 = lane & 15, g2 = lane >> 4;
            typedef __attribute__((address_space(3))) s16x4_t* lp_t;
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // B operand: this lane's 8 keys of the 16-key step {16h + 4hi + 0..3, 16h + 8 + 4hi + 0..3} = accumulators 8h .. 8h+7.
                    // The probabilities go into the second product as TWO bf16 terms, p = hi + lo (lo = bf16(p - hi): 16 mantissa bits
                    // together): a single bf16 P (2^-9 per weight, the usual flash-attention choice) made the frozen net's output measurably
                    // noisier than the reference-shaped softmax in fp32 it replaces (tests/test_inpaint.py bf16 bound); the second MFMA per
                    // step costs ~0.1 ms per 8-image pass.
                    uint4 pbw, plw;
                    {
                        uint32_t hw[4], lw[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a0 = s[kt][8 * h + 2 * j], a1 = s[kt][8 * h + 2 * j + 1];
                            hw[j] = f2bf2(a0, a1);
                            lw[j] = f2bf2(a0 - __uint_as_float(hw[j] << 16), a1 - __uint_as_float(hw[j] & 0xffff0000u));
                        }
                        pbw = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                        plw = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                    }
                    const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pbw), pl = __builtin_bit_cast(bf16x8_t, plw);
                    const int row0 = kt * 32 + 16 * h + 4 * hi + (i16 >> 2);          // a1 reads row0 + 8 (same swizzle: (row0 + 8) & 7 == row0 & 7)
                    const int colb = (g2 & 1) * 16 + (i16 & 3) * 4;                    // channel inside a 32-channel block
                    const unsigned char* vrow = vb + row0 * A::VROW + colb * 2;
#pragma unroll
                    for (int dt = 0; dt < DV / 32; ++dt) {
                        const unsigned char* pa = vrow + ((dt ^ (row0 & 7)) << 6);
                        const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pa));
                        const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(pa + 8 * A::VROW));
                        const s16x8_t av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pb, acc[dt], 0, 0, 0);
                        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pl, acc[dt], 0, 0, 0);
                    }
                }
        } else {
            // two-level sums, like the fp32 convolutions (mg_conv_common.h mma_f32_chunk): the 32 keys of this tile are summed from zero in a
            // temporary and added to the running output once -- chaining all L = 4096 keys into the accumulator is one sequential fp32 sum
            // per output.  Two channel blocks in flight keep dependent issues two MFMAs apart.
#pragma unroll
            for (int dt0 = 0; dt0 < DV / 32; dt0 += 2) {
                f32x16_t tsum[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 8 * (r >> 2) + (r & 3) + 4 * hi;
                    const unsigned char* vrow = vb + key * A::VROW + l31 * 4;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x16_t c;
                        if (r == 0) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) c[e] = 0.f;
                        } else {
                            c = tsum[q];
                        }
                        tsum[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(*reinterpret_cast<const float*>(vrow + (dt0 + q) * 128), s[0][r], c, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc[dt0 + q] += tsum[q];
                    asm volatile("" : "+v"(acc[dt0 + q]));           // keep the add here (see mma_f32_chunk)
                }
            }
        }

        if (more) lstore((t + 1) & 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds, per 32-channel block, 4 quads of consecutive channels of its query row
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < L) {
        T* orow = O + (size_t)qrow * a.ldo;
#pragma unroll
        for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[dt][rq * 4 + j] * inv;
                ET<T>::store4(orow + dt * 32 + rq * 8 + hi * 4, o);
            }
    }
}

template <typename T>
int launch_attention(const AttnArgs& a, hipStream_t st)
{
    using A = AT<T>;
    constexpr int LDS = 2 * (A::KVB * A::KROW + A::KVB * A::VROW);
    auto kern = self_attention_kernel<T>;
    mg_raise_lds_cap(reinterpret_cast<const void*>(kern), LDS);
    const dim3 grid((a.L + BQ - 1) / BQ, a.N);
    hipLaunchKernelGGL(kern, grid, dim3(NTHR_A), LDS, st, a);
    return 0;
}

}  // namespace

extern "C" int mg_self_attention(const void* q, const void* k, const void* v, void* out, int32_t dtype, int32_t N, int32_t L,
                                 int32_t d_qk, int32_t d_v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, void* stream)
{
    MG_CHECK_ARG(dtype == MG_F32 || dtype == MG_BF16, "mg_self_attention: bad dtype %d", dtype);
    MG_CHECK_ARG(q && k && v && out, "mg_self_attention: null pointer");
    MG_CHECK_ARG(N > 0 && N <= 65535 && L > 0, "mg_self_attention: bad geometry N=%d L=%d", N, L);
    MG_CHECK_ARG(d_qk == DQK && d_v == DV, "mg_self_attention: built for d_qk = %d, d_v = %d (generator.py:467-470 with dim 256), got %d / %d", DQK, DV, d_qk, d_v);
    const int vec = dtype == MG_BF16 ? 8 : 4;                           // 16-byte row pieces
    MG_CHECK_ARG(ldq >= d_qk && ldk >= d_qk && ldv >= d_v && ldo >= d_v && ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && ldo % 4 == 0,
                 "mg_self_attention: row pitches must cover the rows and keep them 16-byte aligned");
    const size_t es = dtype == MG_BF16 ? 2 : 4;
    MG_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)out % (4 * es) == 0, "mg_self_attention: misaligned pointer");
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.N = N; a.L = L; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MG_BF16) launch_attention<uint16_t>(a, st); else launch_attention<float>(a, st);
    MG_CHECK_LAUNCH("mg_self_attention");
    return MG_OK;
}
