// tools/mfma_ceiling.hip -- what does the matrix pipe of an MI355X deliver when NOTHING but MFMAs runs?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip && /tmp/mfma_ceiling
//
// Every wave keeps 8 independent 32x32 fp32 accumulator tiles (128 registers, the halo conv's wave tile) and issues
// v_mfma_f32_32x32x16_bf16 back to back from operand fragments that live in registers: no LDS, no global memory, no barrier inside the
// timed loop.  Two A fragments x four B fragments per K step and two K steps per iteration, like the conv's tap body, so consecutive MFMAs
// see different operands (data toggling is what the power budget prices: DESIGN.md section 3.1).  Reported for random bf16 operands
// (N(0,1)), half-zero (ReLU-like) and all-zero operands, at one and two workgroups per CU (one / two waves per SIMD).
// The numbers are the ceiling `roofline.frac` of any bf16 MFMA kernel on this part can be read against: a real kernel adds LDS / VMEM /
// VALU power on top and therefore clocks lower still.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void mfma_only(const uint4* __restrict__ frag, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 12 operand fragments per wave (2 K steps x (2 A + 4 B)), 16 bytes per lane each
    bf16x8_t f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        uint4 v = frag[((blockIdx.x * 4 + wave) % 64 * 12 + i) * 64 + lane];
        f[i] = *reinterpret_cast<bf16x8_t*>(&v);
    }
    f32x16_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks * 6 + a], f[ks * 6 + 2 + b], acc[a * 4 + b], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[t][j];
    if (s == 123.456f) out[0] = s;                             // keeps the accumulators alive; never true for these inputs
}

// the same MFMA stream with the conv's operand traffic from LDS: 6 ds_read_b128 per 8 MFMAs (2 A + 4 B fragments per K step), conflict-free
// lane-linear addresses, fresh fragments every K step -- still no global memory, no LDS writes, no barrier in the loop
__global__ __launch_bounds__(256, 2) void mfma_lds(const uint4* __restrict__ frag, float* __restrict__ out, int iters)
{
    __shared__ uint4 lds[48 * 64];                              // 48 KiB of operand fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = frag[((blockIdx.x % 16) * 48 * 64 + i) % (64 * 12 * 64)];
    __syncthreads();
    f32x16_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    int slot = wave * 12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t f[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                uint4 v = lds[((slot + ks * 6 + i) % 48) * 64 + lane];
                f[i] = *reinterpret_cast<bf16x8_t*>(&v);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a], f[2 + b], acc[a * 4 + b], 0, 0, 0);
        }
        slot = (slot + 12) % 48;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[t][j];
    if (s == 123.456f) out[0] = s;
}

// what a one-wave-per-SIMD kernel with a 128 x 128 wave tile (256 accumulator registers) would read: 8 ds_read_b128 (4 A + 4 B) per 16 MFMAs
__global__ __launch_bounds__(256, 1) void mfma_lds_big(const uint4* __restrict__ frag, float* __restrict__ out, int iters)
{
    __shared__ uint4 lds[48 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = frag[((blockIdx.x % 16) * 48 * 64 + i) % (64 * 12 * 64)];
    __syncthreads();
    f32x16_t acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    int slot = wave * 12;
    for (int it = 0; it < iters; ++it) {
        bf16x8_t f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4 v = lds[((slot + i) % 48) * 64 + lane];
            f[i] = *reinterpret_cast<bf16x8_t*>(&v);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a], f[4 + b], acc[a * 4 + b], 0, 0, 0);
        slot = (slot + 8) % 48;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[t][j];
    if (s == 123.456f) out[0] = s;
}

// would lane shifts replace LDS reads?  The three taps of a kernel row read the SAME staged pixels shifted by one: the B fragment of tap
// dx + 1 is the B fragment of tap dx rotated by one lane inside each 16-pixel row (v_mov_b32 row_ror:1, 4 per fragment) + the halo pixel
// for one lane in 16 (an exec-masked ds_read_b128).  Per three K steps: 6 A reads + 4 B reads + 8 masked reads + 32 DPP moves (was 18 reads).
__device__ __forceinline__ bf16x8_t ror1(bf16x8_t v)
{
    typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4_t;
    i32x4_t x = *reinterpret_cast<i32x4_t*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_update_dpp(x[j], x[j], 0x121, 0xf, 0xf, false);     // row_ror:1
    return *reinterpret_cast<bf16x8_t*>(&x);
}

__global__ __launch_bounds__(256, 2) void mfma_dpp(const uint4* __restrict__ frag, float* __restrict__ out, int iters)
{
    __shared__ uint4 lds[48 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) lds[i] = frag[((blockIdx.x % 16) * 48 * 64 + i) % (64 * 12 * 64)];
    __syncthreads();
    f32x16_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    int slot = wave * 12;
    const bool edge = (lane & 15) == 15;
    for (int it = 0; it < iters; ++it) {
        bf16x8_t fb[4];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            bf16x8_t fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint4 v = lds[((slot + ks * 2 + i) % 48) * 64 + lane];
                fa[i] = *reinterpret_cast<bf16x8_t*>(&v);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (ks == 0) {
                    uint4 v = lds[((slot + 6 + b) % 48) * 64 + lane];
                    fb[b] = *reinterpret_cast<bf16x8_t*>(&v);
                } else {
                    fb[b] = ror1(fb[b]);
                    if (edge) {
                        uint4 v = lds[((slot + 6 + b + ks * 4) % 48) * 64 + lane];
                        fb[b] = *reinterpret_cast<bf16x8_t*>(&v);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a * 4 + b], 0, 0, 0);
        }
        slot = (slot + 18) % 48;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[t][j];
    if (s == 123.456f) out[0] = s;
}

static uint16_t bf16_of(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

static float gauss(uint64_t& st)
{
    auto next = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53); };
    double u1 = next() + 1e-12, u2 = next();
    return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

int main()
{
    const size_t nfrag = 64 * 12 * 64;                         // 64 distinct wave operand sets
    std::vector<uint16_t> h(nfrag * 8);
    uint4* d; float* out;
    CHECK(hipMalloc(&d, nfrag * 16)); CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int cus = 0; CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 4000;                                     // x 16 MFMAs: ~1.4 ms per launch at one workgroup per CU
    const char* names[3] = {"random N(0,1)", "half zero (ReLU-like)", "all zero"};
    for (int mode = 0; mode < 3; ++mode) {
        uint64_t st = 12345;
        for (size_t i = 0; i < h.size(); ++i) {
            float g = gauss(st);
            h[i] = mode == 2 ? 0 : bf16_of(mode == 1 && g < 0.f ? 0.f : g);
        }
        CHECK(hipMemcpy(d, h.data(), nfrag * 16, hipMemcpyHostToDevice));
        for (int variant = 0; variant < 4; ++variant)
        for (int wgs = 1; wgs <= (variant == 2 ? 1 : 2); ++wgs) {
            const int grid = cus * wgs;
            auto kern = variant == 3 ? mfma_dpp : variant == 2 ? mfma_lds_big : variant ? mfma_lds : mfma_only;
            const int per_it = variant == 3 ? 24 : 16;             // MFMAs per loop iteration
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, out, iters);     // warm the clocks
            CHECK(hipDeviceSynchronize());
            const int reps = 40;                                // ~60-100 ms of sustained load: the DVFS loop has settled
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, out, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)reps * grid * 4 * iters * per_it * (2.0 * 32 * 32 * 16);
            const double tf = flop / (ms * 1e-3) / 1e12;
            // 32 cycles per MFMA and SIMD at full rate: cycles = waves_per_simd x iters x 16 x 32 per launch -> implied clock
            const double clk = (double)wgs * iters * per_it * 32 * reps / (ms * 1e-3) / 1e9;
            printf("%-22s %-29s %d workgroup(s)/CU (%d wave(s)/SIMD): %7.1f TFLOP/s = %.3f of 2500; clock if the pipe never idles %.2f GHz\n",
                   names[mode], variant == 3 ? "MFMA + 3.3 reads + 11 dpp / 8" : variant == 2 ? "MFMA + 8 ds_read_b128 / 16" : variant ? "MFMA + 6 ds_read_b128 / 8" : "MFMA only (registers)", wgs, wgs, tf, tf / 2500.0, clk);
        }
    }
    return 0;
}
