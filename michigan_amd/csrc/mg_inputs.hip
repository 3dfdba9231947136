// mg_inputs.hip -- the input pipeline on the device: everything between the decoded
// u8 maps of a sample and the float tensors Pix2PixModel hands to the networks, which the reference does with
// PIL / numpy / cv2 on loader worker processes (data/pix2pix_dataset.py:66-194, data/base_dataset.py:335-396,
// models/pix2pix_model.py:209-254).  All of it is HBM-bound byte / index work: one thread per pixel, coalesced
// along x, no MFMA, no reshaping into GEMMs.  Integer / byte outputs are bit-exact restatements of the reference
// arithmetic (the float conversions use the same IEEE single operations torchvision's ToTensor / Normalize perform);
// the multi-octave noise follows cv2.resize(INTER_LINEAR) on float64 fields.
#include "mg_common.h"
#include <math.h>

namespace {

constexpr int NTHR = 256;
static inline int ew_grid(int64_t n) { int64_t b = (n + NTHR - 1) / NTHR; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---- crop + flip + ToTensor (+ Normalize / "* 255") ------------------------------------------------------------
// dst[n][c][y][x] = conv(src[n][ytab[y0 + y]][xtab[x0 + (flip ? W-1-x : x)]][c]) (* mul[n][0][y][x])
//   mode 0: (v/255 - 0.5)/0.5   (ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)), base_dataset.py:449-454)
//   mode 1: (v/255)*255, then == 255 -> unknown  (transform_label(.) * 255.0, pix2pix_dataset.py:72-73,117,146-147)
//   mode 2: v/255               (ToTensor only: orient_rgb, pix2pix_dataset.py:127,130)
template <int MODE>
__global__ void crop_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ crop,
                               const int32_t* __restrict__ ytab, const int32_t* __restrict__ xtab, const float* __restrict__ mul,
                               int N, int Hs, int Ws, int C, int H, int W, float unknown, int has_unknown)
{
    const int64_t n = (int64_t)N * H * W;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % W); int64_t p = i / W;
        const int y = (int)(p % H); const int b = (int)(p / H);
        const int x0 = crop[b * 3 + 0], y0 = crop[b * 3 + 1], flip = crop[b * 3 + 2];
        int ly = y0 + y, lx = x0 + (flip ? W - 1 - x : x);
        if (ytab) ly = ytab[ly];
        if (xtab) lx = xtab[lx];
        const uint8_t* s = src + (((size_t)b * Hs + ly) * Ws + lx) * C;
        const float m = mul ? mul[i] : 1.f;
        for (int c = 0; c < C; ++c) {
            float v = __fdiv_rn((float)s[c], 255.f);
            if (MODE == 0) v = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
            if (MODE == 1) { v = __fmul_rn(v, 255.f); if (has_unknown && v == 255.f) v = unknown; }
            if (mul) v = __fmul_rn(v, m);
            dst[(((size_t)b * C + c) * H + y) * W + x] = v;
        }
    }
}

// ---- one-hot label maps (pix2pix_model.py:231-246: FloatTensor(bs, nc, h, w).zero_().scatter_(1, label.long(), 1.0)) ----
__global__ void onehot_kernel(const float* __restrict__ label, float* __restrict__ out, int N, int64_t HW, int nc)
{
    const int64_t n = (int64_t)N * HW;
    GRID_STRIDE(i, n) {
        const int64_t b = i / HW, p = i - b * HW;
        const int64_t k = (int64_t)label[i];                 // .long(): truncation toward zero
        for (int c = 0; c < nc; ++c) out[(b * nc + c) * HW + p] = (c == k) ? 1.f : 0.f;
    }
}

// ---- trans_orient_to_rgb (base_dataset.py:363-385): u8 orientation (0..255 = 0..pi) -> RGB-coded u8 image ----
// rgb = ((cos 2t + 1)/2, (sin 2t + 1)/2, 0.5) * label * 255 -> np.uint8 (truncation).  The 256 possible float64
// colours come from a host-built table (mg_orient_rgb_table: libm double, as numpy computes them); the products are
// IEEE double on the device, so the truncation boundaries are the reference's.
__global__ void orient_rgb_kernel(const uint8_t* __restrict__ orient, const uint8_t* __restrict__ label,
                                  const double* __restrict__ table, uint8_t* __restrict__ out, int64_t npix)
{
    GRID_STRIDE(i, npix) {
        const int o = orient[i];
        const double l = (double)label[i];
        for (int c = 0; c < 3; ++c) {
            const double v = __dmul_rn(__dmul_rn(table[o * 3 + c], l), 255.0);
            out[i * 3 + c] = (uint8_t)(long long)v;          // np.uint8(float64): truncate, wrap modulo 256
        }
    }
}

// ---- generate_hole (base_dataset.py:335-361) -------------------------------------------------------------------
// One workgroup per sample.  coord = row-major list of the non-zero pixels of orient_mask, nums = its length,
// rr = int(int(th * nums) / pi), centre = coord[center_idx], hole = orient_mask * [(h-ch)^2 + (w-cw)^2 < rr]
// + (mask - orient_mask) evaluated as numpy does: the u8 subtraction wraps, the sum is float64, np.uint8 truncates
// and wraps.  The two random draws of the reference (random.uniform(0.5, 1.2), random.randint(0, nums-1)) arrive
// as th[n] and u[n] in [0,1): center_idx = min(int(u * nums), nums - 1) -- same distribution, the count never
// leaves the device.  nums == 0: the hole is orient_mask itself (all zeros).
#ifndef MG_HOLE_THREADS
#define MG_HOLE_THREADS 1024          // (the host-emulation build of tests/hostemu runs one thread per workgroup)
#endif
constexpr int HOLE_THREADS = MG_HOLE_THREADS;
__global__ __launch_bounds__(HOLE_THREADS)
void hole_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ omask, const double* __restrict__ th,
                 const double* __restrict__ u, uint8_t* __restrict__ hole, int32_t* __restrict__ info, int H, int W)
{
    __shared__ int s_cnt[HOLE_THREADS];
    __shared__ int s_center[3];                               // pixel index of the centre, rr, nums
    const int b = blockIdx.x, t = threadIdx.x;
    const int npix = H * W;
    const uint8_t* om = omask + (size_t)b * npix;
    const uint8_t* mk = mask + (size_t)b * npix;
    const int per = (npix + HOLE_THREADS - 1) / HOLE_THREADS;          // contiguous chunk per thread: keeps row-major order
    const int lo = t * per, hi = min(lo + per, npix);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += om[i] != 0;
    s_cnt[t] = cnt;
    __syncthreads();
    // inclusive scan (Hillis-Steele; 1024 values, done once per sample)
    for (int off = 1; off < HOLE_THREADS; off <<= 1) {
        const int v = (t >= off) ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    const int nums = s_cnt[HOLE_THREADS - 1];
    if (nums == 0) {
        for (int i = t; i < npix; i += HOLE_THREADS) hole[(size_t)b * npix + i] = om[i];
        if (t == 0 && info) { info[b * 4 + 0] = 0; info[b * 4 + 1] = -1; info[b * 4 + 2] = -1; info[b * 4 + 3] = 0; }
        return;
    }
    long long cidx = (long long)__dmul_rn(u[b], (double)nums);
    if (cidx > nums - 1) cidx = nums - 1;
    if (cidx < 0) cidx = 0;
    const int before = s_cnt[t] - cnt;                        // non-zeros in earlier chunks
    if (cnt > 0 && cidx >= before && cidx < before + cnt) {
        int k = (int)cidx - before;
        for (int i = lo; i < hi; ++i)
            if (om[i] != 0 && k-- == 0) { s_center[0] = i; break; }
        const long long crop_nums = (long long)__dmul_rn(th[b], (double)nums);        // int(th * nums)
        s_center[1] = (int)(long long)__ddiv_rn((double)crop_nums, 3.141592653589793);  // int(crop_nums / math.pi)
        s_center[2] = nums;
    }
    __syncthreads();
    const int ch = s_center[0] / W, cw = s_center[0] % W, rr = s_center[1];
    if (t == 0 && info) { info[b * 4 + 0] = nums; info[b * 4 + 1] = ch; info[b * 4 + 2] = cw; info[b * 4 + 3] = rr; }
    for (int i = t; i < npix; i += HOLE_THREADS) {
        const int h = i / W, w = i - h * W;
        const int inside = ((h - ch) * (h - ch) + (w - cw) * (w - cw)) < rr;
        const int wrapped = (uint8_t)(mk[i] - om[i]);          // numpy u8 - u8 wraps
        hole[(size_t)b * npix + i] = (uint8_t)(om[i] * inside + wrapped);   // float64 sum -> np.uint8 wraps the same way
    }
}

// ---- generate_noise (base_dataset.py:387-396): sum over octaves of cv2.resize(field_o, INTER_LINEAR) / n_octaves ----
// fields: float64, per sample the octaves back to back, octave o = [S >> o][S >> o][3] (HWC, as np.random.normal
// draws them).  cv2's linear resize of a 64-bit source: coordinates and the two weights in float, horizontal pass
// then vertical pass in double, source columns clamped with the weight reset at the border, source rows clamped.
// The running sum is float32 (noise is a float32 array that numpy accumulates float64 terms into).
struct LinTap { int i0, i1; double w0, w1; };
__device__ __forceinline__ LinTap cv_linear_x(int d, int ssize, double scale)
{
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);        // no fused multiply-add: cv2 rounds the product
    int s = (int)floorf(f);
    f -= (float)s;
    LinTap t;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; t.i0 = s; t.i1 = s; t.w0 = 1.0; t.w1 = 0.0; return t; }   // dx >= xmax: S[sx] * 1
    t.i0 = s; t.i1 = s + 1; t.w0 = (double)(1.f - f); t.w1 = (double)f;
    return t;
}
__device__ __forceinline__ LinTap cv_linear_y(int d, int ssize, double scale)
{
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);        // no fused multiply-add: cv2 rounds the product
    int s = (int)floorf(f);
    f -= (float)s;
    LinTap t;
    t.i0 = min(max(s, 0), ssize - 1); t.i1 = min(max(s + 1, 0), ssize - 1);
    t.w0 = (double)(1.f - f); t.w1 = (double)f;
    return t;
}
__global__ void noise_kernel(const double* __restrict__ fields, float* __restrict__ out, int N, int S, int noct, int64_t per_sample)
{
    const int64_t n = (int64_t)N * S * S;
    const float wsum = (float)noct;                           // weight stays 1.0 in the reference loop
    GRID_STRIDE(i, n) {
        const int x = (int)(i % S); int64_t p = i / S;
        const int y = (int)(p % S); const int b = (int)(p / S);
        const double* f = fields + (size_t)b * per_sample;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int o = 0; o < noct; ++o) {
            const int s = S >> o;
            if (s == S) {                                      // same size: cv2.resize copies
                for (int c = 0; c < 3; ++c) acc[c] = (float)__dadd_rn((double)acc[c], f[((size_t)y * s + x) * 3 + c]);
            } else {
                const double scale = 1.0 / ((double)S / (double)s);
                const LinTap tx = cv_linear_x(x, s, scale), ty = cv_linear_y(y, s, scale);
                const double* r0 = f + (size_t)ty.i0 * s * 3;
                const double* r1 = f + (size_t)ty.i1 * s * 3;
                for (int c = 0; c < 3; ++c) {
                    double h0, h1;
                    if (tx.i0 == tx.i1) { h0 = r0[tx.i0 * 3 + c]; h1 = r1[tx.i0 * 3 + c]; }
                    else {
                        h0 = __dadd_rn(__dmul_rn(r0[tx.i0 * 3 + c], tx.w0), __dmul_rn(r0[tx.i1 * 3 + c], tx.w1));
                        h1 = __dadd_rn(__dmul_rn(r1[tx.i0 * 3 + c], tx.w0), __dmul_rn(r1[tx.i1 * 3 + c], tx.w1));
                    }
                    const double v = __dadd_rn(__dmul_rn(h0, ty.w0), __dmul_rn(h1, ty.w1));
                    acc[c] = (float)__dadd_rn((double)acc[c], v);
                }
            }
            f += (size_t)s * s * 3;
        }
        for (int c = 0; c < 3; ++c) out[(((size_t)b * 3 + c) * S + y) * S + x] = __fdiv_rn(acc[c], wsum);
    }
}

// Tiled form of the same arithmetic (A/B variant, mg_inputs_set_option(0, 1); NOT the default: it measured 12 % slower than
// the per-pixel kernel -- the gathered taps hit in L1/L2 and were not the limit): a workgroup owns a 32 x 8 pixel tile of one sample (128-byte output rows per
// channel plane) and first stages, for every octave below full size, the few source texels its pixels interpolate
// between ((32 >> o) + 3) x ((8 >> o) + 3) at most) into LDS -- the per-pixel kernel above issues 72 gathered 8-byte
// loads per pixel through the vector L1; here each field value is fetched once per tile and the taps are LDS reads.
// Octaves whose patch does not fit the LDS budget (only possible for non power-of-two sizes) read global memory.
constexpr int NT_W = 32, NT_H = 8, NT_MAXTEX = 448, NT_MAXOCT = 16;
__global__ __launch_bounds__(NT_W * NT_H)
void noise_tiled_kernel(const double* __restrict__ fields, float* __restrict__ out, int N, int S, int noct, int64_t per_sample,
                        int tiles_x, int tiles_y)
{
    __shared__ double s_tex[NT_MAXTEX * 3];
    __shared__ int s_meta[NT_MAXOCT][5];                      // LDS offset (texels, -1 = not staged), x0, y0, patch width, patch height
    const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
    const int X0 = (tile % tiles_x) * NT_W, Y0 = (tile / tiles_x) * NT_H;
    const int X1 = min(X0 + NT_W, S) - 1, Y1 = min(Y0 + NT_H, S) - 1;
    const double* fb = fields + (size_t)b * per_sample;
    const float wsum = (float)noct;
    for (int t = threadIdx.x; t < 1; t += blockDim.x) {       // one thread lays out the patches (a dozen integers)
        int used = 0;
        for (int o = 1; o < noct && o < NT_MAXOCT; ++o) {
            const int s = S >> o;
            const double scale = 1.0 / ((double)S / (double)s);
            const int x0 = cv_linear_x(X0, s, scale).i0, x1 = cv_linear_x(X1, s, scale).i1;
            const int y0 = cv_linear_y(Y0, s, scale).i0, y1 = cv_linear_y(Y1, s, scale).i1;
            const int pw = x1 - x0 + 1, ph = y1 - y0 + 1;
            const bool fits = used + pw * ph <= NT_MAXTEX;
            s_meta[o][0] = fits ? used : -1; s_meta[o][1] = x0; s_meta[o][2] = y0; s_meta[o][3] = pw; s_meta[o][4] = ph;
            if (fits) used += pw * ph;
        }
    }
    __syncthreads();
    {
        const double* f = fb + (size_t)S * S * 3;
        for (int o = 1; o < noct && o < NT_MAXOCT; ++o) {
            const int s = S >> o, off = s_meta[o][0], x0 = s_meta[o][1], y0 = s_meta[o][2], pw = s_meta[o][3], ph = s_meta[o][4];
            if (off >= 0)
                for (int e = threadIdx.x; e < pw * ph * 3; e += blockDim.x) {
                    const int c = e % 3, px = (e / 3) % pw, py = e / (3 * pw);
                    s_tex[off * 3 + e] = f[((size_t)(y0 + py) * s + (x0 + px)) * 3 + c];
                }
            f += (size_t)s * s * 3;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NT_W * NT_H; t += blockDim.x) {
        const int x = X0 + t % NT_W, y = Y0 + t / NT_W;
        if (x >= S || y >= S) continue;
        float acc[3];
        for (int c = 0; c < 3; ++c) acc[c] = (float)__dadd_rn(0.0, fb[((size_t)y * S + x) * 3 + c]);    // octave 0: same size, a copy
        const double* f = fb + (size_t)S * S * 3;
        for (int o = 1; o < noct; ++o) {
            const int s = S >> o;
            const double scale = 1.0 / ((double)S / (double)s);
            const LinTap tx = cv_linear_x(x, s, scale), ty = cv_linear_y(y, s, scale);
            const bool staged = o < NT_MAXOCT && s_meta[o][0] >= 0;
            const double* r0; const double* r1; int i0 = tx.i0, i1 = tx.i1;
            if (staged) {
                const int off = s_meta[o][0], x0 = s_meta[o][1], y0 = s_meta[o][2], pw = s_meta[o][3];
                r0 = s_tex + (size_t)(off + (ty.i0 - y0) * pw) * 3; r1 = s_tex + (size_t)(off + (ty.i1 - y0) * pw) * 3;
                i0 -= x0; i1 -= x0;
            } else { r0 = f + (size_t)ty.i0 * s * 3; r1 = f + (size_t)ty.i1 * s * 3; }
            for (int c = 0; c < 3; ++c) {
                double h0, h1;
                if (i0 == i1) { h0 = r0[i0 * 3 + c]; h1 = r1[i0 * 3 + c]; }
                else {
                    h0 = __dadd_rn(__dmul_rn(r0[i0 * 3 + c], tx.w0), __dmul_rn(r0[i1 * 3 + c], tx.w1));
                    h1 = __dadd_rn(__dmul_rn(r1[i0 * 3 + c], tx.w0), __dmul_rn(r1[i1 * 3 + c], tx.w1));
                }
                const double v = __dadd_rn(__dmul_rn(h0, ty.w0), __dmul_rn(h1, ty.w1));
                acc[c] = (float)__dadd_rn((double)acc[c], v);
            }
            f += (size_t)s * s * 3;
        }
        for (int c = 0; c < 3; ++c) out[(((size_t)b * 3 + c) * S + y) * S + x] = __fdiv_rn(acc[c], wsum);
    }
}

// ---- transforms.Resize(osize, Image.BICUBIC) on u8 images (base_dataset.py:421-424 -> Pillow Resample.c, 8 bits per channel):
// two separable passes, horizontal first, each output = clip8((2^21 + sum_k src[first + k] * coef[k]) >> 22) with
// the 22-bit fixed-point coefficients and [first, count) windows of mg_bicubic_table; the intermediate image is u8
// like Pillow's.  Integer multiply-adds only: bit-exact by construction once the tables agree.
template <bool HORIZ>
__global__ void bicubic_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int32_t* __restrict__ bounds,
                                    const int32_t* __restrict__ coef, int ksize, int N, int Hin, int Win, int Hout, int Wout, int C)
{
    const int64_t n = (int64_t)N * Hout * Wout;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % Wout); int64_t p = i / Wout;
        const int y = (int)(p % Hout); const int b = (int)(p / Hout);
        const int o = HORIZ ? x : y;
        const int first = bounds[o * 2], cnt = bounds[o * 2 + 1];
        const int32_t* k = coef + (size_t)o * ksize;
        const uint8_t* s = HORIZ ? src + (((size_t)b * Hin + y) * Win + first) * C : src + (((size_t)b * Hin + first) * Win + x) * C;
        const size_t step = HORIZ ? (size_t)C : (size_t)Win * C;
        int acc[4] = {1 << 21, 1 << 21, 1 << 21, 1 << 21};
        for (int j = 0; j < cnt; ++j) {
            const int w = k[j];
            for (int c = 0; c < C; ++c) acc[c] += (int)s[j * step + c] * w;
        }
        for (int c = 0; c < C; ++c) {
            int v = acc[c] >> 22;                              // arithmetic shift, then Pillow's clip8 table
            dst[i * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

}  // namespace

// ---- host helpers (no GPU involved) --------------------------------------------------------------------------------
extern "C" int mg_nearest_table(int32_t src, int32_t dst, int32_t* table)
{
    MG_CHECK_ARG(src > 0 && dst > 0 && table, "mg_nearest_table: bad arguments");
    // Pillow's nearest-neighbour resize walks the source coordinate by repeated addition in double
    // (transforms.Resize(osize, Image.NEAREST), base_dataset.py:421-424)
    const double sc = (double)src / (double)dst;
    double xo = sc * 0.5;
    for (int i = 0; i < dst; ++i) { int v = (int)xo; table[i] = v < src ? v : src - 1; xo += sc; }
    return MG_OK;
}
extern "C" int mg_orient_rgb_table(double* table)
{
    MG_CHECK_ARG(table, "mg_orient_rgb_table: null pointer");
    for (int o = 0; o < 256; ++o) {
        const double t = (double)o / 255.0 * 3.141592653589793;
        table[o * 3 + 0] = (cos(2 * t) + 1) / 2;
        table[o * 3 + 1] = (sin(2 * t) + 1) / 2;
        table[o * 3 + 2] = 0.5;
    }
    return MG_OK;
}
// Pillow Resample.c: bicubic_filter (a = -0.5), precompute_coeffs, normalize_coeffs_8bpc
static inline double mg_bicubic_filter(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
extern "C" int mg_bicubic_ksize(int32_t in_size, int32_t out_size)
{
    if (in_size <= 0 || out_size <= 0) return -1;
    double filterscale = (double)((float)in_size - 0.f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}
extern "C" int mg_bicubic_table(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coef)
{
    MG_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && coef, "mg_bicubic_table: bad arguments");
    const int ksize = mg_bicubic_ksize(in_size, out_size);
    MG_CHECK_ARG(ksize <= 256, "mg_bicubic_table: scale factor too large");
    double scale = (double)((float)in_size - 0.f) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    double k[256];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.f + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) { const double w = mg_bicubic_filter((x + xmin - center + 0.5) * ss); k[x] = w; ww += w; }
        for (int x = 0; x < xmax; ++x) if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < ksize; ++x) {
            const double v = x < xmax ? k[x] : 0.0;
            coef[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
        }
        bounds[xx * 2] = xmin; bounds[xx * 2 + 1] = xmax;
    }
    return MG_OK;
}
extern "C" int64_t mg_noise_field_len(int32_t S)
{
    int64_t n = 0;
    for (int s = S; s >= 8; s /= 2) n += (int64_t)s * s * 3;
    return n;
}

static int g_noise_tiled = 0;                                  // A/B switch: measured on MI355X at 8 x 512^2 the LDS-tiled form takes 86 us, the per-pixel form 77 us
extern "C" int mg_inputs_set_option(int32_t key, int32_t value)
{
    MG_CHECK_ARG(key == 0, "mg_inputs_set_option: unknown key");
    g_noise_tiled = value != 0;
    return MG_OK;
}

// ---- launches ---------------------------------------------------------------------------------------------------------
extern "C" int mg_input_crop_u8(const uint8_t* src, float* dst, const int32_t* crop, const int32_t* ytab, const int32_t* xtab,
                                const float* mul, int32_t N, int32_t Hs, int32_t Ws, int32_t C, int32_t H, int32_t W,
                                int32_t mode, int32_t unknown_label, void* stream)
{
    MG_CHECK_ARG(src && dst && crop, "mg_input_crop_u8: null pointer");
    MG_CHECK_ARG(N > 0 && Hs > 0 && Ws > 0 && C > 0 && C <= 4 && H > 0 && W > 0, "mg_input_crop_u8: bad geometry");
    MG_CHECK_ARG(mode >= 0 && mode <= 2, "mg_input_crop_u8: mode must be 0 (image), 1 (label map) or 2 (ToTensor only)");
    MG_CHECK_ARG((ytab && xtab) || (H <= Hs && W <= Ws), "mg_input_crop_u8: crop larger than the source");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = ew_grid((int64_t)N * H * W);
    const float unk = (float)unknown_label; const int has = unknown_label >= 0;
    if (mode == 0) hipLaunchKernelGGL(crop_u8_kernel<0>, dim3(grid), dim3(NTHR), 0, st, src, dst, crop, ytab, xtab, mul, N, Hs, Ws, C, H, W, unk, has);
    else if (mode == 1) hipLaunchKernelGGL(crop_u8_kernel<1>, dim3(grid), dim3(NTHR), 0, st, src, dst, crop, ytab, xtab, mul, N, Hs, Ws, C, H, W, unk, has);
    else hipLaunchKernelGGL(crop_u8_kernel<2>, dim3(grid), dim3(NTHR), 0, st, src, dst, crop, ytab, xtab, mul, N, Hs, Ws, C, H, W, unk, has);
    MG_CHECK_LAUNCH("mg_input_crop_u8");
    return MG_OK;
}

extern "C" int mg_onehot_labels(const float* label, float* out, int32_t N, int64_t HW, int32_t nc, void* stream)
{
    MG_CHECK_ARG(label && out && N > 0 && HW > 0 && nc > 0, "mg_onehot_labels: bad arguments");
    hipLaunchKernelGGL(onehot_kernel, dim3(ew_grid((int64_t)N * HW)), dim3(NTHR), 0, reinterpret_cast<hipStream_t>(stream), label, out, N, HW, nc);
    MG_CHECK_LAUNCH("mg_onehot_labels");
    return MG_OK;
}

extern "C" int mg_orient_to_rgb_u8(const uint8_t* orient, const uint8_t* label, const double* table, uint8_t* out, int64_t npix, void* stream)
{
    MG_CHECK_ARG(orient && label && table && out && npix > 0, "mg_orient_to_rgb_u8: bad arguments");
    hipLaunchKernelGGL(orient_rgb_kernel, dim3(ew_grid(npix)), dim3(NTHR), 0, reinterpret_cast<hipStream_t>(stream), orient, label, table, out, npix);
    MG_CHECK_LAUNCH("mg_orient_to_rgb_u8");
    return MG_OK;
}

extern "C" int mg_generate_hole_u8(const uint8_t* mask, const uint8_t* orient_mask, const double* th, const double* u,
                                   uint8_t* hole, int32_t* info, int32_t N, int32_t H, int32_t W, void* stream)
{
    MG_CHECK_ARG(mask && orient_mask && th && u && hole, "mg_generate_hole_u8: null pointer");
    MG_CHECK_ARG(N > 0 && H > 0 && W > 0 && (int64_t)H * W < (1 << 30) && H < 32768 && W < 32768, "mg_generate_hole_u8: bad geometry");
    hipLaunchKernelGGL(hole_kernel, dim3(N), dim3(HOLE_THREADS), 0, reinterpret_cast<hipStream_t>(stream), mask, orient_mask, th, u, hole, info, H, W);
    MG_CHECK_LAUNCH("mg_generate_hole_u8");
    return MG_OK;
}

extern "C" int mg_noise_octaves(const double* fields, float* out, int32_t N, int32_t S, void* stream)
{
    MG_CHECK_ARG(fields && out && N > 0 && S >= 8, "mg_noise_octaves: bad arguments");
    int noct = 0;
    for (int s = S; s >= 8; s /= 2) ++noct;                    // width //= 2 while >= 8: octave o has side S >> o
    const int tx = (S + NT_W - 1) / NT_W, ty = (S + NT_H - 1) / NT_H;
    if (g_noise_tiled && (int64_t)N * tx * ty < (1ll << 31))
        hipLaunchKernelGGL(noise_tiled_kernel, dim3((unsigned)(N * tx * ty)), dim3(NT_W * NT_H), 0, reinterpret_cast<hipStream_t>(stream),
                           fields, out, N, S, noct, mg_noise_field_len(S), tx, ty);
    else
        hipLaunchKernelGGL(noise_kernel, dim3(ew_grid((int64_t)N * S * S)), dim3(NTHR), 0, reinterpret_cast<hipStream_t>(stream),
                           fields, out, N, S, noct, mg_noise_field_len(S));
    MG_CHECK_LAUNCH("mg_noise_octaves");
    return MG_OK;
}

extern "C" int mg_resize_bicubic_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                                    const int32_t* ybounds, const int32_t* ycoef, int32_t ky, int32_t N, int32_t Hs, int32_t Ws,
                                    int32_t Hd, int32_t Wd, int32_t C, void* stream)
{
    MG_CHECK_ARG(src && tmp && dst && xbounds && xcoef && ybounds && ycoef, "mg_resize_bicubic_u8: null pointer");
    MG_CHECK_ARG(N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C <= 4 && kx > 0 && ky > 0, "mg_resize_bicubic_u8: bad geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bicubic_pass_kernel<true>, dim3(ew_grid((int64_t)N * Hs * Wd)), dim3(NTHR), 0, st, src, tmp, xbounds, xcoef, kx, N, Hs, Ws, Hs, Wd, C);
    MG_CHECK_LAUNCH("mg_resize_bicubic_u8(horizontal)");
    hipLaunchKernelGGL(bicubic_pass_kernel<false>, dim3(ew_grid((int64_t)N * Hd * Wd)), dim3(NTHR), 0, st, (const uint8_t*)tmp, dst, ybounds, ycoef, ky, N, Hs, Wd, Hd, Wd, C);
    MG_CHECK_LAUNCH("mg_resize_bicubic_u8(vertical)");
    return MG_OK;
}
