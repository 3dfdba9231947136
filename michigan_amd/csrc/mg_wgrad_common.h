// mg_wgrad_common.h -- kernel-argument block of the 3x3 weight-gradient kernel (mg_wgrad3x3.hip) and
// its launcher, called from the generic entry point in mg_wgrad.hip.
#pragma once
#include "mg_common.h"

struct Wg3K {
    const void* x; const void* dy; float* dw; float* dbias;
    int N, H, W, Cin, Cg;
    int nstg;               // N*H*W / 32 stages of 32 output pixels
    int sps;                // stages per split
    int tiles_m, tiles_n;
    int splitk;             // requested split count (0 = choose)
    long det_stride;        // deterministic mode: floats per split slab (dw / dbias then point INTO the workspace), 0 = fp32 atomics
    int half_cu;            // mg_wgrad_desc.flags bit 1: leave half of every CU to the other stream (one workgroup of this kernel per CU)
    int stripe_w;           // stage order inside an image: column stripes of this many pixels (a multiple of 32), each walked top to bottom; W = plain raster order
};

// One partial value of split `split`: an fp32 atomic into the shared dW, or (deterministic mode) a plain store into the split's own slab.
__device__ __forceinline__ void wg_accum(float* base, long det_stride, int split, size_t idx, float v)
{
    if (det_stride) base[(size_t)split * det_stride + idx] = v;
    else atomicAdd(base + idx, v);
}
int launch_wgrad_det_finish(const float* ws, int nsplit, long stride, float* dw, long ndw, float* dbias, int nbias, hipStream_t st);   // mg_wgrad.hip

int launch_wgrad3x3(Wg3K& k, hipStream_t st, int* nsplit = nullptr, bool dry = false);   // mg_wgrad3x3.hip; dry: only report the split count
bool wgrad_thin_applies(const Wg3K& k);           // mg_conv_thin.hip: 8-channel X, 64 / 128-channel dY
int launch_wgrad_thin(Wg3K& k, hipStream_t st, int* nsplit = nullptr, bool dry = false);

// generic tap-window weight gradient over an 8-channel X (mg_conv_thin.hip)
struct WgT {
    const void* x; const void* dy; float* ws;
    int N, Hin, Win, Hj, Wj, ntaps;
    int HH, HW, dy0, dx0;   // halo rows / pixels per row, first tap of the window
    long slab;              // floats per workgroup slab: ntaps*64*8 (+ 64 bias sums)
    int has_bias;
    int tap[MG_MAX_TAPS];   // pixel offset of each tap inside the halo
};
bool wgrad_thin_taps_applies(const mg_wgrad_desc* d);
int launch_wgrad_thin_taps(const mg_wgrad_desc* d, hipStream_t st, float* dw, float* dbias, long det_stride, int* nsplit, bool dry);
float* mg_stream_scratch(hipStream_t st, size_t bytes);   // mg_conv.hip: grow-only fp32 scratch, one buffer per stream
