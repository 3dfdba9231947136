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
};

int launch_wgrad3x3(Wg3K& k, hipStream_t st);   // mg_wgrad3x3.hip
bool wgrad_thin_applies(const Wg3K& k);           // mg_conv_thin.hip: 8-channel X, 64 / 128-channel dY
int launch_wgrad_thin(Wg3K& k, hipStream_t st);
