// Shared by wgrad.hip (exact fp32) and wgrad_bf16.hip (split-bf16): the argument block of the weight-gradient contraction,
// the split-K plan and the launcher of the deterministic unpack kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

namespace {

struct WgradArgs {
    const float* S;      // planar [Sp][Fs][JpS]
    const float* L;      // planar [Lp][Fl][JpL]
    int Sp, Lp, Fs, Fl;
    int J, JpS, JpL;
    int dt0;             // L column = S column + kt + dt0
    float* part;         // [nsplit][TAPS][SpPad][LpPad]
    int SpPad, LpPad;
    int jtiles;          // ceil(J / WG_JT)
    int jt_per_split;
    // three-product (Gauss) form of the complex weight gradient (wgrad.hip): nprod = 3 independent contractions in one launch,
    // blockIdx.z = product * tilesL + L tile; product p reads (Sx[p], Lx[p]) and writes part + p * prod_stride
    // (separate scalar fields, selected with ?: -- an array indexed by the product would send the whole by-value argument
    //  block, and with it every field the main loop reads, through scratch memory: 2.3x slower, measured)
    int nprod, tilesL;
    const float *S1, *S2, *L1, *L2;      // products 1 and 2 (product 0: S, L)
    long long prod_stride;
};

struct Plan { int tilesS, tilesL, nsplit, jtiles, jt_per_split, SpPad, LpPad; };

inline Plan make_plan(int Sp, int Lp, int J, int MS, int ML, int JT) {
    Plan p;
    p.tilesS = (Sp + MS - 1) / MS;
    p.tilesL = (Lp + ML - 1) / ML;
    p.SpPad = p.tilesS * MS;
    p.LpPad = p.tilesL * ML;
    p.jtiles = (J + JT - 1) / JT;
    // workgroups aimed at (two per CU are resident): more, shorter workgroups even out the last round at the price of more
    // partial tiles for the unpack kernel to sum (IDV_WGRAD_WGS overrides, experiments)
    static const int target = [] { const char* e = getenv("IDV_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    int want = (target + p.tilesS * p.tilesL - 1) / (p.tilesS * p.tilesL);
    if (want < 1) want = 1;
    if (want > p.jtiles) want = p.jtiles;
    p.jt_per_split = (p.jtiles + want - 1) / want;
    p.nsplit = (p.jtiles + p.jt_per_split - 1) / p.jt_per_split;
    return p;
}

inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// sums the split-K partial tiles in a fixed order and folds the four real products of a complex pair into (dW_re, dW_im)
void launch_wgrad_unpack_conv(const float* part, int nsplit, int SpPad, int LpPad, int Cout, int Cx, int Cin_total, int ci_off,
                              int transposed, float* dw_re, float* dw_im, hipStream_t st);
// out[rowmap(m)][k] (+)= sum over the split-K partial tiles; rowmap 1: LSTM gate order (colp -> g*H + u)
void launch_wgrad_unpack_plain(const float* part, int nsplit, int SpPad, int LpPad, int M, int K, int ldw, int rowmap, int H,
                               int accumulate, float* dw, hipStream_t st);
