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
    // balanced split-K (make_plan_rounds): split i owns the steps [i * steps_total / nsplit_bal, (i + 1) * steps_total / nsplit_bal)
    // of the flattened (column tile, frequency row) sequence; 0: whole column tiles per split (jt_per_split)
    int nsplit_bal;
    long long steps_total;
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

inline int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n <= 0) {
            (void)hipGetLastError();
            return 256;                       // no device visible (sizing queries in the build container): the MI355X figure
        }
        return n;
    }();
    return cus;
}

// Split-K plan aimed at WHOLE rounds of resident workgroups: `tiles` output tiles (S tiles x L tiles x products) x nsplit equal
// splits = just under rounds x (CUs x occupancy) workgroups, every split the same number of (column tile, frequency row)
// steps to within one.  The round count (1 .. WGRAD_MAX_ROUNDS) minimises  rounds x (steps per split + ~16 steps of per-workgroup
// prologue / partial-tile write), a model fitted to tests/tools/wgrad_layers_probe.py at B = 32 (it picks within 0.3 % of
// the best measured round count on every DCCRN-CL layer; the 1024-workgroup target of make_plan left partial last rounds,
// e.g. 5.02, that cost up to 25 % on single layers and 6 % over the step).  Fs = 0: the plan with the most splits (work-space bound).
// IDV_WGRAD_ROUNDS forces the round count (experiments).
constexpr int WGRAD_MAX_ROUNDS = 4;
inline Plan make_plan_rounds(int Sp, int Lp, int J, int MS, int ML, int JT, int nprod, int occ, int Fs) {
    Plan p;
    p.tilesS = (Sp + MS - 1) / MS;
    p.tilesL = (Lp + ML - 1) / ML;
    p.SpPad = p.tilesS * MS;
    p.LpPad = p.tilesL * ML;
    p.jtiles = (J + JT - 1) / JT;
    p.jt_per_split = 0;
    static const int forced = [] { const char* e = getenv("IDV_WGRAD_ROUNDS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const long long slots = (long long)device_cus() * occ, tiles = (long long)p.tilesS * p.tilesL * nprod;
    auto splits = [&](int r) {
        long long ns = slots * r / tiles;
        if (ns < 1) ns = 1;
        if (ns > p.jtiles) ns = p.jtiles;
        return ns;
    };
    if (forced) { p.nsplit = (int)splits(forced); return p; }
    if (Fs <= 0) { p.nsplit = (int)splits(WGRAD_MAX_ROUNDS); return p; }
    const double steps = (double)p.jtiles * Fs;
    double best_cost = 0;
    p.nsplit = 1;
    for (int r = 1; r <= WGRAD_MAX_ROUNDS; ++r) {
        const long long ns = splits(r), wgs = tiles * ns, rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * (steps / (double)ns + 16.0);
        if (r == 1 || cost < best_cost) { best_cost = cost; p.nsplit = (int)ns; }
    }
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
