// Host-side configuration choice + launch of cgemm_kernel, and the C-ABI entry points built on it.
#include <cstdlib>
#include "cgemm.hpp"
#include "../../include/idccrn_hip.h"

int idv_launch_ctconv_c1_f32(const CgemmArgs& a, hipStream_t st);      // ctconv_c1_f32.hip

namespace {

template <int MODE, int WM, int WN, int MT_W, int FO_T, int JC_W, int CCK, bool SWAP, bool STATS, bool VEC>
int launch_vec(const CgemmArgs& a, hipStream_t st) {
    using G = CgemmGeom<MODE, FO_T>;
    constexpr int JT = 32 * JC_W * WN;
    constexpr int NE = CCK * G::FR * (VEC ? JT + 8 : JT + 2);
    constexpr size_t smem = 2 * NE * sizeof(float);
    const int rows = (MODE == IDV_TCONV) ? a.Fin : a.Fout;          // TCONV tiles over input rows m
    CgemmArgs b = a;
    b.jtiles = (a.J + JT - 1) / JT;
    b.ftiles = (rows + FO_T - 1) / FO_T;
    b.mblocks = ((a.M + 31) / 32 + WM * MT_W - 1) / (WM * MT_W);
    const long long tiles = (long long)b.jtiles * b.ftiles;
    // transposed conv: all frequency tiles of a column block on ONE XCD (they share 2 of their 5 input rows through its L2):
    // -27 % L2-miss traffic on dec1-3 at unchanged speed; the conv mode lost 1 % with it and keeps the tile-major order
    static const bool map_ft = [] { const char* e = getenv("IDV_MAP_FT"); return !e || e[0] != '0'; }();
    b.map_ft = (map_ft && MODE == IDV_TCONV) ? 1 : 0;
    const long long nblk = b.map_ft ? (long long)((b.jtiles + 7) / 8) * 8 * b.ftiles * b.mblocks : ((tiles + 7) / 8) * 8 * b.mblocks;
    if (nblk > 0x7fffffffLL) return IDV_EINVAL;
    dim3 grid((unsigned)nblk);
    auto k = cgemm_kernel<MODE, WM, WN, MT_W, FO_T, JC_W, CCK, SWAP, STATS, VEC>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return IDV_ELAUNCH;
    }
    hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), smem, st, b);
    return idv_launch_status();
}

// vector staging needs 16-byte aligned rows and the same column mapping for both sources
template <int MODE, int WM, int WN, int MT_W, int FO_T, int JC_W, int CCK, bool SWAP, bool STATS>
int launch_cfg(const CgemmArgs& a, hipStream_t st) {
    const bool vec = (a.Jp % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x0) & 15) == 0) &&
                     (a.C1 == 0 || (a.x1_div == 1 && a.Jp1 == a.Jp && (reinterpret_cast<uintptr_t>(a.x1) & 15) == 0)) &&
                     (MODE == IDV_PW || (2 * a.C0) % CCK == 0);
    if (vec) return launch_vec<MODE, WM, WN, MT_W, FO_T, JC_W, CCK, SWAP, STATS, true>(a, st);
    return launch_vec<MODE, WM, WN, MT_W, FO_T, JC_W, CCK, SWAP, STATS, false>(a, st);
}

inline int waste(int n, int t) { return ((n + t - 1) / t) * t - n; }

const bool USE_C1_F32 = [] { const char* e = getenv("IDV_C1_F32"); return !e || e[0] != '0'; }();

// configuration id = the template arguments <MODE, WM, WN, MT_W, FO_T, JC_W, CCK> as decimal digits
int conv_config(int mode, int CC, int M, int rows) {
    const bool fo5 = waste(rows, 5) <= waste(rows, 3);
    if (mode == IDV_CONV) {
        if (CC % 4 != 0) return 221332;
        return fo5 ? 221524 : 221334;
    }
    if (CC % 4 != 0) return -1;
    if (M <= 32) return 1141314;
    return fo5 ? 1221514 : 1221324;
}

template <bool STATS>
int launch_conv(const CgemmArgs& a, int mode, hipStream_t st) {
    const int rows = (mode == IDV_TCONV) ? a.Fin : a.Fout;
    switch (conv_config(mode, 2 * (a.C0 + a.C1), a.M, rows)) {
        case 221332: return launch_cfg<IDV_CONV, 2, 2, 1, 3, 3, 2, false, STATS>(a, st);
        case 221524: return launch_cfg<IDV_CONV, 2, 2, 1, 5, 2, 4, false, STATS>(a, st);
        case 221334: return launch_cfg<IDV_CONV, 2, 2, 1, 3, 3, 4, false, STATS>(a, st);
        case 1141314: return launch_cfg<IDV_TCONV, 1, 4, 1, 3, 1, 4, false, STATS>(a, st);
        case 1221514: return launch_cfg<IDV_TCONV, 2, 2, 1, 5, 1, 4, false, STATS>(a, st);
        case 1221324: return launch_cfg<IDV_TCONV, 2, 2, 1, 3, 2, 4, false, STATS>(a, st);
        default: return IDV_EINVAL;
    }
}

}  // namespace

extern "C" int idv_cconv_cck(int cin_used) { return ((2 * cin_used) % 4 == 0) ? 4 : 2; }

extern "C" int idv_cconv_config(int transposed, int cin_used, int Cout, int Fin) {
    const int rows = transposed ? Fin : (Fin - 1) / 2 + 1;
    if (transposed && Cout == 1 && USE_C1_F32) return 1000001;          // ctconv_c1_f32_kernel (vector ALU)
    return conv_config(transposed ? IDV_TCONV : IDV_CONV, 2 * cin_used, 2 * Cout, rows);
}

extern "C" int idv_cconv2d_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div,
                               const float* wfrag, const float* bias, const float* prelu_slope, float* out,
                               double* stats, double* stats_work, int stats_rep, int transposed, int tshift, int Cout, int Fin,
                               int B, int Tp, int Jp, int t_valid_out, void* stream) {
    if (!x0 || !wfrag || !bias || !out || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    if (stats && stats_work && (stats_rep < 2 || (stats_rep & (stats_rep - 1)))) return IDV_EINVAL;
    if (C1 > 0 && (!x1 || x1_div < 1)) return IDV_EINVAL;
    if (tshift != 0 && tshift != -1) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp1; a.x1_div = x1_div < 1 ? 1 : x1_div;
    a.wfrag = wfrag; a.bias = bias; a.slope = prelu_slope; a.out = out;
    a.M = 2 * Cout; a.Mtiles = (a.M + 31) / 32; a.cplx_rows = 1; a.Cout = Cout;
    a.tshift = tshift; a.t_valid = t_valid_out; a.stats = stats; a.ldo = 0; a.nB = B;
    if (Jp < a.J) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int mode = transposed ? IDV_TCONV : IDV_CONV;
    // one output channel (last decoder block): 2 of the 32 MFMA rows would work; memory-shaped vector-ALU kernel instead
    if (transposed && Cout == 1 && USE_C1_F32) return idv_launch_ctconv_c1_f32(a, st);
    if (!stats) return launch_conv<false>(a, mode, st);
    if (!stats_work) return launch_conv<true>(a, mode, st);
    a.stats = stats_work; a.stats_rep = stats_rep;
    const int rc = launch_conv<true>(a, mode, st);
    return rc ? rc : idv_launch_stats_collapse(stats_work, stats_rep, Cout * 5, stats, st);
}

// idv_cconv2d_fwd (eval mode, x1_div == 1) that can also / instead write its result as a split-bf16 image
extern "C" int idv_cconv2d_fwd_img(const float* x0, int C0, const float* x1, int C1, int Jp1, const float* wfrag,
                                   const float* bias, const float* prelu_slope, float* out_planar, void* out_img,
                                   long long out_lo_off, int transposed, int tshift, int Cout, int Fin, int B, int Tp,
                                   int Jp, int t_valid_out, void* stream) {
    if (!x0 || !wfrag || !bias || (!out_planar && !out_img) || C0 <= 0 || Cout <= 0 || Fin <= 0 || B <= 0 || Tp <= 1)
        return IDV_EINVAL;
    if ((C1 > 0 && !x1) || (tshift != 0 && tshift != -1)) return IDV_EINVAL;
    if (out_img && ((Cout % 4) || (out_lo_off % 8) || (reinterpret_cast<uintptr_t>(out_img) & 15))) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.Fin = Fin;
    a.Fout = transposed ? 2 * Fin - 1 : (Fin - 1) / 2 + 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = Jp1; a.x1_div = 1;
    a.wfrag = wfrag; a.bias = bias; a.slope = prelu_slope; a.out = out_planar;
    a.out_img = out_img; a.out_lo_off = out_lo_off;
    a.M = 2 * Cout; a.Mtiles = (a.M + 31) / 32; a.cplx_rows = 1; a.Cout = Cout;
    a.tshift = tshift; a.t_valid = t_valid_out; a.stats = nullptr; a.ldo = 0; a.nB = B;
    if (Jp < a.J) return IDV_EINVAL;
    return launch_conv<false>(a, transposed ? IDV_TCONV : IDV_CONV, (hipStream_t)stream);
}

extern "C" int idv_pw_gemm(const float* x, int K, const float* wfrag, const float* bias, const float* prelu_slope,
                           float* out, int M, int B, int Tp, int Jp, int t_valid, int swap, int ldo, void* stream) {
    if (!x || !wfrag || !bias || !out || K <= 0 || (K & 1) || M <= 0 || B <= 0 || Tp <= 1) return IDV_EINVAL;
    CgemmArgs a{};
    a.x0 = x; a.x1 = nullptr; a.C0 = K / 2; a.C1 = 0;
    a.Fin = 1; a.Fout = 1;
    a.J = B * Tp; a.Jp = Jp; a.Tp = Tp; a.Jp1 = 0; a.x1_div = 1;
    a.wfrag = wfrag; a.bias = bias; a.slope = prelu_slope; a.out = out;
    a.M = M; a.Mtiles = (M + 31) / 32; a.cplx_rows = 0; a.Cout = M;
    a.tshift = 0; a.t_valid = t_valid; a.stats = nullptr; a.ldo = ldo; a.nB = B;
    if (Jp < a.J) return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // IDV_PW_CFG (experiments): 1 = four column tiles per wave, 2 = four row tiles per wave (rows in whole 256-row blocks only:
    // the fragment buffer is allocated in 128-row blocks)
    static const int pwcfg = [] { const char* e = getenv("IDV_PW_CFG"); return e ? atoi(e) : 0; }();
    if (pwcfg == 1) {
        if (swap) return launch_cfg<IDV_PW, 2, 2, 2, 1, 4, 8, true, false>(a, st);
        return launch_cfg<IDV_PW, 2, 2, 2, 1, 4, 8, false, false>(a, st);
    }
    if (pwcfg == 2 && a.Mtiles % 8 == 0) {
        if (swap) return launch_cfg<IDV_PW, 2, 2, 4, 1, 2, 8, true, false>(a, st);
        return launch_cfg<IDV_PW, 2, 2, 4, 1, 2, 8, false, false>(a, st);
    }
    // four row-tile pairs x ONE column group per workgroup where the rows come in whole 256-row blocks (the LSTM input
    // projections: M = 8H): a 72-column patch per workgroup instead of 136, as for the wide transposed conv (cgemm_gauss.hip):
    // M = 1024 / 3072 / 6144, K = 1280, B = 64: 1.20 / 3.08 / 5.88 -> 1.16 / 2.97 / 5.60 ms.  IDV_PW_CFG=7 keeps 2 x 2.
    // experiments: 16 / 32 planes per K chunk (a quarter of the barriers per MFMA); K in whole chunks only
    if ((pwcfg == 16 || pwcfg == 32) && a.Mtiles % 8 == 0 && K % pwcfg == 0) {
        if (pwcfg == 16) {
            if (swap) return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 16, true, false>(a, st);
            return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 16, false, false>(a, st);
        }
        if (swap) return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 32, true, false>(a, st);
        return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 32, false, false>(a, st);
    }
    if (pwcfg != 7 && a.Mtiles % 8 == 0) {
        if (swap) return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 8, true, false>(a, st);
        return launch_cfg<IDV_PW, 4, 1, 2, 1, 2, 8, false, false>(a, st);
    }
    if (swap) return launch_cfg<IDV_PW, 2, 2, 2, 1, 2, 8, true, false>(a, st);
    return launch_cfg<IDV_PW, 2, 2, 2, 1, 2, 8, false, false>(a, st);
}
