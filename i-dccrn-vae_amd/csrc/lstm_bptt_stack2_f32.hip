// Back-propagation through time of BOTH layers of the H = 128 complex LSTM (DCCRN-CL bottleneck; what torch.autograd runs for
// nn.LSTM(num_layers = 2) behind `loss.backward()`, reference model/complex_progress.py:50-74, supervised_dccrn/train.py:239-243)
// in ONE cooperative launch -- the backward twin of lstm_stack2_f32.hip.
//
// lstm_bptt_coop_f32.hip runs one layer as T latency-bound steps on 4 CUs per (run, 16-sequence tile); the per-layer form needs
// layer 1's gate gradients dA1 complete before the hoisted GEMM dh0 = dA1 W_ih1 and then layer 0's BPTT start.  Layer 0's step
// for time t only needs dA1[t], so here it runs ONE STEP BEHIND layer 1 on its own 4 CUs per (run, tile):
//   * the layer-1 workgroups are those of lstm_bptt_coop_f32.hip, except that their gate gradients go to the G1 buffer (which
//     the weight gradients read afterwards anyway) as 16-byte write-through (sc1) stores; they never wait for layer 0;
//   * a layer-0 workgroup keeps the slice of W_ih1^T for its 32 hidden units next to its W_hh0^T slice in registers (2 x 64
//     VGPRs per lane), requests the rows of dA1[t-1] BEFORE its own stores and arrive of step t (layer 1 is ahead: its arrival
//     count is cached and polled only when the cached value does not cover the step) and contracts dA1[t] W_ih1 while the loads
//     of its own dA0[t+1] are in flight; the dh0 buffer and its GEMM disappear.
// Synchronisation: the fence-free hand-off of lstm_pers.hip (sc1 stores drained by every storing wave, one agent-scope atomic
// add per workgroup and counter replica behind the workgroup barrier, an sc1 poll, sc1 loads behind poll + barrier); all
// 8 x 4 x tiles workgroups resident (one per CU); bounded spins, NaN poison and the sticky status word on time-out (coop.hpp).
#include <cstdlib>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_bstack2 {

typedef int v4i __attribute__((ext_vector_type(4)));

struct Args {
    float* g1;            // layer 1: activated gates in, gate gradients out; [run][T*B][4H] (run stride 4*T*B*H, ld 4H)
    unsigned g1_bytes;
    float* g0;            // layer 0: the same with the addressing of its forward (g_run_z, g_run_s, ldg)
    long long g0_run_z, g0_run_s;
    int ldg0;
    const float* c1;      // [4][T*B][H] cell states
    const float* c0;
    const float* dhout;   // [4][T*B][H] gradient arriving at layer 1's h_t from above
    const float* whhT1;   // idv_pack_lstm_hh_bwd fragments: [set][unit tile][blk][lane][4]
    const float* wihT1;   // W_ih of layer 1 in the same order (idv_pack_lstm_hh_bwd on weight_ih_l1)
    const float* whhT0;
    float* hx;            // exchange [2 layers][2 parity][4 runs][Bpad][4H] fp32: dA_t, row-major in the gate-column order
    unsigned hx_bytes;
    unsigned* sync;       // [abort flag: 256 B][layer slot][group = run * tiles + tile][replica][256 B]
    int nrep;
    int B, T, Bpad, tiles;
    unsigned* status;
    int fault;
};

constexpr int H = 128, NSL = 4;
constexpr unsigned long long SPIN_LIMIT_TICKS = 40000000ull;     // 0.4 s of the 100 MHz wall clock
constexpr int MAX_GROUPS = 64, MAX_REP = 8;
constexpr int SYNC_BYTES = 256 + 2 * MAX_GROUPS * MAX_REP * 256;

// thread 0: wait until *counter >= want (bounded); (other, other_seen): a second counter read ONCE if the first check fails
__device__ __forceinline__ void wait_for(unsigned* counter, unsigned want, unsigned* abortf, int* abort_sh, unsigned* seen = nullptr,
                                         unsigned* other = nullptr, unsigned* other_seen = nullptr) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long spins = 0;
    unsigned v;
    while ((v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
        if (other) {
            *other_seen = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            other = nullptr;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023) == 0) {
            if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { *abort_sh = 1; break; }
            if (wall_clock64() - t0 > SPIN_LIMIT_TICKS) {
                __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *abort_sh = 1;
                break;
            }
        }
    }
    if (seen) *seen = v;
}

__global__ __launch_bounds__(256, 1) void lstm_bptt_stack2_f32_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float red[];                // [4 waves][2 tiles][4 r][64 lanes]
    __shared__ int abort_sh;
    __shared__ unsigned seen1_sh;                                              // layer 0: layer-1 arrivals last observed
    __shared__ __attribute__((aligned(16))) float stage[16][128];              // dA_t of this workgroup: [row][its 128 colp]
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, a.hx_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g1r = __builtin_amdgcn_make_buffer_rsrc((void*)a.g1, 0, a.g1_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int low = blockIdx.x >> 2;              // 0: layer 1 (runs ahead), 1: layer 0 (one step behind)
    const int sl = blockIdx.x & 3, run = blockIdx.y, tile = blockIdx.z;
    const int z = run >> 1, s = run & 1;
    const int col = lane & 15, rq = lane >> 4;
    const int b0 = tile * 16;
    unsigned* abortf = a.sync;
    const int groups = 4 * a.tiles, grp = run * a.tiles + tile;
    unsigned* cntA = a.sync + 64 + (size_t)((0 * groups + grp) * a.nrep) * 64;          // layer-1 arrivals of this (run, tile)
    unsigned* cntB = a.sync + 64 + (size_t)((1 * groups + grp) * a.nrep) * 64;          // layer-0 arrivals
    unsigned* mine = low ? cntB : cntA;
    const int rep = sl & (a.nrep - 1);
    const size_t TB = (size_t)a.T * a.B, TBH = TB * H;
    // this layer's gate buffer, cell states and incoming gradient
    float* g = low ? a.g0 + z * a.g0_run_z + s * a.g0_run_s : a.g1 + (size_t)run * TB * 4 * H;
    const int ldg = low ? a.ldg0 : 4 * H;
    const float* cst = (low ? a.c0 : a.c1) + (size_t)run * TBH;
    const float* dho = a.dhout + (size_t)run * TBH;          // layer 1 only
    const unsigned hx_layer = (unsigned)low * 2u * 4u * (unsigned)a.Bpad * 512u * 4u;
    const unsigned hx_par = 4u * (unsigned)a.Bpad * 512u * 4u;

    // W_hh^T slice (and, layer 0, the W_ih1^T slice): output units of tiles 2 sl, 2 sl + 1; this wave's 128 gate columns k in
    // four chunks of 32; lane (row, kq) holds k = 128 w + 32 c + 8 kq + j as MFMA k-step j of chunk c -- as lstm_bptt_coop_f32.hip
    float breg[2][4][8], ireg[2][4][8];
    {
        const float* wb = (low ? a.whhT0 : a.whhT1) + (size_t)s * (H / 16) * (H / 4) * 64 * 4;
        const float* wi = a.wihT1 + (size_t)s * (H / 16) * (H / 4) * 64 * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 128 * wave + 32 * c + 8 * rq + j;
                    const size_t e = ((((size_t)(2 * sl + u) * (H / 4) + (k >> 4)) * 64 + (k & 3) * 16 + col) << 2) + ((k >> 2) & 3);
                    breg[u][c][j] = wb[e];
                    ireg[u][c][j] = low ? wi[e] : 0.f;
                }
    }
    const int myrow = rq * 4 + wave;
    const int brow = b0 + myrow;
    const bool ok = brow < a.B;
    const int bclamp = ok ? brow : a.B - 1;
    const int lrow = (b0 + col < a.B) ? b0 + col : a.B - 1;       // the row this lane supplies to the A operand (clamped)
    float dcreg[2] = {0.f, 0.f};

    bool aborted = false;
    if (tid == 0) { abort_sh = 0; seen1_sh = 0; }
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    __syncthreads();

    // layer 0: rows of dA1 for time t (all 4H gate columns of this (run, tile); this wave's 128 of them) from the G1 buffer
    auto in_known = [&](int step) -> bool { return seen1_sh >= (unsigned)(step + 1) * (unsigned)NSL; };       // uniform
    auto in_load = [&](int t, f32x4 (&v)[4][2]) {
        const unsigned off = (unsigned)((((size_t)run * TB + (size_t)t * a.B + lrow) * 512u + 128 * wave + 8 * rq) * 4u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g1r, off + c * 128u, 0, 16));
            v[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g1r, off + c * 128u + 16u, 0, 16));
        }
    };
    auto in_late = [&](int step, int t, f32x4 (&v)[4][2]) -> bool {            // poll layer 1, then load
        if (tid == 0) wait_for(cntA + (size_t)rep * 64, (unsigned)(step + 1) * (unsigned)NSL, abortf, &abort_sh, &seen1_sh);
        __syncthreads();
        if (abort_sh) return false;
        in_load(t, v);
        return true;
    };
    f32x4 cur[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) cur[c][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (low && !in_late(0, a.T - 1, cur)) aborted = true;

    for (int t = a.T - 1; t >= 0 && !aborted; --t) {
        const int step = a.T - 1 - t;                // 0, 1, ...
        // inputs of the cell backward (independent of the contraction): issue first
        float gi[2], gf[2], gg_[2], go[2], cc[2], cp[2], dhv[2];
        const size_t row = (size_t)t * a.B + bclamp;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ub = 2 * sl + u, unit = ub * 16 + col;
            const float* gp = g + row * ldg + ub * 64 + col;
            gi[u] = gp[0]; gf[u] = gp[16]; gg_[u] = gp[32]; go[u] = gp[48];
            cc[u] = cst[row * H + unit];
            cp[u] = (t > 0) ? cst[(row - a.B) * H + unit] : 0.f;
            dhv[u] = (!low && ok) ? dho[row * H + unit] : 0.f;
        }
        f32x4 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][r] = 0.f;
        auto contract = [&](const f32x4 (&v)[4][2], const float (&w)[2][4][8]) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 h4 = v[c][j >> 2];
                    const float aj = (j & 3) == 0 ? h4[0] : ((j & 3) == 1 ? h4[1] : ((j & 3) == 2 ? h4[2] : h4[3]));
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, w[u][c][j], acc[u], 0, 0, 0);
                }
        };
        if (low && step == 0) contract(cur, ireg);             // dh0[T-1] = dA1[T-1] W_ih1 (no recurrent term yet)
        if (step > 0) {
            if (tid == 0)       // (layer 0: the view of layer 1's progress is refreshed while waiting for the siblings, if it waits)
                wait_for(mine + (size_t)rep * 64, (unsigned)step * (unsigned)NSL, abortf, &abort_sh, nullptr,
                         low ? cntA + (size_t)rep * 64 : nullptr, &seen1_sh);
            __syncthreads();                 // the polling wave joins after its match; every load below is sc1
            if (abort_sh) { aborted = true; break; }
            // dA_{t+1} of this layer's group: row = lane & 15, 8 consecutive gate columns per lane and chunk
            const unsigned par_r = hx_layer + (unsigned)((step - 1) & 1) * hx_par;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + col) * 512u + 128 * wave + 8 * rq) * 4u;
            f32x4 av[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                av[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + c * 128u, par_r, 16));
                av[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + c * 128u + 16u, par_r, 16));
            }
            if (low) contract(cur, ireg);    // dA1[t] W_ih1 under the latency of the loads above
            contract(av, breg);
        }
        // layer 0: the rows of dA1[t-1] for the next step, if layer 1 is known to have published them (it usually is)
        const bool early = low && t > 0 && in_known(step + 1);
        f32x4 nx[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) nx[c][h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (early) in_load(t - 1, nx);
        // ---- reduce the 4 K-partials through LDS: [wave][tile][r][lane]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * 2 + u) * 4 + r) << 6) + lane] = acc[u][r];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float dhr = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) dhr += red[(((w * 2 + u) * 4 + wave) << 6) + lane];
            const float dh = dhv[u] + dhr;
            const float tc = tanhf_(cc[u]);
            const float d_o = dh * tc;
            const float dc = dcreg[u] + dh * go[u] * (1.f - tc * tc);
            const float d_i = dc * gg_[u], d_g = dc * gi[u], d_f = dc * cp[u];
            float ai = d_i * gi[u] * (1.f - gi[u]);
            float af = d_f * gf[u] * (1.f - gf[u]);
            float ag = d_g * (1.f - gg_[u] * gg_[u]);
            float ao = d_o * go[u] * (1.f - go[u]);
            dcreg[u] = dc * gf[u];
            if (!ok) ai = af = ag = ao = 0.f;            // padded rows publish zeros
            if (low && ok) {                             // layer 0: the gate gradients replace the saved gates, plain stores
                float* gp = g + ((size_t)t * a.B + brow) * ldg + (2 * sl + u) * 64 + col;
                gp[0] = ai; gp[16] = af; gp[32] = ag; gp[48] = ao;
            }
            float* sp = &stage[myrow][u * 64 + col];
            sp[0] = ai; sp[16] = af; sp[32] = ag; sp[48] = ao;
        }
        __syncthreads();
        {
            // 16 rows x 32 float4 = 512 stores, two per thread: write-through (sc1) to the exchange buffer and, layer 1, to the
            // G1 rows the layer-0 workgroups of this (run, tile) read (and the weight gradients afterwards)
            const unsigned par_w = hx_layer + (unsigned)(step & 1) * hx_par;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = tid + 256 * h, r = e >> 5, c4 = e & 31;
                const v4i pk = *(const v4i*)&stage[r][c4 * 4];
                const unsigned off = (((unsigned)run * a.Bpad + b0 + r) * 512u + sl * 128 + c4 * 4) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
                if (!low && b0 + r < a.B) {
                    const unsigned o1 = (unsigned)((((size_t)run * TB + (size_t)t * a.B + b0 + r) * 512u + sl * 128 + c4 * 4) * 4u);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, g1r, o1, 0, 16);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < a.nrep) __hip_atomic_fetch_add(mine + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // layer 0: the next step's input rows, the late way, if they were not requested above
        if (low && t > 0) {
            if (!early && !in_late(step + 1, t - 1, nx)) aborted = true;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) cur[c][h] = nx[c][h];
        }
    }
    if (aborted) {
        // poison this workgroup's gate gradients: a timed-out BPTT must never look like a result
        if (tid == 0) idv_coop_raise(a.status);
        const float qnan = __builtin_nanf("");
        for (long long e = tid; e < (long long)a.T * 16 * 128; e += 256) {
            const int cidx = (int)(e & 127), br = (int)((e >> 7) & 15);
            const long long t = e >> 11;
            if (b0 + br < a.B) g[((size_t)t * a.B + b0 + br) * ldg + sl * 128 + cidx] = qnan;
        }
    }
}

}  // namespace idv_bstack2

extern "C" int idv_lstm_bptt_stack2_supported(int H, int B) {
    static const bool on = [] { const char* e = getenv("IDV_LSTM_BPTT_STACK2"); return !e || e[0] != '0'; }();
    if (!on || H != 128 || B <= 0) return 0;
    const int tiles = (B + 15) / 16;
    return 4 * tiles <= idv_bstack2::MAX_GROUPS && 2 * 4 * 4 * tiles <= idv_coop_max_workgroups();      // all resident, one per CU
}

extern "C" long long idv_lstm_bptt_stack2_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return idv_bstack2::SYNC_BYTES + 2LL * 2 * 4 * Bpad * 4 * H * 4;
}

// BPTT of both H = 128 layers in one cooperative launch.  g1: [run][T*B][4H] activated gates of layer 1 -> its gate gradients;
// g0 (+ g0_run_z, g0_run_s, ldg0): layer 0's, addressed as in its forward; c1, c0: cell states; dhout1: gradient arriving at
// layer 1's output; whhT1 / wihT1 / whhT0: idv_pack_lstm_hh_bwd fragments of weight_hh_l1, weight_ih_l1, weight_hh_l0;
// work: idv_lstm_bptt_stack2_work_bytes bytes, 16-byte aligned.
extern "C" int idv_lstm_bptt_stack2(float* g1, float* g0, long long g0_run_z, long long g0_run_s, int ldg0, const float* c1,
                                    const float* c0, const float* dhout1, const float* whhT1, const float* wihT1, const float* whhT0,
                                    int H, int B, int T, void* work, void* stream) {
    using namespace idv_bstack2;
    if (!g1 || !g0 || !c1 || !c0 || !dhout1 || !whhT1 || !wihT1 || !whhT0 || !work || T <= 0 || ldg0 < 4 * H ||
        !idv_lstm_bptt_stack2_supported(H, B))
        return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(work) & 15) || (reinterpret_cast<uintptr_t>(g1) & 15)) return IDV_EINVAL;
    const long long g1_bytes = 16LL * T * B * H * 4;
    if (g1_bytes >= 0xfffffe00LL) return IDV_EINVAL;          // 32-bit buffer offsets
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (B + 15) / 16;
    const long long Bpad = 16LL * tiles;
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    Args a{};
    a.g1 = g1; a.g1_bytes = (unsigned)g1_bytes;
    a.g0 = g0; a.g0_run_z = g0_run_z; a.g0_run_s = g0_run_s; a.ldg0 = ldg0;
    a.c1 = c1; a.c0 = c0; a.dhout = dhout1;
    a.whhT1 = whhT1; a.wihT1 = wihT1; a.whhT0 = whhT0;
    a.sync = (unsigned*)work;
    a.hx = (float*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)(2LL * 2 * 4 * Bpad * 4 * H * 4);
    a.nrep = 4;
    a.B = B; a.T = T; a.Bpad = (int)Bpad; a.tiles = tiles;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    a.status = idv_coop_status_word();
    const size_t smem = 84 * 1024;                   // one workgroup per CU
    if (hipFuncSetAttribute((const void*)lstm_bptt_stack2_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(lstm_bptt_stack2_f32_kernel, dim3(2 * NSL, 4, tiles), dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
