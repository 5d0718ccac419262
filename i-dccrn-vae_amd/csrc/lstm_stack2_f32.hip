// Both layers of the DCCRN-CL bottleneck LSTM (H = 128, exact fp32; reference ComplexLSTM.forward, model/complex_progress.py:50-74:
// two nn.LSTM(num_layers = 2), each applied to the real and to the imaginary input) in ONE cooperative launch (evaluation, and
// the training forward: the activated gates and cell states idv_lstm_bptt reads are kept).
//
// lstm_coop_f32.hip runs a layer as 641 latency-bound steps of 3.3 us on 64 CUs and needs layer 0 finished before layer 1's
// input projection (a separate GEMM) and recurrence start: 2 x 2.1 ms + 0.6 ms per forward, 5 % of the fp32 headline step
// with 3/4 of the chip idle.  Here layer 1 runs ONE STEP BEHIND layer 0 on its own 4 CUs per (run, 16-sequence tile):
//   * layer-0 workgroups are those of lstm_coop_f32.hip; they additionally publish h0[t] with write-through (sc1) stores
//     (the rows of the [run][T*B][H] buffer the per-layer path writes anyway) and never wait for layer 1;
//   * a layer-1 workgroup keeps its W_ih slice next to its W_hh slice in registers (2 x 64 VGPRs per lane) and computes
//     W_ih h0[t+1] -- an input that layer 0 finished a step earlier -- while its siblings' h1[t] is still in flight, i.e. in
//     the hand-off latency the step is bound by; the hoisted layer-1 projection GEMM disappears.
// Synchronisation is the fence-free hand-off of lstm_pers.hip / lstm_coop_f32.hip (sc1 16-byte stores drained by every
// storing wave, one agent-scope atomic add per workgroup and counter replica behind the workgroup barrier, an sc1 poll,
// sc1 loads behind poll + barrier); all 8 x 4 x tiles workgroups must be resident (one per CU: 84 KB of LDS requested,
// idv_lstm_stack2_f32_supported checks the count against idv_coop_max_workgroups()); bounded spins, NaN poison and the sticky
// status word on time-out (coop.hpp).
#include <cstdlib>
#include <mutex>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_stack2 {

typedef int v4i __attribute__((ext_vector_type(4)));

struct Args {
    const float* g;           // layer-0 gate pre-activations (hoisted input projection), addressing as lstm.hip RecArgs
    long long g_run_z, g_run_s;
    int ldg;
    const float* whh0;        // idv_pack_lstm_hh fp32 fragments: [set][tile = ub*4 + gate][kk][lane]
    const float* wih1;        // W_ih of layer 1 ([4H][H] as well) in the same fragment order
    const float* whh1;
    const float* bias1;       // [2 sets][4H], gate-column order (idv_pack_lstm_ih): b_ih + b_hh of layer 1
    float* h0;                // [4 runs][T*B][H]: layer-0 output, read by the layer-1 workgroups
    unsigned h0_bytes;
    float* hout;              // [4 runs][T*B][H]: layer-1 output
    float* gsave1;            // training forward: activated gates of layer 1 [run][T*B][4H] (layer 0: over g, in place) ...
    float* csave0;            // ... and the cell states [4 runs][T*B][H] of both layers; all three NULL in evaluation
    float* csave1;
    float* hx;                // exchange [2 layers][2 parity][4 runs][Bpad][H] fp32
    unsigned hx_bytes;
    unsigned* sync;           // [abort flag: 256 B][layer][group = run * tiles + tile][replica][256 B]
    int nrep;
    int B, T, Bpad, tiles;
    unsigned* status;         // host-mapped sticky status word (coop.hpp) or nullptr
    int fault;                // test hook (IDV_COOP_FAULT=1): workgroup (0, 0, 0) never arrives -> the bounded spins must abort
};

constexpr int H = 128, NSL = 4, UPW = 32;          // hidden size, workgroups per (layer, group), units per workgroup
constexpr unsigned long long SPIN_LIMIT_TICKS = 40000000ull;     // 0.4 s of the 100 MHz wall clock
constexpr int MAX_GROUPS = 64, MAX_REP = 8;
constexpr int SYNC_BYTES = 256 + 2 * MAX_GROUPS * MAX_REP * 256;

// thread 0 of the workgroup: wait until *counter >= want (bounded); 1 in *abort_sh when the launch is being abandoned;
// *seen (if given) <- the last value read
// (other, other_seen): a second counter read ONCE if the first check fails, i.e. only when there is time to spare
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

__global__ __launch_bounds__(256, 1) void lstm_stack2_f32_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float red[];                // [4 waves][8 tiles][4 r][64 lanes]
    __shared__ int abort_sh;
    __shared__ unsigned seen0_sh;                                              // layer 1: layer-0 arrivals last observed
    __shared__ __attribute__((aligned(16))) float stage[16][UPW];              // h_t of this workgroup: [row][unit]
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, a.hx_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t h0r = __builtin_amdgcn_make_buffer_rsrc((void*)a.h0, 0, a.h0_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int layer = blockIdx.x >> 2, sl = blockIdx.x & 3, run = blockIdx.y, tile = blockIdx.z;
    const int z = run >> 1, s = run & 1;
    const int col = lane & 15, rq = lane >> 4;
    const int b0 = tile * 16;
    unsigned* abortf = a.sync;
    const int groups = 4 * a.tiles, grp = run * a.tiles + tile;
    unsigned* cnt0 = a.sync + 64 + (size_t)((0 * groups + grp) * a.nrep) * 64;          // layer-0 arrivals of this (run, tile)
    unsigned* cnt1 = a.sync + 64 + (size_t)((1 * groups + grp) * a.nrep) * 64;
    unsigned* mine = layer ? cnt1 : cnt0;
    const int rep = sl & (a.nrep - 1);
    const size_t TB = (size_t)a.T * a.B, TBH = TB * H;
    // exchange regions: [layer][parity][run][Bpad][H]
    const unsigned hx_layer = (unsigned)layer * 2u * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
    const unsigned hx_par = 4u * (unsigned)a.Bpad * (unsigned)H * 4u;

    // W_hh slice (and, layer 1, the W_ih slice): this workgroup's 8 column tiles (unit blocks 2 sl, 2 sl + 1 x 4 gates), this
    // wave's 32 k, 8 consecutive k per lane (MFMA k-step j of lane (row, kq) is k = 32 w + 8 kq + j) -- as lstm_coop_f32.hip
    float breg[8][8], ireg[8][8];
    {
        const float* wb = (layer ? a.whh1 : a.whh0) + (size_t)s * 4 * H * H;
        const float* wi = a.wih1 + (size_t)s * 4 * H * H;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int ctile = (2 * sl + (t8 >> 2)) * 4 + (t8 & 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * wave + 8 * rq + j;
                const size_t e = ((size_t)ctile * (H / 4) + (k >> 2)) * 64 + (k & 3) * 16 + col;
                breg[t8][j] = wb[e];
                ireg[t8][j] = layer ? wi[e] : 0.f;
            }
        }
    }
    // cell update split by row over the waves: wave w owns rows rq * 4 + w of both unit blocks (unit = 16 ub + lane & 15)
    const int myrow = rq * 4 + wave;
    const int brow = b0 + myrow;
    const bool rowok = brow < a.B;
    const int bclamp = rowok ? brow : a.B - 1;
    const int lrow = (b0 + col < a.B) ? b0 + col : a.B - 1;       // the row this lane supplies to the A operand (clamped)
    float creg[2] = {0.f, 0.f};
    float gb[2][4];                                                // layer 1: the constant part of the gate pre-activations
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) gb[ub][gg] = layer ? a.bias1[(size_t)s * 4 * H + (2 * sl + ub) * 64 + 16 * gg + col] : 0.f;
    const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
    // training forward: where this workgroup keeps its activated gates (row pitch gld) and cell states
    float* gsv = !a.csave0 ? nullptr : (layer ? a.gsave1 + (size_t)run * TB * 4 * H : const_cast<float*>(g));
    const int gld = layer ? 4 * H : a.ldg;
    float* csv = layer ? a.csave1 : a.csave0;

    bool aborted = false;
    if (tid == 0) { abort_sh = 0; seen0_sh = 0; }
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    __syncthreads();

    // layer 1: the rows of h0[t] for the CURRENT step sit in registers (cur0, cur1) when the step starts; W_ih h0[t] is
    // contracted while the loads of h1[t-1] are in flight, so it costs no time of its own.  Layer 0 runs ahead (its step is
    // shorter): its arrival count is polled only when the cached one does not cover the step, and the rows of h0[t+1] are
    // requested EARLY -- before this step's own stores and arrive.
    auto input_known = [&](int tt) -> bool { return seen0_sh >= (unsigned)(tt + 1) * (unsigned)NSL; };       // uniform
    auto input_load = [&](int tt, f32x4& a0, f32x4& a1) {
        const unsigned off = (unsigned)((((size_t)run * TB + (size_t)tt * a.B + lrow) * H + 32 * wave + 8 * rq) * 4u);
        a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h0r, off, 0, 16));
        a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h0r, off + 16u, 0, 16));
    };
    auto input_late = [&](int tt, f32x4& a0, f32x4& a1) -> bool {            // poll layer 0, then load
        if (tid == 0) wait_for(cnt0 + (size_t)rep * 64, (unsigned)(tt + 1) * (unsigned)NSL, abortf, &abort_sh, &seen0_sh);
        __syncthreads();
        if (abort_sh) return false;
        input_load(tt, a0, a1);
        return true;
    };
    f32x4 cur0 = {0.f, 0.f, 0.f, 0.f}, cur1 = {0.f, 0.f, 0.f, 0.f};
    if (layer && !input_late(0, cur0, cur1)) aborted = true;

    for (int t = 0; t < a.T && !aborted; ++t) {
        float gpre[2][4];
        if (layer == 0) {
            const float* gp = g + ((size_t)t * a.B + bclamp) * a.ldg + (2 * sl) * 64 + col;
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gpre[ub][gg] = gp[ub * 64 + 16 * gg];
        } else {
#pragma unroll
            for (int ub = 0; ub < 2; ++ub)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gpre[ub][gg] = gb[ub][gg];
        }
        f32x4 acc[8];
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t8][r] = 0.f;
        auto input_mfma = [&]() {                 // layer 1: acc += W_ih h0[t]
            float av[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = cur0[j];
                av[4 + j] = cur1[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int t8 = 0; t8 < 8; ++t8)
                    acc[t8] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], ireg[t8][j], acc[t8], 0, 0, 0);
        };
        if (layer && t == 0) input_mfma();
        if (t > 0) {
            if (tid == 0)       // (layer 1: the view of layer 0's progress is refreshed while waiting for the siblings, if it waits)
                wait_for(mine + (size_t)rep * 64, (unsigned)t * (unsigned)NSL, abortf, &abort_sh, nullptr,
                         layer ? cnt0 + (size_t)rep * 64 : nullptr, &seen0_sh);
            __syncthreads();                 // the polling wave joins after its match; every load below is sc1
            if (abort_sh) { aborted = true; break; }
            const unsigned par_r = hx_layer + (unsigned)((t - 1) & 1) * hx_par;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + col) * (unsigned)H + 32 * wave + 8 * rq) * 4u;
            const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off, par_r, 16));
            const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hxr, off + 16u, par_r, 16));
            if (layer) input_mfma();         // under the latency of the two loads above
            float av[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = a0[j];
                av[4 + j] = a1[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int t8 = 0; t8 < 8; ++t8)
                    acc[t8] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], breg[t8][j], acc[t8], 0, 0, 0);
        }
        // layer 1: the next step's input rows, if layer 0 is known to have published them (it usually is)
        const bool early = layer && t + 1 < a.T && input_known(t + 1);
        f32x4 nx0 = {0.f, 0.f, 0.f, 0.f}, nx1 = {0.f, 0.f, 0.f, 0.f};
        if (early) input_load(t + 1, nx0, nx1);
        // ---- reduce the 4 K-partials through LDS: [wave][tile][r][lane], conflict-free dword writes and reads
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * 8 + t8) * 4 + r) << 6) + lane] = acc[t8][r];
        __syncthreads();
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            float gate[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                float v = gpre[ub][gg];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[(((w * 8 + ub * 4 + gg) * 4 + wave) << 6) + lane];
                gate[gg] = v;
            }
            const float ig = sigmoidf_(gate[0]), fg = sigmoidf_(gate[1]);
            const float gv = tanhf_(gate[2]), og = sigmoidf_(gate[3]);
            const float cn = fg * creg[ub] + ig * gv;
            creg[ub] = cn;
            stage[myrow][ub * 16 + col] = og * tanhf_(cn);
            if (gsv && rowok) {
                float* gp = gsv + ((size_t)t * a.B + brow) * gld + (2 * sl + ub) * 64 + col;
                gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                csv[(size_t)run * TBH + ((size_t)t * a.B + brow) * H + sl * UPW + ub * 16 + col] = cn;
            }
        }
        __syncthreads();
        if (tid < 128) {
            // 16 rows x 8 float4: write-through (sc1) to the exchange buffer; the sequence output plain (layer 1) or, layer 0,
            // write-through as well: the layer-1 workgroups of this (run, tile) read it
            const int row = tid >> 3, c4 = tid & 7;
            const v4i pk = *(const v4i*)&stage[row][c4 * 4];
            const unsigned par_w = hx_layer + (unsigned)(t & 1) * hx_par;
            const unsigned off = (((unsigned)run * a.Bpad + b0 + row) * (unsigned)H + sl * UPW + c4 * 4) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
            if (b0 + row < a.B) {
                const size_t e = (size_t)run * TBH + ((size_t)t * a.B + b0 + row) * H + sl * UPW + c4 * 4;
                if (layer) *(v4i*)&a.hout[e] = pk;
                else __builtin_amdgcn_raw_buffer_store_b128(pk, h0r, (unsigned)(e * 4u), 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < a.nrep) __hip_atomic_fetch_add(mine + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // layer 1: the next step's input rows, the late way, if they were not requested above
        if (layer && t + 1 < a.T) {
            if (!early && !input_late(t + 1, nx0, nx1)) aborted = true;
            cur0 = nx0; cur1 = nx1;
        }
    }
    if (aborted) {
        if (tid == 0) idv_coop_raise(a.status);
        if (layer) {
            const float qnan = __builtin_nanf("");
            for (long long e = tid; e < (long long)a.T * 16 * UPW; e += 256) {
                const int u = (int)(e % UPW), br = (int)((e / UPW) & 15);
                const long long t = e / (UPW * 16);
                if (b0 + br < a.B) a.hout[(size_t)run * TBH + ((size_t)t * a.B + b0 + br) * H + sl * UPW + u] = qnan;
            }
        }
    }
}

}  // namespace idv_stack2


extern "C" int idv_lstm_stack2_f32_supported(int H, int B) {
    static const bool on = [] { const char* e = getenv("IDV_LSTM_STACK2"); return !e || e[0] != '0'; }();
    if (!on || H != 128 || B <= 0) return 0;
    const int tiles = (B + 15) / 16;
    return 4 * tiles <= idv_stack2::MAX_GROUPS && 2 * 4 * 4 * tiles <= idv_coop_max_workgroups();      // all resident, one per CU
}

extern "C" long long idv_lstm_stack2_f32_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return idv_stack2::SYNC_BYTES + 2LL * 2 * 4 * Bpad * H * 4;
}

// Both layers of the H = 128 complex LSTM in one cooperative launch.  g: layer-0 gate pre-activations as for
// idv_lstm_rec_coop_f32; whh0 / wih1_hh / whh1: idv_pack_lstm_hh fragments (W_ih of layer 1 is [4H][H] too); bias1: [2][4H] in
// gate-column order (idv_pack_lstm_ih of layer 1); h0, hout: [4 runs][T*B][H]; work: idv_lstm_stack2_f32_work_bytes bytes.
// Training forward: gsave1 ([run][T*B][4H]), csave0, csave1 ([4 runs][T*B][H]) all non-NULL -- the activated gates of layer 0
// replace its pre-activations in g, those of layer 1 go to gsave1 (the layouts idv_lstm_bptt reads); evaluation: all NULL.
extern "C" int idv_lstm_stack2_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh0, const float* wih1_hh,
                                   const float* whh1, const float* bias1, float* h0, float* hout, int H, int B, int T, void* work,
                                   float* gsave1, float* csave0, float* csave1, void* stream) {
    using namespace idv_stack2;
    if (!g || !whh0 || !wih1_hh || !whh1 || !bias1 || !h0 || !hout || !work || T <= 0 || !idv_lstm_stack2_f32_supported(H, B))
        return IDV_EINVAL;
    if ((gsave1 != nullptr) != (csave0 != nullptr) || (csave0 != nullptr) != (csave1 != nullptr)) return IDV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(work) & 15) || (reinterpret_cast<uintptr_t>(h0) & 15) || (reinterpret_cast<uintptr_t>(hout) & 15))
        return IDV_EINVAL;
    const long long h0_bytes = 4LL * T * B * H * 4;
    if (h0_bytes >= 0xfffffe00LL) return IDV_EINVAL;          // 32-bit buffer offsets
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (B + 15) / 16;
    const long long Bpad = 16LL * tiles;
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    Args a{};
    a.g = g; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.whh0 = whh0; a.wih1 = wih1_hh; a.whh1 = whh1; a.bias1 = bias1;
    a.h0 = h0; a.h0_bytes = (unsigned)h0_bytes; a.hout = hout;
    a.gsave1 = gsave1; a.csave0 = csave0; a.csave1 = csave1;
    a.sync = (unsigned*)work;
    a.hx = (float*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)(2LL * 2 * 4 * Bpad * H * 4);
    a.nrep = 4;
    a.B = B; a.T = T; a.Bpad = (int)Bpad; a.tiles = tiles;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    a.status = idv_coop_status_word();
    const size_t smem = 84 * 1024;                   // > half a CU's LDS: one workgroup per CU (red[] needs 32 KB)
    if (hipFuncSetAttribute((const void*)lstm_stack2_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(lstm_stack2_f32_kernel, dim3(2 * NSL, 4, tiles), dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
