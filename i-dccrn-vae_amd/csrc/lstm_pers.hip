// Persistent cooperative recurrence for the hidden sizes of the VAE encoders (H = 384 = 3*zdim, H = 768 = 6*zdim;
// reference model/pvae_module.py:1819, :2160-2163 feeding ComplexLSTM.forward, model/complex_progress.py:50-74).
//
// W_hh of one weight set (2.4 / 9.4 MB) does not fit one CU, so the 4H gate columns are spread over H/16 workgroups per
// weight set, each holding the (i, f, g, o) columns of 16 hidden units for ALL T steps in registers as split-bf16 MFMA
// fragments (96 / 192 VGPRs per lane) -- one launch per layer instead of one launch per time step.  Per step a workgroup
//   1. waits until every workgroup of its group (same weight set, same batch chunk) has published h_{t-1}
//      (monotonic arrive counter in global memory; the 8 XCDs have private L2s and a CU's L1 is never refreshed, so the
//      exchange buffer is written with write-through `sc1` 16-byte stores, drained with s_waitcnt vmcnt(0) by every storing
//      wave, signalled by ONE agent-scope atomic add per workgroup behind the workgroup barrier, polled with an `sc1` load
//      and read ONLY with `sc1` buffer loads after the poller has joined the workgroup barrier -- the fence-free hand-off
//      form; no buffer_wbl2 / buffer_inv per step, which cost 2.9 us of the 6.9 us step in the first version),
//   2. reads h_{t-1} (all H units of its 2 runs x 16*RTR sequences, split-bf16, straight into MFMA A fragments; the 4
//      waves split K), contracts with its W_hh slice on v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate),
//   3. reduces the 4 K-partials through LDS, adds the hoisted input projection, runs the cell update with c kept in
//      registers, and publishes its 16 units of h_t (fp32 for the next layer, split-bf16 for the next step).
// Every spin is bounded: a workgroup that waits longer than ~0.4 s raises the abort flag, all workgroups drain, and the
// outputs are poisoned with NaN (no hang; results never silently wrong).
#include <cstdlib>
#include <mutex>
#include "common.hpp"
#include "coop.hpp"
#include "../../include/idccrn_hip.h"

namespace idv_pers {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct PersArgs {
    const float* g;           // gate pre-activations (hoisted input projection)
    long long g_run_z, g_run_s;
    int ldg;
    const uint4* whh16;       // split-bf16 fragments of idv_pack_lstm_hh: [set][tile = ub*4 + gate][kb][hi|lo][lane]
    float* hout;              // [4 runs][T*B][H]  (nullptr: not wanted)
    float* gsave;             // training: activated gates (i, f, g, o) written back over the pre-activations (= g), or nullptr
    float* csave;             // training: cell state per step [4 runs][T*B][H], or nullptr
    uint4* kimg;              // or nullptr: K-major split image of h_t, slot ((run * H/8 + octet) * Jp + b * Tp + t + 1)
    long long kimg_lo;        //   hi -> lo distance in 16-byte slots
    int Tp, Jp;
    unsigned short* hx;       // exchange [2 parity][4 runs][Bpad][H/8][hi 8 x bf16 | lo 8 x bf16]
    unsigned hx_bytes;
    unsigned* sync;           // [abort flag: 256 B][group = set * chunks + chunk][replica][256 B] arrive counters
    int nrep;                 // replicas of each arrive counter (1, 2, 4 or 8), each on a 256-byte block of its own
    int H, B, T, Bpad, nchunks;
    int fault;                // test hook (IDV_COOP_FAULT=1): workgroup (0, 0, 0) never arrives -> the bounded spins must abort
    unsigned* status;         // host-mapped sticky status word (coop.hpp) or nullptr
    unsigned long long* prof; // diagnostic build only: [workgroup][8] accumulated phase cycles, [7] = XCC id
};

constexpr unsigned long long SPIN_LIMIT_CYCLES = 1000000000ull;     // ~0.4 s at 2.4 GHz

#define IDV_STAMP(i)                                                       \
    if (PROF && pw) {                                                      \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        pacc[i] += now_ - plast;                                           \
        plast = now_;                                                      \
    }

template <int KBW, int NRT, bool PROF = false>     // k-blocks (32) per wave = H/128; 16-row tiles per workgroup
__global__ __launch_bounds__(256, 1) void lstm_pers_kernel(const PersArgs a) {
    // the 2 runs of a weight set (real / imaginary input) x ceil(B/16) row tiles form one tile space of NT = 2 * TPR tiles;
    // a workgroup owns NRT consecutive tiles of it (its batch chunk)
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][NRT][4 gates][4 rows r][64 lanes]
    __shared__ int abort_sh;
    __shared__ __attribute__((aligned(16))) unsigned short stage[NRT][2][16][16];   // [tile][hi|lo][row][unit]
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc((void*)a.hx, 0, a.hx_bytes, 0x00020000);
    const int H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = blockIdx.x, s = blockIdx.y, ch = blockIdx.z;
    const int nslice = gridDim.x;
    const int col = lane & 15, rq = lane >> 4;
    const int KB = H / 32;
    const int TPR = a.Bpad / 16, NT = 2 * TPR;
    // every arriving workgroup adds to ALL replicas of its group's counter (one wave instruction, one lane per replica);
    // a workgroup polls ONE replica: 1/nrep of the pollers per word, and no two groups share a memory channel (with the four
    // group counters in one 16-byte block the poll round trip grew by 28 ns per polling workgroup of the LAUNCH: 2.8 us of
    // a 6.2 us step at 96 workgroups)
    unsigned* abortf = a.sync;
    unsigned* counter0 = a.sync + 64 + (size_t)((s * a.nchunks + ch) * a.nrep) * 64;
    unsigned* counter = counter0 + (size_t)(sl & (a.nrep - 1)) * 64;
    const size_t TBH = (size_t)a.T * a.B * H;

    // this workgroup's tiles: run (2 z + s), first row, validity (the last chunk of an odd tile count is ragged)
    int t_run[NRT], t_b0[NRT];
    bool t_ok[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        int tile = ch * NRT + rt;
        t_ok[rt] = tile < NT;
        if (tile >= NT) tile = NT - 1;
        const int z = tile / TPR;
        t_run[rt] = 2 * z + s;
        t_b0[rt] = (tile - z * TPR) * 16;
    }

    // W_hh slice: gate tiles (sl*4 + g), this wave's k-blocks
    uint4 wreg[4][KBW][2];
    {
        const uint4* wb = a.whh16 + (size_t)s * (H / 4) * KB * 2 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < KBW; ++k)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    wreg[g][k][sp] = wb[((size_t)((sl * 4 + g) * KB + wave * KBW + k) * 2 + sp) * 64];
    }
    // the cell update is split by ROW over the waves: wave w owns rows rq * 4 + w of every tile (unit = lane & 15), so the
    // transcendental work (5 exp2 + 5 rcp per element, quarter rate) is spread over all four SIMDs whatever NRT is
    float creg[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) creg[rt] = 0.f;
    const int myrow = rq * 4 + wave;

    bool aborted = false;
    if (tid == 0) abort_sh = 0;
    if (a.fault && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) return;      // injected failure (tests only)
    const bool pw = PROF && wave == 0;
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = 0;
    if (PROF && pw) plast = __builtin_readcyclecounter();
    for (int t = 0; t < a.T; ++t) {
        // ---- inputs of the cell update (independent of h): issue first
        float gpre[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int z = t_run[rt] >> 1;
            const float* g = a.g + z * a.g_run_z + s * a.g_run_s;
            int b = t_b0[rt] + myrow;
            if (b >= a.B) b = a.B - 1;
            const float* gp = g + ((size_t)t * a.B + b) * a.ldg + sl * 64 + col;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gpre[rt][gg] = gp[16 * gg];
        }
        f32x4 acc[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt][g][r] = 0.f;

        if (t > 0) {
            // ---- wait for h_{t-1} of the whole group
            if (tid == 0) {
                const unsigned want = (unsigned)t * (unsigned)nslice;
                const unsigned long long t0 = wall_clock64();
                unsigned long long spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0) {
                        if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abort_sh = 1; break; }
                        if (wall_clock64() - t0 > SPIN_LIMIT_CYCLES / 24) {     // wall_clock64 ticks at 100 MHz
                            __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_sh = 1;
                            break;
                        }
                    }
                }
                IDV_STAMP(0)                                 // gpre issue + spin
            }
            // no acquire fence: every byte of the exchange buffer was stored sc1 and drained before the arrive, and every
            // load of it below is an sc1 buffer load issued after this barrier, which the polling wave joins after its match
            __syncthreads();
            if (abort_sh) { aborted = true; break; }
            IDV_STAMP(2)
            // ---- gates += h_{t-1} W_hh^T : A fragments straight from the exchange buffer (row = lane & 15, 8 k per lane)
            const unsigned par_r = (unsigned)((t - 1) & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;
            // every row tile's fragments are in flight at once where the registers allow (<= 96 VGPRs), else the fragments of
            // row tile rt+1 are in flight while row tile rt multiplies (left to itself hipcc issues each (hi, lo) pair right
            // before its 12 MFMAs, i.e. 4 * KBW serial memory round trips)
            constexpr bool ALL_UP = (NRT * KBW <= 12);
            constexpr int NB = ALL_UP ? NRT : 2;
            uint4 ah[NB][KBW], al[NB][KBW];
            auto load_a = [&](int rt, uint4 (&h_)[KBW], uint4 (&l_)[KBW]) {
                const unsigned rowoff = ((unsigned)t_run[rt] * a.Bpad + t_b0[rt] + col) * (unsigned)(H / 8);
#pragma unroll
                for (int k = 0; k < KBW; ++k) {
                    const unsigned ko = (rowoff + 4 * (wave * KBW + k) + rq) * 32u;
                    h_[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(hxr, ko, par_r, 16));
                    l_[k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(hxr, ko + 16u, par_r, 16));
                }
            };
            if (ALL_UP) {
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) load_a(rt, ah[rt % NB], al[rt % NB]);
            } else {
                load_a(0, ah[0], al[0]);
            }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                if (!ALL_UP && rt + 1 < NRT) load_a(rt + 1, ah[(rt + 1) % NB], al[(rt + 1) % NB]);
                __builtin_amdgcn_sched_barrier(0);          // the prefetch is issued BEFORE this row tile's MFMAs
#pragma unroll
                for (int k = 0; k < KBW; ++k) {
                    const bf16x8_t vh = __builtin_bit_cast(bf16x8_t, ah[rt % NB][k]);
                    const bf16x8_t vl = __builtin_bit_cast(bf16x8_t, al[rt % NB][k]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, wreg[g][k][0]);
                        const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, wreg[g][k][1]);
                        f32x4 c = acc[rt][g];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, bh, c, 0, 0, 0);
                        acc[rt][g] = c;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            IDV_STAMP(3)                                     // A loads + MFMA
        }
        // ---- reduce the 4 K-partials through LDS; layout [wave][tile][gate][r][lane]: conflict-free dword writes and reads
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((((wave * NRT + rt) * 4 + g) * 4 + r) << 6) + lane] = acc[rt][g][r];
        __syncthreads();
        const unsigned par_w = (unsigned)(t & 1) * 4u * (unsigned)a.Bpad * (unsigned)H * 4u;     // byte offset of parity t & 1
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = gpre[rt][g];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[((((w * NRT + rt) * 4 + g) * 4 + wave) << 6) + lane];
                gate[g] = v;
            }
            const float ig = sigmoidf_(gate[0]), fg = sigmoidf_(gate[1]);
            const float gv = tanhf_(gate[2]), og = sigmoidf_(gate[3]);
            const float cn = fg * creg[rt] + ig * gv;
            creg[rt] = cn;
            const float hv = og * tanhf_(cn);
            if (a.gsave) {                           // what idv_lstm_bptt reads: the same layout the per-step kernels leave
                const int z = t_run[rt] >> 1;
                int b = t_b0[rt] + myrow;
                if (t_ok[rt] && b < a.B) {
                    float* gp = a.gsave + z * a.g_run_z + s * a.g_run_s + ((size_t)t * a.B + b) * a.ldg + sl * 64 + col;
                    gp[0] = ig; gp[16] = fg; gp[32] = gv; gp[48] = og;
                    a.csave[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + col] = cn;
                }
            }
            const __bf16 hh = (__bf16)hv;
            const __bf16 hl = (__bf16)(hv - (float)hh);
            // transpose through LDS: a lane holds one unit of one row, a 16-byte store wants 8 units of a row
            stage[rt][0][myrow][col] = __builtin_bit_cast(unsigned short, hh);
            stage[rt][1][myrow][col] = __builtin_bit_cast(unsigned short, hl);
            const int b = t_b0[rt] + myrow;
            if (a.hout && t_ok[rt] && b < a.B) a.hout[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + col] = hv;
        }
        __syncthreads();
        if (wave < NRT && t_ok[wave < NRT ? wave : 0]) {
            // lanes 0..31: hi halves, 32..63: lo halves; (row, 8-unit chunk) = ((lane & 31) >> 1, lane & 1); 16-byte
            // write-through (sc1) stores: the exchange buffer never sits dirty in this XCD's L2
            const int rt = wave;
            const int sp = lane >> 5, row = (lane & 31) >> 1, c8 = lane & 1;
            const v4i pk = *(const v4i*)&stage[rt][sp][row][c8 * 8];
            const unsigned off = (((unsigned)t_run[rt] * a.Bpad + t_b0[rt] + row) * (unsigned)(H / 8) + sl * 2 + c8) * 32u + sp * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(pk, hxr, off, par_w, 16);       // aux 16 = sc1
            // the same 16 bytes are a slot of the K-major split image the next layer's input projection reads (plain store)
            const int b = t_b0[rt] + row;
            if (a.kimg && b < a.B)
                a.kimg[(size_t)sp * a.kimg_lo + ((size_t)t_run[rt] * (H / 8) + sl * 2 + c8) * a.Jp + (size_t)b * a.Tp + t + 1] =
                    __builtin_bit_cast(uint4, pk);
        }
        // ---- publish: every storing wave drains its stores, the workgroup meets, ONE wave instruction arrives
        IDV_STAMP(4)                                         // LDS reduction + cell update + stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        IDV_STAMP(5)                                         // store drain + barrier
        if (tid < a.nrep) __hip_atomic_fetch_add(counter0 + (size_t)tid * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PROF && pw && lane == 0) {
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = 0; i < 7; ++i) a.prof[wg * 8 + i] = pacc[i];
        a.prof[wg * 8 + 7] = __builtin_amdgcn_s_getreg(6164) & 15;      // HW_REG_XCC_ID (id 20, offset 0, size 4)
    }
    if (aborted) {
        // poison this workgroup's outputs: a timed-out recurrence must never look like a result; the host learns it through
        // the sticky status word (the next cooperative entry / idv_coop_last_status returns IDV_ECOOP)
        if (tid == 0) idv_coop_raise(a.status);
        const float qnan = __builtin_nanf("");
        for (int rt = 0; rt < NRT; ++rt) {
            if (!t_ok[rt]) continue;
            for (long long e = tid; e < (long long)a.T * 16 * 16; e += 256) {
                const int u = (int)(e & 15), br = (int)((e >> 4) & 15);
                const long long t = e >> 8;
                const int b = t_b0[rt] + br;
                if (b < a.B && a.hout) a.hout[(size_t)t_run[rt] * TBH + ((size_t)t * a.B + b) * H + sl * 16 + u] = qnan;
                if (b < a.B && a.kimg && u < 2)      // hi halves of both octets: NaN bit patterns
                    a.kimg[((size_t)t_run[rt] * (H / 8) + sl * 2 + u) * a.Jp + (size_t)b * a.Tp + t + 1] =
                        make_uint4(0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u);
            }
        }
    }
}

// 16-row tiles per workgroup: as few as the residency bound allows (every workgroup must be resident at once, one per CU:
// idv_coop_max_workgroups(), 240 on a 256-CU MI355X).  The step is bound by each CU's read of h_{t-1} (rows x H x 4 bytes of freshly handed-off
// data at ~20 bytes / cycle / CU) and by fixed hand-off latencies, so fewer rows per workgroup on more CUs is faster:
// H = 384, B = 32: 1 tile (192 workgroups); H = 768, B = 32: 2 (192); H = 768, B = 64: 4 (192).
inline int nrt_for(int H, int B) {
    static const int forced = [] { const char* e = getenv("IDV_PERS_NRT"); return e ? atoi(e) : 0; }();
    const int NT = 2 * ((B + 15) / 16);
    for (int nrt = 1; nrt <= 4; nrt *= 2) {
        if (forced > nrt) continue;
        if (2 * (H / 16) * ((NT + nrt - 1) / nrt) <= idv_coop_max_workgroups()) return nrt;
    }
    return 0;
}

}  // namespace idv_pers

extern "C" int idv_lstm_pers_supported(int H, int B) {
    if (H != 384 && H != 768) return 0;
    if (B <= 0) return 0;
    return idv_pers::nrt_for(H, B) > 0;
}

static constexpr int SYNC_BYTES = 256 + 20 * 8 * 256;     // abort flag + (<= 20 groups) x (<= 8 replicas) x 256 B

extern "C" long long idv_lstm_pers_work_bytes(int H, int B) {
    const long long Bpad = (B + 15) / 16 * 16;
    return SYNC_BYTES + 2LL * 4 * Bpad * H * 4;       // [abort flag + arrive counters, zeroed per call][exchange]
}

// ---- per-device process state of the cooperative kernels (coop.hpp) ------------------------------------------------------
// Two cooperative launches must never share the chip: each needs ALL its workgroups resident (it spins on its siblings), and
// two half-resident launches on different streams would wait for each other until the spin bound poisons both.  Launches
// from different streams of one device are therefore chained through an event (other kernels may still overlap them).
static std::mutex g_pers_mu;
static hipEvent_t g_pers_done[16] = {};
static hipStream_t g_pers_stream[16] = {};
static int g_max_wg[16] = {};                 // 0: not queried yet
static unsigned* g_status_host[16] = {};      // host-mapped sticky status words
static unsigned* g_status_dev[16] = {};
static bool g_status_tried[16] = {};

static int cur_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    return dev;
}

extern "C" int idv_coop_max_workgroups(void) {
    // no device visible (sizing queries in the build container): the MI355X figure, 256 CUs - 16
    const int dev = cur_dev();
    if (dev < 0) { (void)hipGetLastError(); return 240; }
    std::lock_guard<std::mutex> lk(g_pers_mu);
    if (g_max_wg[dev] == 0) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) {
            (void)hipGetLastError();
            g_max_wg[dev] = 240;
        } else {
            // one workgroup per CU (every cooperative launch requests > half a CU's LDS); 1/16 of the CUs stay free so that a
            // kernel of another stream still finds a CU and never delays the start of a sibling workgroup indefinitely
            const int cu = pr.multiProcessorCount;
            int n = cu - cu / 16;
            const char* e = getenv("IDV_COOP_MAX_WG");              // experiments / CU-masked devices
            if (e && atoi(e) > 0 && atoi(e) < n) n = atoi(e);
            g_max_wg[dev] = n;
        }
    }
    return g_max_wg[dev];
}

unsigned* idv_coop_status_word() {
    const int dev = cur_dev();
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_pers_mu);
    if (!g_status_tried[dev]) {
        g_status_tried[dev] = true;
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 256, hipHostMallocMapped) == hipSuccess) {
            *(volatile unsigned*)h = 0u;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                g_status_host[dev] = (unsigned*)h;
                g_status_dev[dev] = (unsigned*)d;
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return g_status_dev[dev];
}

// IDV_ECOOP if a cooperative kernel of the current device has timed out since the status was last cleared (its outputs are
// NaN-poisoned); clear != 0 acknowledges and resets it -- until then every cooperative entry of the device refuses with IDV_ECOOP.  The word is written by the device when the kernel aborts, so a launch that is
// still queued is not covered: synchronise the stream first for a definite answer.
extern "C" int idv_coop_last_status(int clear) {
    const int dev = cur_dev();
    if (dev < 0) return IDV_ELAUNCH;
    (void)idv_coop_status_word();
    std::lock_guard<std::mutex> lk(g_pers_mu);
    volatile unsigned* w = g_status_host[dev];
    if (!w || !*w) return IDV_OK;
    if (clear) *w = 0u;
    return IDV_ECOOP;
}

int idv_coop_chain_begin(hipStream_t st) {
    const int dev = cur_dev();
    if (dev < 0) return IDV_ELAUNCH;
    g_pers_mu.lock();
    volatile unsigned* w = g_status_host[dev];
    if (w && *w) {                       // an earlier cooperative launch timed out and nobody has acknowledged it: refuse, like a
        g_pers_mu.unlock();              // sticky device error, until idv_coop_last_status(1) -- the status is NOT consumed here,
        return IDV_ECOOP;                // so an unrelated caller cannot swallow it
    }
    if (g_pers_done[dev] && g_pers_stream[dev] != st && hipStreamWaitEvent(st, g_pers_done[dev], 0) != hipSuccess) {
        g_pers_mu.unlock();
        return IDV_ELAUNCH;
    }
    return IDV_OK;
}
int idv_coop_chain_end(hipStream_t st) {
    const int dev = cur_dev();
    int rc = dev < 0 ? IDV_ELAUNCH : IDV_OK;
    if (!rc && !g_pers_done[dev] && hipEventCreateWithFlags(&g_pers_done[dev], hipEventDisableTiming) != hipSuccess) rc = IDV_ELAUNCH;
    if (!rc && hipEventRecord(g_pers_done[dev], st) != hipSuccess) rc = IDV_ELAUNCH;
    if (!rc) g_pers_stream[dev] = st;
    g_pers_mu.unlock();
    return rc;
}

static unsigned long long* g_prof = nullptr;

// diagnostic: while a device buffer of 256 x 8 counters is registered, idv_lstm_rec_pers launches the instrumented twin of
// the kernel, in which wave 0 of every workgroup (index (chunk * 2 + set) * H/16 + slice) accumulates the core-clock cycles
// it spends per phase over the T steps: [0] gate-input issue + spin on the arrive counter, [2] barrier, [3] h loads + MFMA,
// [4] LDS reduction + cell update + store issue, [5] store drain + barrier; [7] = the XCC the workgroup ran on.
// nullptr restores the production kernel.
extern "C" void idv_lstm_pers_set_profile(unsigned long long* prof_cycles) { g_prof = prof_cycles; }

// one layer of the recurrence; work: idv_lstm_pers_work_bytes(H, B) bytes (16-byte aligned), contents arbitrary.
// gsave (== g) / csave: training forward, the activated gates overwrite the pre-activations and the cell states are kept
// for idv_lstm_bptt (layouts as idv_clstm_fwd flags bit 2).  Outputs, at least one: hout [4 runs][T*B][H] fp32 and / or kimg, the K-major split image of h (4 runs x H/8 octets x Jp
// columns, column b*Tp + t + 1; kimg_lo_slots 16-byte slots from the hi to the lo plane) that idv_lstm_proj1_bf16x3 reads.
extern "C" int idv_lstm_rec_pers(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout,
                                 int H, int B, int T, void* work, void* kimg, long long kimg_lo_slots, int Tp, int Jp,
                                 float* gsave, float* csave, void* stream) {
    using namespace idv_pers;
    unsigned long long* prof = g_prof;
    if (!g || !whh_frag || (!hout && !kimg) || !work || T <= 0 || !idv_lstm_pers_supported(H, B)) return IDV_EINVAL;
    if ((gsave != nullptr) != (csave != nullptr) || (gsave && gsave != g)) return IDV_EINVAL;   // gates are saved in place
    if (reinterpret_cast<uintptr_t>(work) & 15) return IDV_EINVAL;
    if (kimg && ((reinterpret_cast<uintptr_t>(kimg) & 15) || Tp < T + 1 || Jp < B * Tp || kimg_lo_slots < 4LL * (H / 8) * Jp))
        return IDV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nrt = nrt_for(H, B);
    const int TPR = (B + 15) / 16, NT = 2 * TPR;
    const int chunks = (NT + nrt - 1) / nrt;
    const long long Bpad = (long long)TPR * 16;
    const size_t hx_bytes = (size_t)2 * 4 * Bpad * H * 4;
    if (2 * chunks > 20) return IDV_EINVAL;
    // only the polled words are zeroed: every row of the exchange buffer a step reads was written by the step before
    if (hipMemsetAsync(work, 0, SYNC_BYTES, st) != hipSuccess) return IDV_ELAUNCH;
    static const int nrep = [] {
        const char* e = getenv("IDV_PERS_REPL");
        const int v = e ? atoi(e) : 8;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 8;
    }();
    PersArgs a{};
    a.g = g; a.g_run_z = g_run_z; a.g_run_s = g_run_s; a.ldg = ldg;
    a.whh16 = (const uint4*)(whh_frag + (size_t)2 * 4 * H * H);
    a.hout = hout;
    a.kimg = (uint4*)kimg; a.kimg_lo = kimg_lo_slots; a.Tp = Tp; a.Jp = Jp;
    a.gsave = gsave; a.csave = csave;
    a.sync = (unsigned*)work;
    a.hx = (unsigned short*)((char*)work + SYNC_BYTES);
    a.hx_bytes = (unsigned)hx_bytes;
    a.nrep = nrep;
    a.H = H; a.B = B; a.T = T; a.Bpad = (int)Bpad; a.nchunks = chunks; a.prof = prof;
    { const char* e = getenv("IDV_COOP_FAULT"); a.fault = (e && e[0] == '1') ? 1 : 0; }
    dim3 grid(H / 16, 2, chunks);
    // at least 84 KB of LDS per workgroup: ONE workgroup per CU whatever the register count (the hand-off form above is the
    // one measured for one workgroup per CU, and co-located workgroups would share one CU's miss bandwidth)
    size_t smem = (size_t)4 * nrt * 4 * 4 * 64 * sizeof(float);
    if (smem < 84 * 1024) smem = 84 * 1024;
    a.status = idv_coop_status_word();
    // the kernel is chosen and its LDS attribute set BEFORE the chain lock is taken: no early return may hold the lock
    typedef void (*kern_t)(const PersArgs);
    kern_t k = nullptr;
#define IDV_PERS_PICK(KBW, NRT) k = prof ? (kern_t)lstm_pers_kernel<KBW, NRT, true> : (kern_t)lstm_pers_kernel<KBW, NRT, false>
    if (H == 384) {
        if (nrt == 1) IDV_PERS_PICK(3, 1); else if (nrt == 2) IDV_PERS_PICK(3, 2); else IDV_PERS_PICK(3, 4);
    } else {
        if (nrt == 1) IDV_PERS_PICK(6, 1); else if (nrt == 2) IDV_PERS_PICK(6, 2); else IDV_PERS_PICK(6, 4);
    }
#undef IDV_PERS_PICK
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return IDV_ELAUNCH;
    int rc = idv_coop_chain_begin(st);
    if (rc) return rc;
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a);
    if ((rc = idv_coop_chain_end(st))) return rc;
    return idv_launch_status();
}
