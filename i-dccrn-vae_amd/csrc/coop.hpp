// Shared host-side state of the cooperative (spin-synchronised) recurrences: lstm_pers.hip, lstm_pers_f32.hip,
// lstm_coop_f32.hip, lstm_stack2_f32.hip, lstm_bptt_coop_f32.hip, lstm_bptt_stack2_f32.hip.  Implemented in lstm_pers.hip.
//
// These kernels need ALL sibling workgroups resident at once (each spins on the others), so the library keeps three pieces
// of per-device process state for them (and nothing else in the library is stateful):
//   * the residency bound: workgroups a cooperative launch may use, from hipDeviceProp_t::multiProcessorCount (one workgroup
//     per CU: each launch asks for > half a CU's LDS) minus 1/16 of the CUs as head room;
//   * the launch chain: cooperative launches of different streams of one device are ordered through an event, because two
//     half-resident launches would wait for each other until the spin bound poisons both;
//   * a sticky status word in host-mapped memory: a kernel whose bounded spin ran out (outputs poisoned with NaN) also
//     stores 1 there (system scope); from then on idv_coop_last_status() and EVERY cooperative entry of the device report
//     IDV_ECOOP until idv_coop_last_status(1) acknowledges it (a sticky device error: no caller can consume it by accident).
#pragma once
#include <hip/hip_runtime.h>

#define IDV_ECOOP (-3)

// workgroups a cooperative launch may use on the current device (no device visible: the MI355X figure, 240)
extern "C" int idv_coop_max_workgroups(void);
// take the chain lock and make `st` wait for the previous cooperative launch of this device.  Returns IDV_ECOOP (lock NOT held,
// status NOT cleared) while an earlier cooperative launch's time-out has not been acknowledged with idv_coop_last_status(1).
int idv_coop_chain_begin(hipStream_t st);
// record the launch + release the lock; must follow every successful idv_coop_chain_begin, on every path
int idv_coop_chain_end(hipStream_t st);
// device pointer of the current device's sticky status word (nullptr if it cannot be allocated: the kernels then only poison)
unsigned* idv_coop_status_word();

// device side: called by one thread of a workgroup that aborted
__device__ __forceinline__ void idv_coop_raise(unsigned* host_word) {
    if (host_word) __hip_atomic_store(host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
