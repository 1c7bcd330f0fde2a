// waitflag.hpp -- completion signalling without hipStreamSynchronize.
//
// Measured on this box (profiles/calib/sync_latency.hip): launch + hipStreamSynchronize 12.4 us, launch + spinning on an int the
// kernel stores to pinned host memory 6.5 us.  The last kernel of a phase publishes a sequence number after its results
// (__threadfence_system() first; results live in the same fine-grained pinned allocation class), the host spins on it.
// Multi-workgroup kernels use a device counter: every workgroup fences and increments it, the one that sees it complete publishes.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>

namespace sdvgn {

// call by ONE thread of a workgroup after the workgroup's results are written and a __syncthreads(); nblocks = workgroups that
// will call this for the same (counter, seq).
// Ordering: __syncthreads() is `fence release(workgroup); s_barrier; fence acquire(workgroup)`, so every result store of the
// workgroup happens-before the publishing thread's system-scope release fence below, and release fences are cumulative in the
// AMDGPU memory model: the flag store cannot become visible to the host before those stores.  (A system-scope fence in EVERY storing
// thread was tried for belt and braces: each one is an L2 write-back, +10 us on k_ef_acc_reduce -- profiles/r02_notes.txt.)
// The host side pairs it with an acquire fence after the spin (wait_flag).  The counter must be 0 before the launch; the publisher resets it.
__device__ __forceinline__ void publish_when_all_done(unsigned* counter, unsigned nblocks, volatile int* flag, int seq) {
    __threadfence_system();
    const unsigned prev = atomicAdd(counter, 1u);
    if (prev == nblocks - 1) {
        *counter = 0;
        __threadfence_system();
        *flag = seq;
    }
}

// host: spin until *flag == seq; falls back to a stream synchronisation after ~2 s (a failed launch never publishes)
static inline hipError_t wait_flag(volatile int* flag, int seq, hipStream_t stream) {
    unsigned spins = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (*flag != seq) {
        if ((++spins & 0xffffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            return *flag == seq ? hipSuccess : hipErrorUnknown;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);   // the result reads that follow must not be satisfied before the flag read
    return hipSuccess;
}

}  // namespace sdvgn
