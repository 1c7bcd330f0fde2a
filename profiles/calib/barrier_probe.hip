// barrier_probe.hip -- round 6: what does a kernel boundary cost on the dependent chain of one stream, against a grid-wide barrier inside ONE launch
// (a counter in device memory, device-scope atomics) -- with and without the L2 write-back / invalidate a plain-store hand-over across XCDs needs?
//   hipcc --offload-arch=gfx950 -O3 barrier_probe.hip -o barrier_probe
// T1  N dependent launches of G workgroups, each writes `bytes` per workgroup and reads what workgroup (b + G/2 + 1) % G wrote in the launch before
// T2  ONE launch of G resident workgroups, N phases of the same write / read, separated by a grid barrier: __threadfence() (release) + counter +
//     __threadfence() (acquire); plain loads / stores of the payload
// T3  the same, the payload through device-scope relaxed atomics (write-through, L2-bypassing reads), the barrier without fences
// every variant verifies what it read (a stale read counts as an error).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned val(int b, int i, int phase) { return (unsigned)(b * 1315423911u) ^ (unsigned)(i * 2654435761u) ^ (unsigned)(phase * 97u + 1u); }

template <int MODE>   // 0 plain, 1 device-scope atomics
__device__ __forceinline__ void phase_body(unsigned* buf, int words, int G, int b, int phase, unsigned* errs) {
    // read what the partner wrote in phase - 1 (buffer (phase - 1) & 1), write this phase's into buffer phase & 1
    const int partner = (b + G / 2 + 1) % G;
    const unsigned* src = buf + ((size_t)((phase - 1) & 1) * G + partner) * words;
    unsigned* dst = buf + ((size_t)(phase & 1) * G + b) * words;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        if (phase > 0) {
            const unsigned v = MODE ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i];
            bad += v != val(partner, i, phase - 1);
        }
        if (MODE) __hip_atomic_store(dst + i, val(b, i, phase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else dst[i] = val(b, i, phase);
    }
    if (bad) atomicAdd(errs, bad);
}
template <int MODE>
__global__ void __launch_bounds__(256) k_phase(unsigned* buf, int words, int phase, unsigned* errs) { phase_body<MODE>(buf, words, gridDim.x, blockIdx.x, phase, errs); }

template <int MODE>
__global__ void __launch_bounds__(256) k_persistent(unsigned* buf, int words, int phases, unsigned* ctr, unsigned* errs, unsigned long long* clk) {
    const int G = gridDim.x, b = blockIdx.x;
    unsigned long long t0 = 0;
    for (int phase = 0; phase < phases; ++phase) {
        if (phase == 8 && b == 0 && threadIdx.x == 0) t0 = wall_clock64();
        phase_body<MODE>(buf, words, G, b, phase, errs);
        // grid barrier
        if (MODE == 0) __threadfence();              // every lane's stores written back (release)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(phase + 1) * (unsigned)G;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (MODE == 0) __threadfence();              // (acquire: invalidate)
    }
    if (b == 0 && threadIdx.x == 0) { clk[0] = wall_clock64() - t0; }
}

int main(int argc, char** argv) {
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 208;
    unsigned *buf, *ctr, *errs; unsigned long long* clk;
    const size_t maxwords = (size_t)2 * 2048 * 16384;
    CHK(hipMalloc(&buf, maxwords * 4)); CHK(hipMalloc(&ctr, 64)); CHK(hipMalloc(&errs, 64)); CHK(hipMalloc(&clk, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-8s %-10s | %-28s | %-34s | %-34s\n", "G", "B / wg", "T1 dependent launches us", "T2 one launch, fenced barrier us", "T3 one launch, atomics payload us");
    const int Gs[] = {64, 256, 512, 1024};
    const int Ws[] = {16, 1024, 16384};     // words per workgroup: 64 B, 4 kB, 64 kB
    for (int G : Gs) for (int W : Ws) {
        double r[3] = {0, 0, 0}; unsigned er[3] = {0, 0, 0};
        // T1
        CHK(hipMemsetAsync(errs, 0, 4, s));
        for (int p = 0; p < 8; ++p) k_phase<0><<<G, 256, 0, s>>>(buf, W, p, errs);
        CHK(hipEventRecord(e0, s));
        for (int p = 8; p < N; ++p) k_phase<0><<<G, 256, 0, s>>>(buf, W, p, errs);
        CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); r[0] = ms * 1e3 / (N - 8);
        CHK(hipMemcpy(&er[0], errs, 4, hipMemcpyDeviceToHost));
        // T2 / T3
        for (int mode = 0; mode < 2; ++mode) {
            CHK(hipMemsetAsync(errs, 0, 4, s)); CHK(hipMemsetAsync(ctr, 0, 4, s));
            if (mode == 0) k_persistent<0><<<G, 256, 0, s>>>(buf, W, N, ctr, errs, clk); else k_persistent<1><<<G, 256, 0, s>>>(buf, W, N, ctr, errs, clk);
            CHK(hipStreamSynchronize(s));
            unsigned long long c; CHK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
            r[1 + mode] = (double)c / 100.0 / (N - 8);           // wall_clock64: 100 MHz
            CHK(hipMemcpy(&er[1 + mode], errs, 4, hipMemcpyDeviceToHost));
        }
        printf("%-8d %-10d | %8.2f  (errors %u)%8s | %8.2f  (errors %u)%14s | %8.2f  (errors %u)\n", G, W * 4, r[0], er[0], "", r[1], er[1], "", r[2], er[2]);
    }
    return 0;
}
