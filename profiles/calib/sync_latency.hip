// Launch + completion latency on this box: hipStreamSynchronize vs spinning on a flag the kernel writes to pinned host memory.
// build: hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_null(volatile int* flag, int v) { if (flag && threadIdx.x == 0) { __threadfence_system(); *flag = v; } }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* flag; hipHostMalloc(&flag, 64); *flag = 0;
    for (int i = 0; i < 100; ++i) { k_null<<<1, 64, 0, s>>>(nullptr, 0); hipStreamSynchronize(s); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { k_null<<<1, 64, 0, s>>>(nullptr, 0); hipStreamSynchronize(s); }
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 1; i <= 2000; ++i) { k_null<<<1, 64, 0, s>>>(flag, i); while (*(volatile int*)flag != i) {} }
    auto t2 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    for (int i = 0; i < 2000; ++i) { k_null<<<1, 64, 0, s>>>(nullptr, 0); }
    hipStreamSynchronize(s);
    auto t3 = std::chrono::steady_clock::now();
    printf("launch + hipStreamSynchronize : %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    printf("launch + spin on pinned flag  : %.2f us\n", std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    printf("back-to-back launches (async) : %.2f us each\n", std::chrono::duration<double, std::micro>(t3 - t2).count() / 2000);
    return 0;
}
