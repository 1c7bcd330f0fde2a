// clock_probe.hip -- what clock does a small (1-workgroup) kernel run at?  Measures a dependent chain of N integer adds
// (4 cycles each on a wave64 SIMD16) with s_memtime (shader-clock counter) and wall_clock64 (100 MHz constant) around it,
// for 1 workgroup alone and for a full-chip launch.   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out, unsigned long long* t, int n) {
    unsigned v = threadIdx.x;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; ++i) { v = v * 3u + 1u; v ^= (v >> 3); v += i; v = v * 5u + 7u; }
    const unsigned long long c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void barrier_probe(unsigned long long* t, int n) {
    __shared__ int s[16];
    const unsigned long long w0 = wall_clock64();
    int acc = 0;
    for (int i = 0; i < n; ++i) { if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = i + acc; __syncthreads(); for (int w = 0; w < 16; ++w) acc += s[w]; }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { t[0] = w1 - w0; t[1] = acc; }
}
int main() {
    unsigned* out; unsigned long long* t; unsigned long long h[2];
    hipMalloc(&out, 4 * 1024 * 4096); hipMalloc(&t, 16);
    for (int grid : {1, 1, 1024, 1}) {
        for (int n : {1000, 10000}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a); probe<<<grid, 256>>>(out, t, n); hipEventRecord(b); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("grid %4d n %5d: s_memtime %llu ticks, wall_clock64 %llu ticks (100 MHz => %.2f us), event %.2f us; ticks/iter %.2f, ns/iter %.2f\n", grid, n, h[0], h[1], h[1] / 100.0,
                   ms * 1e3, (double)h[0] / n, h[1] * 10.0 / n);
        }
    }
    for (int n : {16, 160}) {
        barrier_probe<<<1, 1024>>>(t, n); hipDeviceSynchronize();
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("barrier_probe 1024 lanes, %d x (lds write + barrier + 16 lds reads): %.2f us total, %.1f ns per step\n", n, h[0] / 100.0, h[0] * 10.0 / n);
    }
    return 0;
}
