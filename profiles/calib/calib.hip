// profiles/calib/calib.hip -- calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for THIS project's access patterns
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced reads by 2x; other widths are uncalibrated -- calibrate
// on a known byte count in your own access pattern).  Three kernels over 1 GiB buffers (>> 256 MiB Infinity Cache):
//   k_stream_read   float4 per lane, fully coalesced                         known: 1 GiB read
//   k_gather24      per lane: dwordx4 + dwordx2 at a 12-B-aligned address in its own 128-B line (the {I,dx,dy} tap pair of
//                   interp33), lines visited in a scrambled order             known: N lines touched = N*128 B if whole lines move
//   k_plane_store   4 B per lane, 256 B per wave-instruction (the J-plane stores of k_ef_linearize)   known: 1 GiB written
// Build: hipcc --offload-arch=gfx950 -O3 calib.hip -o calib ; run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_stream_read(const float4* __restrict__ in, size_t n, float* __restrict__ out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_gather24(const float* __restrict__ in, size_t nlines, float* __restrict__ out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * 2654435761ull) % nlines;                 // scrambled, each line exactly once (odd multiplier, nlines = 2^k)
        const float* p = in + line * 32 + 3 * ((i >> 3) & 7);             // 12-B-aligned start inside the line, 24 B stay inside it
        acc += p[0] + p[1] + p[2] + p[3] + p[4] + p[5];
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_plane_store(float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
int main() {
    const size_t bytes = 1ull << 30;
    float *a, *o;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&o, 256));
    CHK(hipMemset(a, 0, bytes));
    CHK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, 0, (const float4*)a, bytes / 16, o);
        hipLaunchKernelGGL(k_gather24, dim3(4096), dim3(256), 0, 0, (const float*)a, bytes / 128, o);
        hipLaunchKernelGGL(k_plane_store, dim3(4096), dim3(256), 0, 0, a, bytes / 4);
    }
    CHK(hipDeviceSynchronize());
    printf("calib done: 1 GiB streamed, %zu lines gathered (24 B each), 1 GiB stored\n", bytes / 128);
    return 0;
}
