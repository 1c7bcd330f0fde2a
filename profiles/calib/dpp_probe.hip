// dpp_probe.hip -- round 6: does v_fmac_f64 with DPP row_newbcast (gfx90a+: the one DPP control 64-bit ALU operations take) issue at the plain fp64 FMA rate, and what do
// the gfx950 v_permlane{16,32}_swap cost?  One wave, lane 0 times with s_memtime.  hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__global__ void __launch_bounds__(64) probe(double* out, unsigned long long* t, int n) {
    const int lane = threadIdx.x;
    double a[32], y = 1e-3 * lane, l = 1e-7 * (lane + 1);
#pragma unroll
    for (int j = 0; j < 32; ++j) a[j] = lane * 0.001 + j;
    unsigned u0 = lane, u1 = lane * 3;
    const unsigned long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
        if (T == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(y), "v"(l), "n"(5));
        } else if (T == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(l));
        } else if (T == 2) {
#pragma unroll
            for (int r = 0; r < 32; ++r) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u0), "+v"(u1)); asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u0), "+v"(u1)); }
        } else if (T == 3) {     // dependent: permlane32_swap -> permlane16_swap -> dpp fmac -> (feeds the next swap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned lo = __double2loint(y), hi = __double2hiint(y), lo2 = lo, hi2 = hi;
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
                unsigned lo3 = lo, hi3 = hi;
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(lo3));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(hi), "+v"(hi3));
                const double yy = __hiloint2double(hi, lo);
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(yy), "v"(l), "n"(7));
            }
        }
    }
    const unsigned long long c1 = clock64();
    double s = y + (double)u0 + (double)u1;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += a[j];
    out[lane] = s;
    if (lane == 0) t[0] = c1 - c0;
}
template <int T> static void run(const char* what, int per, double* out, unsigned long long* t) {
    unsigned long long h;
    probe<T><<<1, 64>>>(out, t, 2000); hipDeviceSynchronize();
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-70s %7.2f clocks per unit\n", what, (double)h / 2000 / per);
}
__global__ void check(double* out) {      // what the swaps and the broadcast deliver
    const int lane = threadIdx.x;
    unsigned v = lane, w, v2, w2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(v));                  // (a copy in a register of its own: the swaps exchange between two registers)
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));      // v = {lo32 of v, lo32 of w} ; w = {hi32 of v, hi32 of w}
    asm volatile("v_mov_b32 %0, %1" : "=v"(v2) : "v"(v));
    asm volatile("v_mov_b32 %0, %1" : "=v"(w2) : "v"(w));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(v2));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w), "+v"(w2));
    double y = 100.0 + lane, acc = 0.0, one = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(one));
    out[lane] = v; out[64 + lane] = v2; out[128 + lane] = w; out[192 + lane] = w2; out[256 + lane] = acc;
}
int main() {
    double* out; unsigned long long* t;
    hipMalloc(&out, 512 * 8); hipMalloc(&t, 16);
    run<1>("v_fmac_f64_e32, independent (unit: 1)", 64, out, t);
    run<0>("v_fmac_f64_dpp row_newbcast, independent (unit: 1)", 64, out, t);
    run<2>("v_permlane32_swap + v_permlane16_swap, dependent pair (unit: pair)", 32, out, t);
    run<3>("2 x permlane32_swap, 2 x permlane16_swap, dpp fmac: dependent (unit: group)", 16, out, t);
    check<<<1, 64>>>(out); hipDeviceSynchronize();
    double h[320]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[5] = {"v  after 32swap,16swap(v,v2)", "v2", "w  after 32swap,16swap(w,w2)", "w2", "dpp row_newbcast:3 of (100+lane)"};
    for (int k = 0; k < 5; ++k) { printf("%-34s", nm[k]); for (int i = 0; i < 64; i += 5) printf(" %g", h[64 * k + i]); printf("  | lanes 0,5,10,..\n"); }
    return 0;
}
