// onewave_probe.hip -- what ONE wave can issue per clock on gfx950 (round 6: sizing the one-wave LDL^T of k_ef_tail_resub).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off onewave_probe.hip -o onewave_probe
// Each test: N repetitions of an unrolled body, timed with s_memtime (clock64) and wall_clock64 (100 MHz) by lane 0 of the only wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int T>
__global__ void __launch_bounds__(64) probe(double* out, unsigned long long* t, int n) {
    __shared__ __attribute__((aligned(16))) double sh[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) sh[i] = 1e-9 * i;
    __syncthreads();
    double a[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) a[j] = lane * 0.001 + j;
    double l = 1e-7 * lane;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < n; ++it) {
        if (T == 0) {            // 32 independent fp64 FMAs, register operands
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) a[j] = __builtin_fma(-l, a[(j + 1) & 31], a[j]);
        } else if (T == 1) {     // dependent chain of fp64 FMAs
#pragma unroll
            for (int r = 0; r < REP; ++r) a[0] = __builtin_fma(-l, a[0], a[1]);
        } else if (T == 2) {     // broadcast ds_read_b128 + 2 FMAs each (the LDL^T's update)
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const double2 c = *reinterpret_cast<const double2*>(&sh[(it & 63) * 54 + j + r * 32]);
                    a[j] = __builtin_fma(-l, c.x, a[j]); a[j + 1] = __builtin_fma(-l, c.y, a[j + 1]);
                }
        } else if (T == 3) {     // ds_read_b128 broadcast only (values summed sparsely)
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                const double2 c = *reinterpret_cast<const double2*>(&sh[(it & 63) * 54 + 2 * r]);
                if (r % 16 == 0) a[r / 16] += c.x;
                asm volatile("" :: "v"(c.x), "v"(c.y));
            }
        } else if (T == 4) {     // the pivot chain: readlane x2 -> rcp -> fma -> fma -> mul -> fma
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double d = readlane_f64(a[0], r);
                const double rc = __builtin_amdgcn_rcp(d);
                const double rinv = __builtin_fma(__builtin_fma(-d, rc, 1.0), rc, rc);
                l = a[0] * rinv;
                a[0] = __builtin_fma(-l, readlane_f64(a[1], r + 1), a[2]);
            }
        } else if (T == 5) {     // per-lane ds_read_b128 (row stride 54 doubles) + 2 FMAs
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const double2 c = *reinterpret_cast<const double2*>(&sh[lane * 54 + j + r * 32]);
                    a[j] = __builtin_fma(-l, c.x, a[j]); a[j + 1] = __builtin_fma(-l, c.y, a[j + 1]);
                }
        } else if (T >= 10 && T < 30) {   // ds_read_b128 at base + off(lane) + j, 2 FMAs each: how lanes share addresses
            int off = 0;
            if (T == 10) off = 0;                       // all lanes one address (broadcast)
            if (T == 11) off = (lane & 1) * 66;         // 2 copies
            if (T == 12) off = (lane & 3) * 66;         // 4 copies, lanes interleaved
            if (T == 13) off = (lane & 7) * 66;         // 8 copies
            if (T == 14) off = (lane & 15) * 66;        // 16 copies
            if (T == 15) off = (lane >> 4) * 66;        // 4 copies, 16 consecutive lanes share
            if (T == 16) off = (lane >> 3) * 66;        // 8 copies, 8 consecutive lanes share
            if (T == 17) off = (lane >> 5) * 66;        // 2 copies, halves
            if (T == 18) off = (lane & 31) * 66;        // 32 copies
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const double2 c = *reinterpret_cast<const double2*>(&sh[off + j + r * 32]);
                    a[j] = __builtin_fma(-l, c.x, a[j]); a[j + 1] = __builtin_fma(-l, c.y, a[j + 1]);
                }
        } else if (T == 30) {    // broadcast ds_read_b64 + 1 FMA
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) { const double c = sh[(it & 63) * 54 + j + r * 32]; a[j] = __builtin_fma(-l, c, a[j]); }
        } else if (T == 31) {    // v_readlane x2 + FMA with the SGPR pair
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) a[j] = __builtin_fma(-l, readlane_f64(a[(j + 7) & 31], j), a[j]);
        } else if (T == 40 || T == 41 || T == 42) {   // T == 2 with the issue order pinned: one read, two FMAs (40); two reads, four FMAs (41); reads software-pipelined one iteration ahead (42)
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const double2 c = *reinterpret_cast<const double2*>(&sh[(it & 63) * 54 + j + r * 32]);
                    a[j] = __builtin_fma(-l, c.x, a[j]); a[j + 1] = __builtin_fma(-l, c.y, a[j + 1]);
                }
            if (T == 40) {
#pragma unroll
                for (int i = 0; i < REP / 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
            } else if (T == 41) {
#pragma unroll
                for (int i = 0; i < 8; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int i = 0; i < REP / 2 - 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (T == 6) {     // fp32 FMAs for comparison
            float f[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) f[j] = (float)a[j];
#pragma unroll
            for (int r = 0; r < REP / 32; ++r)
#pragma unroll
                for (int j = 0; j < 32; ++j) f[j] = __builtin_fmaf(-(float)l, f[(j + 1) & 31], f[j]);
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = f[j];
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    double s = l;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += a[j];
    out[lane] = s;
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int T>
static void run(const char* what, int per_iter, double* out, unsigned long long* t) {
    unsigned long long h[2];
    for (int n : {200, 2000}) {
        probe<T><<<1, 64>>>(out, t, n); hipDeviceSynchronize();
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("%-58s n %5d: %8llu s_memtime ticks, %7.2f us wall; per unit: %6.2f ticks, %6.2f ns\n", what, n, h[0], h[1] / 100.0, (double)h[0] / n / per_iter, h[1] * 10.0 / n / per_iter);
    }
}
int main() {
    double* out; unsigned long long* t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    run<0>("independent v_fma_f64 (unit: 1 FMA)", REP, out, t);
    run<1>("dependent v_fma_f64 chain (unit: 1 FMA)", REP, out, t);
    run<6>("independent v_fma_f32 (unit: 1 FMA, + 64 cvt per iter)", REP, out, t);
    run<2>("broadcast ds_read_b128 + 2 FMA (unit: 1 read + 2 FMA)", REP / 2, out, t);
    run<3>("broadcast ds_read_b128 alone (unit: 1 read)", REP, out, t);
    run<5>("per-lane ds_read_b128 stride 432 B + 2 FMA (unit: 1 read)", REP / 2, out, t);
    run<4>("pivot chain readlane-rcp-newton-mul-fma (unit: 1 pivot)", 8, out, t);
    run<30>("broadcast ds_read_b64 + 1 FMA (unit: 1 read + 1 FMA)", REP, out, t);
    run<31>("2 v_readlane + FMA (unit: 1 FMA)", REP, out, t);
    run<10>("b128 + 2 FMA, 1 address (unit: read + 2 FMA)", REP / 2, out, t);
    run<11>("b128 + 2 FMA, 2 copies lane&1", REP / 2, out, t);
    run<12>("b128 + 2 FMA, 4 copies lane&3", REP / 2, out, t);
    run<13>("b128 + 2 FMA, 8 copies lane&7", REP / 2, out, t);
    run<14>("b128 + 2 FMA, 16 copies lane&15", REP / 2, out, t);
    run<18>("b128 + 2 FMA, 32 copies lane&31", REP / 2, out, t);
    run<15>("b128 + 2 FMA, 4 copies lane>>4", REP / 2, out, t);
    run<16>("b128 + 2 FMA, 8 copies lane>>3", REP / 2, out, t);
    run<17>("b128 + 2 FMA, 2 copies lane>>5", REP / 2, out, t);
    run<40>("b128 broadcast + 2 FMA, pinned read,fma,fma (unit: pair)", REP / 2, out, t);
    run<41>("b128 broadcast + 2 FMA, 8 reads ahead then interleaved", REP / 2, out, t);
    return 0;
}
