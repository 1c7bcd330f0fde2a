// backend_stitch.hpp -- k_ef_stitch: the double-precision "stitch" of solveSystemF on the device.
//
// Replaces, for the solve path, the host loops stitch_top / stitch_sc of backend.hip (which restate
//   AccumulatedTopHessianSSE::stitchDoubleInternal + stitchDoubleMT tail   AccumulatedTopHessian.cpp:181-242, .h:63-114
//   AccumulatedSCHessianSSE::stitchDoubleInternal + stitchDoubleMT tail    AccumulatedSCHessian.cpp:64-135, .h:108-113
// ) and the assembly  HFinal = HA + HM - Hsc,  bFinal = bA + bM + HM*delta - bsc  (EnergyFunctional.cpp:668-699).
// One workgroup; output-stationary: every element of the (4+6nF)^2 system is owned by one lane, which adds its
// contributions in a fixed order (deterministic, no atomics).  Only HFinal/bFinal (22 kB at nF=8) leave the device instead
// of the 295 kB packed accumulator buffer; the host keeps the damped LDLT.
#pragma once
#include <hip/hip_runtime.h>

namespace sdvgn {

constexpr int kStitchThreads = 1024;
constexpr int kMaxDim = 4 + 6 * 8;

struct StitchState {          // small per-solve inputs, uploaded with the precalc records
    double delta[kMaxDim];    // getStitchedDeltaF(): [cDeltaF | frame deltas]
    double prior[8 * 6];      // EFFrame::prior
    double delta_prior[8 * 6];
    double cPrior[4];
    double cDelta[4];         // (double)cDeltaF
};

__device__ __forceinline__ double top_g(const double* __restrict__ top, int nF, int h, int t, int r, int c) {
    return top[(size_t)(h * nF + t) * 256 + r * 16 + c];
}
__device__ __forceinline__ double sc_g(const double* __restrict__ sc, int h, int row, int col) {
    int ti = row >> 4, tj = col >> 4;
    if (ti > tj) { const int x = row; row = col; col = x; const int y = ti; ti = tj; tj = y; }
    const int a = ti * 4 - (ti * (ti - 1)) / 2 + (tj - ti);
    return sc[((size_t)h * 10 + a) * 256 + (row & 15) * 16 + (col & 15)];
}

// acc: packed accumulators (top | sc | resInA); adH: [nF*nF][36] doubles, index h + t*nF (adHost; adTarget = diag(sT))
// out: HFinal [n*n] then bFinal [n]
__global__ void __launch_bounds__(kStitchThreads) k_ef_stitch(int nF, const double* __restrict__ acc, const double* __restrict__ adH,
                                                             const StitchState* __restrict__ st, const double* __restrict__ HM,
                                                             const double* __restrict__ bM, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = 4 + 6 * nF, nf6 = 6 * nF, pairs = nF * nF;
    double* AH = sm;                       // [pairs][36]   (index h + t*nF)
    double* T1 = AH + pairs * 36;          // [pairs][36]   AH * A66, device pair index h*nF + t
    double* B = T1 + pairs * 36;           // [nF][6][nf6]  A_h * D_h
    double* Ds = B + nF * 6 * nf6;         // [nf6][nf6]    sum_h D_h
    const double* top = acc;
    const double* sc = acc + (size_t)pairs * 256;
    const int tid = threadIdx.x;
    const double sT[6] = {0.5, 0.5, 0.5, 1.0, 1.0, 1.0};   // adTarget = diag(SCALE_XI_TRANS x3, SCALE_XI_ROT x3)

    for (int i = tid; i < pairs * 36; i += kStitchThreads) AH[i] = adH[i];
    __syncthreads();
    // ---- phase 1: T1, B, Dsum
    for (int o = tid; o < pairs * 36; o += kStitchThreads) {
        const int pr = o / 36, r = (o % 36) / 6, c = o % 6;
        const int h = pr / nF, t = pr % nF;
        double s = 0;
        if (h != t) {
            const double* A = AH + (size_t)(h + t * nF) * 36;
            for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * top_g(top, nF, h, t, 4 + q, 4 + c);
        }
        T1[o] = s;
    }
    for (int o = tid; o < nF * 6 * nf6; o += kStitchThreads) {
        const int h = o / (6 * nf6), r = (o / nf6) % 6, c = o % nf6;
        double s = 0;
        for (int j = 0; j < nF; ++j) {
            const double* A = AH + (size_t)(h + j * nF) * 36;
            for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * sc_g(sc, h, 6 * j + q, c);
        }
        B[o] = s;
    }
    for (int o = tid; o < nf6 * nf6; o += kStitchThreads) {
        const int x = o / nf6, y = o % nf6;
        double s = 0;
        for (int h = 0; h < nF; ++h) s += sc_g(sc, h, x, y);
        Ds[o] = s;
    }
    __syncthreads();
    // ---- phase 2: one lane per output element
    for (int o = tid; o < n * n + n; o += kStitchThreads) {
        const bool isb = o >= n * n;
        int a = isb ? o - n * n : o / n, b = isb ? -1 : o % n;
        if (!isb && a < 4 && b >= 4) { const int x = a; a = b; b = x; }   // calib-frame entries are transposed copies
        double hA = 0, hS = 0;
        const int i = a >= 4 ? (a - 4) / 6 : -1, r = a >= 4 ? (a - 4) % 6 : a;
        if (isb) {
            if (a < 4) {
                for (int pr = 0; pr < pairs; ++pr) hA += top[(size_t)pr * 256 + a * 16 + 10];
                hA += st->cPrior[a] * st->cDelta[a];
                for (int h = 0; h < nF; ++h) hS += sc_g(sc, h, 48 + a, 52);
            } else {
                for (int t = 0; t < nF; ++t) {
                    if (t == i) continue;
                    const double* A = AH + (size_t)(i + t * nF) * 36;
                    double s = 0;
                    for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * top_g(top, nF, i, t, 4 + q, 10);
                    hA += s;
                    hA += sT[r] * top_g(top, nF, t, i, 4 + r, 10);   // i as target of host t
                }
                hA += st->prior[i * 6 + r] * st->delta_prior[i * 6 + r];
                for (int j = 0; j < nF; ++j) {
                    const double* A = AH + (size_t)(i + j * nF) * 36;
                    double s = 0;
                    for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * sc_g(sc, i, 6 * j + q, 52);
                    hS += s;
                }
                for (int h = 0; h < nF; ++h) hS += sT[r] * sc_g(sc, h, 6 * i + r, 52);
            }
            double hm = bM[a];
            for (int j = 0; j < n; ++j) hm += HM[(size_t)a * n + j] * st->delta[j];
            out[(size_t)n * n + a] = hA + hm - hS;
            continue;
        }
        if (a < 4) {   // calib-calib
            for (int pr = 0; pr < pairs; ++pr) hA += top[(size_t)pr * 256 + a * 16 + b];
            if (a == b) hA += st->cPrior[a];
            for (int h = 0; h < nF; ++h) hS += sc_g(sc, h, 48 + a, 48 + b);
        } else if (b < 4) {   // frame(i,r) - calib(b)
            for (int t = 0; t < nF; ++t) {
                if (t == i) continue;
                const double* A = AH + (size_t)(i + t * nF) * 36;
                double s = 0;
                for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * top_g(top, nF, i, t, 4 + q, b);
                hA += s;
                hA += sT[r] * top_g(top, nF, t, i, 4 + r, b);
            }
            for (int j = 0; j < nF; ++j) {
                const double* A = AH + (size_t)(i + j * nF) * 36;
                double s = 0;
                for (int q = 0; q < 6; ++q) s += A[r * 6 + q] * sc_g(sc, i, 6 * j + q, 48 + b);
                hS += s;
            }
            for (int h = 0; h < nF; ++h) hS += sT[r] * sc_g(sc, h, 6 * i + r, 48 + b);
        } else {   // frame(i,r) - frame(k,c)
            const int k = (b - 4) / 6, c = (b - 4) % 6;
            if (i == k) {
                for (int t = 0; t < nF; ++t) {
                    if (t == i) continue;
                    const double* A = AH + (size_t)(i + t * nF) * 36;
                    const double* T = T1 + (size_t)(i * nF + t) * 36;
                    double s = 0;
                    for (int q = 0; q < 6; ++q) s += T[r * 6 + q] * A[c * 6 + q];
                    hA += s;
                    hA += sT[r] * top_g(top, nF, t, i, 4 + r, 4 + c) * sT[c];
                }
                if (r == c) hA += st->prior[i * 6 + r];
                const double* Bi = B + (size_t)i * 6 * nf6;
                double s = 0;
                for (int j = 0; j < nF; ++j) {
                    const double* A = AH + (size_t)(i + j * nF) * 36;
                    for (int q = 0; q < 6; ++q) s += Bi[r * nf6 + 6 * j + q] * A[c * 6 + q];
                }
                hS += s;
            } else {
                hA += T1[(size_t)(i * nF + k) * 36 + r * 6 + c] * sT[c] + T1[(size_t)(k * nF + i) * 36 + c * 6 + r] * sT[r];
            }
            hS += B[(size_t)i * 6 * nf6 + r * nf6 + 6 * k + c] * sT[c];
            hS += sT[r] * B[(size_t)k * 6 * nf6 + c * nf6 + 6 * i + r];
            hS += sT[r] * Ds[(size_t)(6 * i + r) * nf6 + 6 * k + c] * sT[c];
        }
        const int oa = o / n, ob = o % n;
        out[(size_t)oa * n + ob] = hA + HM[(size_t)oa * n + ob] - hS;
    }
}

inline size_t stitch_smem_bytes(int nF) {
    const int nf6 = 6 * nF, pairs = nF * nF;
    return sizeof(double) * ((size_t)pairs * 36 * 2 + (size_t)nF * 6 * nf6 + (size_t)nf6 * nf6);
}

}  // namespace sdvgn
