// tracker_kernels.hpp -- gfx950 kernels of the photometric coarse tracker.
//
//   k_pyr_level   a1  FrameHessian::makeImages               src/FullSystem/HessianBlocks.cpp:107-167
//   k_res_gs      a4  CoarseTracker::calcRes                 src/FullSystem/CoarseTracker.cpp:486-634
//                 a5  CoarseTracker::calcGSSSE               src/FullSystem/CoarseTracker.cpp:427-484
//                 a6  Accumulator9 (the 45-term sum)         src/OptimizationBackend/MatrixAccumulators.h:934-1115
//                 a9  getInterpolatedElement33               src/util/globalFuncs.h:51-65
//   k_finalize        1/n normalisation + SCALE_* of H,b     CoarseTracker.cpp:468-483, Vec6 of :625-633
//
// Design (MI355X): the reference runs calcRes (scalar loop writing 8 warped planes) and calcGSSSE (SSE loop
// re-reading them into an in-memory 45x4-lane accumulator).  Here one pass does both: a lane owns a reference
// point, loads its packed {u,v,idepth,colour} with one coalesced 16-B load, gathers the 2x2 {I,dx,dy} taps of
// the target level and forms residual / Huber weight / the 8 Jacobian entries in registers.  Reduction: the wave
// stages its 64 rows [J r | E counters flow] as a 16-feature LDS tile and the matrix cores accumulate the weighted Gram
// (v_mfma_f32_16x16x4_f32, 2 per 4 points) -- no per-lane 45-term accumulator, no shuffle tree; one LDS stage across
// the waves of a workgroup, one float partial row per workgroup; k_finalize adds the partial rows in a fixed order in
// fp64 (deterministic, no atomics) and emits the scaled 8x8 H, b and the Vec6.
//
// Per-point arithmetic is written operation by operation like the reference and the translation unit is built
// with -ffp-contract=off, so residuals, weights and Jacobian entries are bit-identical to the CPU path; only
// the order of the sums differs (tree instead of sequential/tiered).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace sdvgn {

// Precision modes of the tolerance study (BASELINE.json configs[4]); mode 0 is the product path, the others exist only
// to measure what reduced precision would cost (tests/test_fp16_study_gpu.py).
//   0  fp32 pyramid, fp32 arithmetic, fp32 MFMA accumulation (reference-faithful)
//   1  fp16 pyramid {I,dx,dy,0} (8 B/px, one 8-B load per tap), everything else fp32
//   2  mode 1 + Jacobian/residual operands rounded to fp16 before the Gram (== f16-input MFMA, fp32 accumulate)
//   3  mode 2 + the Gram accumulator rounded to fp16 after every 4-point MFMA step ("fp16 residual accumulation")
enum { PREC_F32 = 0, PREC_H_PYR = 1, PREC_H_OPER = 2, PREC_H_ACC = 3,
       // fp32 arithmetic on the GATHER-FRIENDLY copy of the pyramid (k_pyr_to_records): per pixel one 64-byte record holding the {I,dx,dy}
       // of its 2x2 neighbourhood -- the four taps of a bilinear lookup at any position come out of ONE 64-byte-aligned record (three
       // 16-byte loads, one 128-byte line) instead of two image rows = two to four lines.  Same 12 floats, same operations: results are
       // bit-identical to PREC_F32; only the addresses change.  (Negative so that the `MODE >= PREC_H_*` tests of the study modes stay false.)
       PREC_F32_REC = -1 };

constexpr int kNAcc = 45;                 // upper triangle of the 9x9 [J r][J r]^T
constexpr int kNRed = 52;                 // 45 + E + nE + nSat + nWarped + flowT + flowRT + flowNum
constexpr int kRedE = 45, kRedNE = 46, kRedNSat = 47, kRedNW = 48, kRedFT = 49, kRedFRT = 50, kRedFN = 51;
constexpr int kOutStride = 80;            // doubles per problem: out6[6] H[64] b[8] pad[2]

// HessianBlocks.h:33-40
#define SDVGN_SCALE_XI_ROT 1.0f
#define SDVGN_SCALE_XI_TRANS 0.5f
#define SDVGN_SCALE_A 10.0f
#define SDVGN_SCALE_B 1000.0f

struct LevelParams {  // everything calcRes/calcGSSSE derive from (lvl, refToNew, aff_g2l) before their loops
    float RKi[9];
    float t[3];
    float Ki[9];
    float fx, fy, cx, cy;
    float affLL0, affLL1;  // AffLight::fromToVecExposure(...).cast<float>()
    float b0;              // lastRef_aff_g2l.b
    float cutoff, huber, maxEnergy;
    int wl, hl, lvl, n;
};

// a9: bilinear {I,dx,dy} with truncating index and the reference's weight/tap order
__device__ __forceinline__ void interp33(const float* __restrict__ img, float x, float y, int width, float& o0,
                                         float& o1, float& o2) {
    const int ix = (int)x;
    const int iy = (int)y;
    const float dx = x - ix;
    const float dy = y - iy;
    const float dxdy = dx * dy;
    const float* bp = img + 3 * (ix + iy * width);
    const float* bq = bp + 3 * width;
    const float a0 = bp[0], a1 = bp[1], a2 = bp[2], b0 = bp[3], b1 = bp[4], b2 = bp[5];
    const float c0 = bq[0], c1 = bq[1], c2 = bq[2], d0 = bq[3], d1 = bq[4], d2 = bq[5];
    // all twelve values in flight together: without this the compiler sinks the loads of the gradient components (used only once the
    // intensity has proved finite) under that branch -- a second round trip per lookup behind the first
    asm volatile("" ::"v"(b1), "v"(b2), "v"(d1), "v"(d2));
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    o0 = ((w11 * d0 + w01 * c0) + w10 * b0) + w00 * a0;
    o1 = ((w11 * d1 + w01 * c1) + w10 * b1) + w00 * a1;
    o2 = ((w11 * d2 + w01 * c2) + w10 * b2) + w00 * a2;
}

// PREC_F32_REC: the four taps from the pixel's neighbourhood record [a(3) b(3) c(3) d(3) pad(4)] (a = (ix,iy), b = (ix+1,iy), c = (ix,iy+1), d = (ix+1,iy+1))
__device__ __forceinline__ void interp33_rec(const float* __restrict__ rec16, float x, float y, int width, float& o0, float& o1, float& o2) {
    const int ix = (int)x;
    const int iy = (int)y;
    const float dx = x - ix;
    const float dy = y - iy;
    const float dxdy = dx * dy;
    const float4* r = reinterpret_cast<const float4*>(rec16) + 4 * (size_t)(ix + iy * width);
    const float4 r0 = r[0], r1 = r[1], r2 = r[2];
    const float a0 = r0.x, a1 = r0.y, a2 = r0.z, b0 = r0.w, b1 = r1.x, b2 = r1.y;
    const float c0 = r1.z, c1 = r1.w, c2 = r2.x, d0 = r2.y, d1 = r2.z, d2 = r2.w;
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    o0 = ((w11 * d0 + w01 * c0) + w10 * b0) + w00 * a0;
    o1 = ((w11 * d1 + w01 * c1) + w10 * b1) + w00 * a1;
    o2 = ((w11 * d2 + w01 * c2) + w10 * b2) + w00 * a2;
}

// mode >= 1: the same bilinear formula on an fp16 {I,dx,dy,pad} pyramid (values widen to fp32 before the arithmetic)
__device__ __forceinline__ void interp33_h(const __half* __restrict__ img4, float x, float y, int width, float& o0, float& o1, float& o2) {
    const int ix = (int)x;
    const int iy = (int)y;
    const float dx = x - ix;
    const float dy = y - iy;
    const float dxdy = dx * dy;
    const uint2* bp = reinterpret_cast<const uint2*>(img4) + (ix + iy * width);
    const uint2 ra = bp[0], rb = bp[1], rc = bp[width], rd = bp[width + 1];
    auto lo = [](unsigned v) { return __half2float(__ushort_as_half((unsigned short)(v & 0xffffu))); };
    auto hi = [](unsigned v) { return __half2float(__ushort_as_half((unsigned short)(v >> 16))); };
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    o0 = ((w11 * lo(rd.x) + w01 * lo(rc.x)) + w10 * lo(rb.x)) + w00 * lo(ra.x);
    o1 = ((w11 * hi(rd.x) + w01 * hi(rc.x)) + w10 * hi(rb.x)) + w00 * hi(ra.x);
    o2 = ((w11 * lo(rd.y) + w01 * lo(rc.y)) + w10 * lo(rb.y)) + w00 * lo(ra.y);
}

__device__ __forceinline__ void project(const float* M, const float* t, float x, float y, float id, bool plus,
                                        float& p0, float& p1, float& p2) {
    const float m0 = (M[0] * x + M[1] * y) + M[2] * 1.0f;
    const float m1 = (M[3] * x + M[4] * y) + M[5] * 1.0f;
    const float m2 = (M[6] * x + M[7] * y) + M[8] * 1.0f;
    if (plus) { p0 = m0 + t[0] * id; p1 = m1 + t[1] * id; p2 = m2 + t[2] * id; }
    else      { p0 = m0 - t[0] * id; p1 = m1 - t[1] * id; p2 = m2 - t[2] * id; }
}

// One reference point: calcRes body (:517-600) followed by the Jacobian row of calcGSSSE (:444-465) on the values that calcRes
// would have appended to the warped buffers.  Output = the lane's row of the wave's [16 features][64 points] tile:
//   f[0..7] = J (8 Jacobian entries), f[8] = residual     -- all 0 unless the point is an inlier (st == 1)
//   f[9] = energy term, f[10] = saturated flag, f[11] = in-warped flag (numTermsInE = their sum), f[12..14] = flow sums (T, RT, count),
//   f[15] = 0 (row 15 of the A operand is the constant 1 that turns the Gram's last row into the column sums)
//   w = Huber weight (0 unless st == 1)
template <bool WRITE_TERMS, int MODE = PREC_F32>
__device__ __forceinline__ void point_features(const LevelParams& P, const float* __restrict__ img, float4 pc, int i, bool valid,
                                               float* f, float& w, float* __restrict__ terms, int* __restrict__ status) {
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = 0.0f;
    w = 0.0f;
    if (!valid) return;
    const float x = pc.x, y = pc.y, id = pc.z, refColor = pc.w;
    float p0, p1, p2;
    project(P.RKi, P.t, x, y, id, true, p0, p1, p2);
    const float u = p0 / p2;
    const float v = p1 / p2;
    const float Ku = P.fx * u + P.cx;
    const float Kv = P.fy * v + P.cy;
    const float new_idepth = id / p2;

    if (P.lvl == 0 && (i & 31) == 0) {  // flow indicators :538-566
        float q0, q1, q2;
        project(P.Ki, P.t, x, y, id, true, q0, q1, q2);
        const float KuT = P.fx * (q0 / q2) + P.cx, KvT = P.fy * (q1 / q2) + P.cy;
        project(P.Ki, P.t, x, y, id, false, q0, q1, q2);
        const float KuT2 = P.fx * (q0 / q2) + P.cx, KvT2 = P.fy * (q1 / q2) + P.cy;
        project(P.RKi, P.t, x, y, id, false, q0, q1, q2);
        const float Ku3 = P.fx * (q0 / q2) + P.cx, Kv3 = P.fy * (q1 / q2) + P.cy;
        float fT = 0.0f, fRT = 0.0f;
        fT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
        fT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
        fRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
        fRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
        f[12] = fT; f[13] = fRT; f[14] = 2.0f;
    }

    int st = 0;
    float hit0 = 0, hit1 = 0, hit2 = 0, residual = 0, hw = 0;
    if (Ku > 2 && Kv > 2 && Ku < P.wl - 3 && Kv < P.hl - 3 && new_idepth > 0) {
        if (MODE == PREC_F32) interp33(img, Ku, Kv, P.wl, hit0, hit1, hit2);
        else if (MODE == PREC_F32_REC) interp33_rec(img, Ku, Kv, P.wl, hit0, hit1, hit2);
        else interp33_h(reinterpret_cast<const __half*>(img), Ku, Kv, P.wl, hit0, hit1, hit2);
        if (isfinite(hit0)) {
            residual = hit0 - (float)(P.affLL0 * refColor + P.affLL1);
            const float ar = fabsf(residual);
            hw = ar < P.huber ? 1.0f : P.huber / ar;
            if (ar > P.cutoff) {
                st = 2;
                f[9] = P.maxEnergy;
                f[10] = 1.0f;
            } else {
                st = 1;
                f[9] = hw * residual * residual * (2 - hw);
                f[11] = 1.0f;
                const float dx = hit1 * P.fx;
                const float dy = hit2 * P.fy;
                f[0] = new_idepth * dx;
                f[1] = new_idepth * dy;
                f[2] = 0.0f - new_idepth * (u * dx + v * dy);
                f[3] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
                f[4] = (u * v) * dy + dx * (1.0f + u * u);
                f[5] = u * dy - v * dx;
                f[6] = P.affLL0 * (P.b0 - refColor);
                f[7] = -1.0f;
                f[8] = residual;
                w = hw;
                if (MODE >= PREC_H_OPER) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) f[k] = __half2float(__float2half(f[k]));
                }
            }
        }
    }
    if (WRITE_TERMS) {
        const int n = P.n;
        const bool in = (st == 1);
        terms[0 * n + i] = in ? new_idepth : 0.0f;
        terms[1 * n + i] = in ? u : 0.0f;
        terms[2 * n + i] = in ? v : 0.0f;
        terms[3 * n + i] = in ? hit1 : 0.0f;
        terms[4 * n + i] = in ? hit2 : 0.0f;
        terms[5 * n + i] = in ? residual : 0.0f;
        terms[6 * n + i] = in ? hw : 0.0f;
        terms[7 * n + i] = in ? refColor : 0.0f;
        status[i] = st;
    }
}

// Tolerance-mode arithmetic (sdvgn_tracker_set_arith(1)): the same formulas with fused multiply-adds (contraction allowed inside this
// function only) and divisions as v_rcp_f32 + one Newton step (~1e-7 relative).  BASELINE.json's contract for this path is 1e-4 relative
// on pose increments, not bit-exact per-point floats; the exact build above stays the default and the parity reference.  About 40 % fewer
// VALU instructions per point (the fused kernel is VALU-bound in the batched run).
__device__ __forceinline__ float fast_div(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    return a * r;
}
template <bool WRITE_TERMS>
__device__ __forceinline__ void point_features_fast(const LevelParams& P, const float* __restrict__ img, float4 pc, int i, bool valid,
                                                    float* f, float& w, float* __restrict__ terms, int* __restrict__ status) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = 0.0f;
    w = 0.0f;
    if (!valid) return;
    const float x = pc.x, y = pc.y, id = pc.z, refColor = pc.w;
    const float p0 = P.RKi[0] * x + P.RKi[1] * y + P.RKi[2] + P.t[0] * id;
    const float p1 = P.RKi[3] * x + P.RKi[4] * y + P.RKi[5] + P.t[1] * id;
    const float p2 = P.RKi[6] * x + P.RKi[7] * y + P.RKi[8] + P.t[2] * id;
    const float ip2 = fast_div(1.0f, p2);
    const float u = p0 * ip2, v = p1 * ip2;
    const float Ku = P.fx * u + P.cx, Kv = P.fy * v + P.cy;
    const float new_idepth = id * ip2;
    if (P.lvl == 0 && (i & 31) == 0) {  // flow indicators :538-566
        float q0, q1, q2, iq;
        project(P.Ki, P.t, x, y, id, true, q0, q1, q2); iq = fast_div(1.0f, q2);
        const float KuT = P.fx * (q0 * iq) + P.cx, KvT = P.fy * (q1 * iq) + P.cy;
        project(P.Ki, P.t, x, y, id, false, q0, q1, q2); iq = fast_div(1.0f, q2);
        const float KuT2 = P.fx * (q0 * iq) + P.cx, KvT2 = P.fy * (q1 * iq) + P.cy;
        project(P.RKi, P.t, x, y, id, false, q0, q1, q2); iq = fast_div(1.0f, q2);
        const float Ku3 = P.fx * (q0 * iq) + P.cx, Kv3 = P.fy * (q1 * iq) + P.cy;
        f[12] = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y) + (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
        f[13] = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y) + (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
        f[14] = 2.0f;
    }
    int st = 0;
    float hit0 = 0, hit1 = 0, hit2 = 0, residual = 0, hw = 0;
    if (Ku > 2 && Kv > 2 && Ku < P.wl - 3 && Kv < P.hl - 3 && new_idepth > 0) {
        interp33(img, Ku, Kv, P.wl, hit0, hit1, hit2);
        if (isfinite(hit0)) {
            residual = hit0 - (P.affLL0 * refColor + P.affLL1);
            const float ar = fabsf(residual);
            hw = ar < P.huber ? 1.0f : fast_div(P.huber, ar);
            if (ar > P.cutoff) {
                st = 2;
                f[9] = P.maxEnergy;
                f[10] = 1.0f;
            } else {
                st = 1;
                f[9] = hw * residual * residual * (2 - hw);
                f[11] = 1.0f;
                const float dx = hit1 * P.fx;
                const float dy = hit2 * P.fy;
                f[0] = new_idepth * dx;
                f[1] = new_idepth * dy;
                f[2] = -new_idepth * (u * dx + v * dy);
                f[3] = -((u * v) * dx + dy * (1.0f + v * v));
                f[4] = (u * v) * dy + dx * (1.0f + u * u);
                f[5] = u * dy - v * dx;
                f[6] = P.affLL0 * (P.b0 - refColor);
                f[7] = -1.0f;
                f[8] = residual;
                w = hw;
            }
        }
    }
    if (WRITE_TERMS) {
        const int n = P.n;
        const bool in = (st == 1);
        terms[0 * n + i] = in ? new_idepth : 0.0f;
        terms[1 * n + i] = in ? u : 0.0f;
        terms[2 * n + i] = in ? v : 0.0f;
        terms[3 * n + i] = in ? hit1 : 0.0f;
        terms[4 * n + i] = in ? hit2 : 0.0f;
        terms[5 * n + i] = in ? residual : 0.0f;
        terms[6 * n + i] = in ? hw : 0.0f;
        terms[7 * n + i] = in ? refColor : 0.0f;
        status[i] = st;
    }
}

// ----------------------------------------------------------------------------------------------------
// The 45-term weighted sum of calcGSSSE (Accumulator9) as a Gram matrix on the matrix cores.
// A wave stages its 64 points as a [16 features][64 points] LDS tile (row stride 68 floats: 16-byte fragment reads, rows 4 banks apart),
// then per 4 points issues ONE v_mfma_f32_16x16x4_f32  G += A F^T  with  A = [w .* F (rows 0..14) ; 1 (row 15)]:
//     rows/cols 0..8 : sum_i hw_i [J_i; r_i][J_i; r_i]^T            (A = (J*w), B = J like MatrixAccumulators.h:1047-1110)
//     row 15         : column sums of F: features 9..14 = E, nSat, nWarped, flow sums (unweighted)
// The accumulator (4 VGPRs) persists over all points of the wave, so a lane needs no private 45-term accumulator and the
// cross-lane reduction is done by the MFMA itself (f32 products, f32 accumulate).
// Fragment maps (cdna guide section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg.
// ----------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kTrkTileStride = 68;   // floats per feature row: a multiple of 4 (ds_read_b128 fragments), 4 banks apart from row to row
constexpr int kTrkTileFloats = 16 * kTrkTileStride + 64;   // tile + weight row, per wave (16-byte aligned)

// MFMA round ks of a wave's 64 points takes the four points {ks, 16 + ks, 32 + ks, 48 + ks}: lane (fi = lane & 15, kq = lane >> 4) then needs
// feature fi of the 16 CONSECUTIVE points 16 kq .. 16 kq + 15 -- four 16-byte LDS reads for the values and four for their weights, all issued
// before the first MFMA (the earlier form read one float per round behind a wait: 32 serial LDS latencies per 64 points).
// Row 15 of A is all ones (column sums): the tile's feature row 15 holds 1.0 and the lanes with fi = 15 take their "weights" from that row.
template <int MODE = PREC_F32>
__device__ __forceinline__ void wave_gram_points(const float* f, float w, float* tile /*wave-private, 16-byte aligned*/, f32x4_t& G) {
    const int lane = threadIdx.x & 63;
    float* wrow = tile + 16 * kTrkTileStride;
#pragma unroll
    for (int k = 0; k < 15; ++k) tile[k * kTrkTileStride + lane] = f[k];
    tile[15 * kTrkTileStride + lane] = 1.0f;
    wrow[lane] = w;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int fi = lane & 15, kq = lane >> 4;
    const float4* fsrc = reinterpret_cast<const float4*>(tile + fi * kTrkTileStride + 16 * kq);
    const float4* wsrc = reinterpret_cast<const float4*>((fi == 15 ? tile + 15 * kTrkTileStride : wrow) + 16 * kq);
    float fv[16], wv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = fsrc[q], b = wsrc[q];
        fv[4 * q + 0] = a.x; fv[4 * q + 1] = a.y; fv[4 * q + 2] = a.z; fv[4 * q + 3] = a.w;
        wv[4 * q + 0] = b.x; wv[4 * q + 1] = b.y; wv[4 * q + 2] = b.z; wv[4 * q + 3] = b.w;
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        G = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[ks] * wv[ks], fv[ks], G, 0, 0, 0);
        if (MODE == PREC_H_ACC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) G[q] = __half2float(__float2half(G[q]));
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Workgroup-level combine: every wave dumps its D fragment to LDS, thread k < kNRed adds the waves' entries of value k in a
// fixed order in fp64 and writes dst[k].  smem: [nwaves][256] floats (may alias the staging tiles after a barrier).
template <typename OutT>
__device__ __forceinline__ void block_gram_reduce_to(const f32x4_t& G, float* smem, OutT* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    float* mine = smem + wave * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = G[r];
    __syncthreads();
    if (threadIdx.x < kNRed) {
        int off, off2 = -1;
        if (threadIdx.x < kNAcc) {   // upper-triangular (r <= c) index -> (r, c)
            int k = threadIdx.x, r = 0;
            while (k >= 9 - r) { k -= 9 - r; ++r; }
            off = r * 16 + (r + k);
        } else {
            // row 15 = column sums.  kRedE..kRedFN = E, nE (= nSat + nW), nSat, nW, flowT, flowRT, flowN
            const int col[7] = {9, 10, 10, 11, 12, 13, 14};
            off = 15 * 16 + col[threadIdx.x - kNAcc];
            if (threadIdx.x == kRedNE) off2 = 15 * 16 + 11;
        }
        double s = 0;
        for (int wv = 0; wv < nwaves; ++wv) {
            s += (double)smem[wv * 256 + off];
            if (off2 >= 0) s += (double)smem[wv * 256 + off2];
        }
        dst[threadIdx.x] = (OutT)s;
    }
}

// grid = (chunks, B).  params[b] describes problem b (pose/affine specific).  All problems share the reference points `pc` and the
// target level image `img` (the hypotheses of one frame) unless `ptrs` is given: then problem b works on its own template and its own
// image (independent tracking problems side by side -- many frames / agents per launch; same level geometry and point count).
// ARITH 0: the reference's arithmetic operation by operation (default, bit-exact terms); 1: tolerance mode (point_features_fast).
// partial: [B][chunks][kNRed] floats.
struct ProblemPtrs { const float4* pc; const float* img; };
template <bool WRITE_TERMS, int MODE = PREC_F32, int ARITH = 0>
__global__ void __launch_bounds__(256) k_res_gs(const float4* __restrict__ pc, const float* __restrict__ img,
                                                const LevelParams* __restrict__ params, float* __restrict__ partial,
                                                float* __restrict__ terms, int* __restrict__ status, const ProblemPtrs* __restrict__ ptrs = nullptr) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kTrkTileFloats];
    const int prob = blockIdx.y;
    const LevelParams P = params[prob];
    if (ptrs) { pc = ptrs[prob].pc; img = ptrs[prob].img; }
    const int wave = threadIdx.x >> 6;
    float* tile = smem + wave * kTrkTileFloats;
    f32x4_t G = {0, 0, 0, 0};
    const int stride = gridDim.x * blockDim.x;
    // wave-uniform trip count: every lane of a wave takes part in every MFMA round (out-of-range lanes contribute zeros)
    for (int base = blockIdx.x * blockDim.x; base < P.n; base += stride) {
        const int i = base + threadIdx.x;
        const bool valid = i < P.n;
        float f[16], w;
        if (ARITH == 0) point_features<WRITE_TERMS, MODE>(P, img, valid ? pc[i] : make_float4(0, 0, 0, 0), i, valid, f, w, terms, status);
        else point_features_fast<WRITE_TERMS>(P, img, valid ? pc[i] : make_float4(0, 0, 0, 0), i, valid, f, w, terms, status);
        wave_gram_points<MODE>(f, w, tile, G);
    }
    __syncthreads();   // staging tiles are dead: reuse smem for the combine
    block_gram_reduce_to<float>(G, smem, partial + ((size_t)prob * gridDim.x + blockIdx.x) * kNRed);
}

// Tail of calcGSSSE (:468-483: 1/n, cast to double, SCALE_* on rows and columns) and of calcRes (:625-633: the Vec6)
// from the 52 totals S.  Threads 0..72 of the calling workgroup each write a few of the kOutStride doubles of o.
__device__ __forceinline__ void finalize_outputs(const double* S, int tid, double* o) {
    const int nW = (int)S[kRedNW];
    const int n = (nW + 3) & ~3;              // zero-padded to a multiple of 4 (:603-615)
    const double inv_n = (double)(1.0f / n);  // `* (1.0f/n)` :469-470
    const float sc[8] = {SDVGN_SCALE_XI_ROT, SDVGN_SCALE_XI_ROT, SDVGN_SCALE_XI_ROT, SDVGN_SCALE_XI_TRANS,
                         SDVGN_SCALE_XI_TRANS, SDVGN_SCALE_XI_TRANS, SDVGN_SCALE_A, SDVGN_SCALE_B};
    if (tid < 72) {
        int r, c;
        if (tid < 64) { r = tid >> 3; c = tid & 7; } else { r = tid - 64; c = 8; }
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        const int idx = lo * 9 - (lo * (lo - 1)) / 2 + (hi - lo);  // upper-triangular row-major index
        double v = S[idx] * inv_n;
        if (c < 8) { v *= sc[c]; v *= sc[r]; o[6 + r * 8 + c] = v; }
        else       { v *= sc[r]; o[6 + 64 + r] = v; }
    }
    if (tid == 72) {
        const float E = (float)S[kRedE];
        const float fT = (float)S[kRedFT], fRT = (float)S[kRedFRT], fN = (float)S[kRedFN];
        const int nE = (int)S[kRedNE], nSat = (int)S[kRedNSat];
        o[0] = E;
        o[1] = nE;
        o[2] = fT / (fN + 0.1);
        o[3] = 0;
        o[4] = fRT / (fN + 0.1);
        o[5] = nSat / (float)nE;
        o[78] = nW;
        o[79] = n;
    }
}

// Deterministic fp64 combine of the partial rows + the tail of calcGSSSE (:468-483) and calcRes (:625-633).
// grid = B, block = 128.  out: [B][kOutStride] doubles.
static __global__ void __launch_bounds__(128) k_finalize(const float* __restrict__ partial, int chunks, double* __restrict__ out,
                                                         volatile int* done_flag = nullptr, int done_seq = 0) {
    __shared__ double S[kNRed];
    const int b = blockIdx.x;
    if (threadIdx.x < kNRed) {
        double s = 0;
        const float* p = partial + (size_t)b * chunks * kNRed + threadIdx.x;
        for (int c = 0; c < chunks; ++c) s += (double)p[(size_t)c * kNRed];
        S[threadIdx.x] = s;
    }
    __syncthreads();
    finalize_outputs(S, threadIdx.x, out + (size_t)b * kOutStride);
    if (done_flag) {   // single-hypothesis host-driven trial: publish completion to the spinning host (waitflag.hpp)
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence_system(); *done_flag = done_seq; }
    }
}

// ----------------------------------------------------------------------------------------------------
// a1: one pyramid level per launch.  A thread owns a 2x2 quad of level-l pixels: it writes their {I,dx,dy}
// (I is copied from `src_plane` on level 0, already in place otherwise) and the 2x2 mean into level l+1.
// Gradients use flat indices like the reference (idx+-1 wraps across rows at x=0 / x=wl-1); rows 0 and hl-1,
// which the reference leaves uninitialised, are written as 0.
// ----------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_pyr_level(const float* __restrict__ src_plane, float* __restrict__ aos,
                                                   float* __restrict__ aos_next, int wl, int hl, int has_next) {
    const int qx = blockIdx.x * blockDim.x + threadIdx.x;
    const int qy = blockIdx.y;
    const int qw = (wl + 1) >> 1;
    if (qx >= qw) return;
    const int wn = wl >> 1, hn = hl >> 1;
    const int total = wl * hl;
    float Iq[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = 2 * qx + (k & 1), y = 2 * qy + (k >> 1);
        if (x >= wl || y >= hl) continue;
        const int idx = x + y * wl;
        auto I_at = [&](int j) -> float { return src_plane ? src_plane[j] : aos[3 * j]; };
        const float I = I_at(idx);
        Iq[k] = I;
        float dx = 0.0f, dy = 0.0f;
        if (idx >= wl && idx < total - wl) {
            dx = 0.5f * (I_at(idx + 1) - I_at(idx - 1));
            dy = 0.5f * (I_at(idx + wl) - I_at(idx - wl));
            if (!isfinite(dx)) dx = 0;
            if (!isfinite(dy)) dy = 0;
        }
        if (src_plane) aos[3 * idx] = I;
        aos[3 * idx + 1] = dx;
        aos[3 * idx + 2] = dy;
    }
    if (has_next && qx < wn && qy < hn)
        aos_next[3 * (qx + qy * wn)] = 0.25f * (((Iq[0] + Iq[1]) + Iq[2]) + Iq[3]);
}

// Gather-friendly copy of one pyramid level (PREC_F32_REC): record (x, y) = the {I,dx,dy} of pixels (x,y), (x+1,y), (x,y+1), (x+1,y+1) + 16 bytes of
// padding = 64 bytes.  The last column / row have no right / lower neighbour: those slots are 0 (calcRes only looks up positions with
// Ku < wl - 3, Kv < hl - 3).  4.3x the memory of the AoS level (30 MB at 1241x376), written once per frame.
static __global__ void __launch_bounds__(256) k_pyr_to_records(const float* __restrict__ aos3, float4* __restrict__ rec, int w, int h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int x = i % w, y = i / w;
    const bool xr = x + 1 < w, yd = y + 1 < h;
    const float* a = aos3 + 3 * (size_t)i;
    const float* b = a + 3; const float* c = a + 3 * (size_t)w; const float* d = c + 3;
    const float b0 = xr ? b[0] : 0.f, b1 = xr ? b[1] : 0.f, b2 = xr ? b[2] : 0.f;
    const float c0 = yd ? c[0] : 0.f, c1 = yd ? c[1] : 0.f, c2 = yd ? c[2] : 0.f;
    const float d0 = (xr && yd) ? d[0] : 0.f, d1 = (xr && yd) ? d[1] : 0.f, d2 = (xr && yd) ? d[2] : 0.f;
    float4* r = rec + 4 * (size_t)i;
    r[0] = make_float4(a[0], a[1], a[2], b0);
    r[1] = make_float4(b1, b2, c0, c1);
    r[2] = make_float4(c2, d0, d1, d2);
    r[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// fp16 copy of one pyramid level for the precision study: AoS float3 -> half4 {I,dx,dy,0}
static __global__ void __launch_bounds__(256) k_pyr_to_half(const float* __restrict__ aos3, __half* __restrict__ out4, int npix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    out4[4 * i + 0] = __float2half(aos3[3 * i + 0]);
    out4[4 * i + 1] = __float2half(aos3[3 * i + 1]);
    out4[4 * i + 2] = __float2half(aos3[3 * i + 2]);
    out4[4 * i + 3] = __float2half(0.0f);
}

}  // namespace sdvgn
