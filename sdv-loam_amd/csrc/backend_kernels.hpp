// backend_kernels.hpp -- gfx950 kernels of the sliding-window back end.
//
//   k_ef_linearize   b1  PointFrameResidual::linearize            src/FullSystem/Residuals.cpp:60-224
//                        projectPoint (both overloads)             src/FullSystem/ResidualProjections.h:20-59
//   k_ef_apply       b1  PointFrameResidual::applyRes(true)        src/FullSystem/Residuals.cpp:252-275
//                        EFResidual::takeDataF                     src/OptimizationBackend/EnergyFunctionalStructs.cpp:15-25
//   point_body       b2  per-point part of addPoint<0>/<1>         src/OptimizationBackend/AccumulatedTopHessian.cpp:26-110
//                    b7  head of AccumulatedSCHessianSSE::addPoint src/OptimizationBackend/AccumulatedSCHessian.cpp:10-37
//   top_gram_body    b2  acc[h,t].update/updateTopRight/BotRight   AccumulatedTopHessian.cpp:68-82 (AccumulatorApprox, b3)
//   sc_gram_body     b7  accD / accE / accEB / accHcc / accbc      AccumulatedSCHessian.cpp:39-61
//   k_ef_acc_fused       the whole accumulate of a solveSystemF in one launch: per 64-point tile point_body then its sc_gram part | top_gram_body
//   k_ef_acc_stage1      top_gram_body | point_body workgroups side by side (sharded windows and > 64 * kMaxChunks points per host)
//   k_ef_sc_gram         sc_gram_body alone (same cases, where one combined reduce follows)
//   k_ef_acc_reduce      fixed-order fp64 sum of the per-workgroup partial Gram tiles into the packed accumulator buffer
//   resubstitute_body b6 EnergyFunctional::resubstituteFPt         src/OptimizationBackend/EnergyFunctional.cpp:250-282
//                        + doStepFromBackup on the per-point idepths  FullSystemOptimize.cpp:165-262   (workgroups of k_ef_tail_resub, backend_solve.inc)
//   select_th_body       FullSystem::setNewFrameEnergyTH           FullSystemOptimize.cpp:63-97 (k_ef_select_th; a workgroup of k_ef_stitch in the loop)
//   reclassify_slot      the IN / OUTLIER decision of the re-linearisation after a rejected step (k_ef_reclassify; workgroups of k_ef_tail_resub)
//   apply_slot           body of k_ef_apply and of the apply workgroups of k_ef_stats_apply (backend.hip)
//
// Data layout in HBM (one window): points are sorted by host frame; the residual of point p in target frame t
// lives in slot s = t*nP + p of every per-residual plane ("dense residual table": for a fixed (host,target) pair the
// slots are contiguous, for a fixed point they are nP apart) -- both the per-pair reductions and the per-point loops are
// coalesced.  Jacobians are kept as 24 SoA planes x 2 buffers; `sel` says which buffer the EnergyFunctional side owns
// (the reference swaps two heap pointers per residual in takeDataF; here a bit flips).
//
// The two reductions are Gram matrices and run on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, f32
// accumulate -- the same numerics class as the reference's float accumulators, in a different order):
//   top:  for a pair (h,t)   G = sum_r [Jc;Jxi;res]_x [..]_x^T + [..]_y [..]_y^T        11 x 11  (AccumulatorApprox)
//   SC:   for a host h       G = sum_p HdiF_p f_p f_p^T, f_p = [JpJdF_t(6) t=0..7 | Hcd(4) | bdSum]   53 x 53
// Operands are staged through LDS as [feature][row] tiles (row stride 68 floats: 16-byte fragment reads, 4 banks apart from row to row).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "waitflag.hpp"

namespace sdvgn {

constexpr int kMaxFrames = 8;
constexpr int kJPlanes = 24;   // resF(2) Jpdxi[0](6) Jpdxi[1](6) Jpdc[0](4) Jpdc[1](4) Jpdd(2)
constexpr int kTileStride = 68;   // floats per feature row of a Gram tile: a multiple of 4 (16-byte fragment reads), 4 banks apart from row to row

enum : int { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };
// bit kept next to state_NewState in its plane: the linearisation found wJI2_sum < 2 (Residuals.cpp:212), which k_ef_reclassify needs
// to repeat the IN / OUTLIER decision under another threshold; consumers of the state mask with RS_MASK
enum : int { RS_WJLOW = 16, RS_MASK = 3 };
enum : uint8_t { RF_EXISTS = 1, RF_MATCHER = 2, RF_LINEARIZED = 4, RF_ACTIVE = 8, RF_SEL = 16 };

#define SDVGN_SCALE_F 50.0f
#define SDVGN_SCALE_C 50.0f
#define SDVGN_SCALE_IDEPTH 1.0f

struct PrecalcDev {  // FrameFramePrecalc fields linearize reads (HessianBlocks.h:51-79), one per (host,target)
    float KRKi[9], Kt[3], R0[9], t0[3];
    float aff0, aff1, b0;
    float unused_th;      // (thresholds live in EFArrays::frameTH_r: they change after every linearizeAll)
    float dp[6];          // adHTdeltaF[h + t*nF]  (EnergyFunctional.cpp:140-141)
    int P0, np;           // point range of the host frame (8-byte aligned: read as one int2 where both are wanted at once)
    int pad[2];
};

struct EFConst {
    int nF, nP, w, h;
    float fxl, fyl, cxl, cyl, fxli, fyli;
    float wM3G, hM3G;
    float cDeltaF[4];
    float huberTH, outlierTHSumComponent;
    int debug_flags;   // profiling experiments only, read by the diagnostic instantiation k_ef_linearize<true> alone (SDVGN_DEBUG_FLAGS bit5), never by the product's: bit1 skip the J stores, bit2 disable the XCD mapping, bit7 shared gather positions,
                       // bit5 stage stamps of k_ef_linearize into EFArrays::dbg_stamps (tools/exp_linearize_stages.py)
};

struct EFArrays {
    // points (sorted by host)
    const float* pu; const float* pv; const float* pidz; const float* pid;  // u, v, idepth_zero_scaled, idepth_scaled
    const float4* pcolor;   // [nP][2]
    const float4* pweights; // [nP][2]
    const float* ppriorF; const float* pdeltaF; const uint8_t* psensor;
    // residual slots
    uint8_t* rflags; int8_t* rstate; int8_t* rstate_new;
    const float2* rmatcher;
    float* renergy; float* renergy_new; float* renergy_wo;
    const float* renergy_new_prev;   // state_NewEnergy as the previous linearisation left it (== renergy_new unless a trial set is written)
    float* rres_toZero;      // [2][slots]
    float* J;                // [2 buffers][24][slots]
    float* JpJd;             // [6][slots]
    // per point outputs
    float* pHddA; float* pbdA; float* pHcdA;  // Hcd: [4][nP]
    float* pHddL; float* pbdL; float* pHcdL;
    float* pHdi; float* pbdSum; float* pHcd;  // SC inputs (Hcd = A + L)
    float* pstep;
    uint8_t* pnogood;        // [nP] takes nogood_epoch when a solve finds the point without an active residual (AccumulatedSCHessian.cpp:14-21 zeroes PointHessian::maxRelBaseline
    uint8_t nogood_epoch;    // there); every sdvgn_ef_optimize call has its own epoch (1..255: no memset launch per call), sdvgn_ef_get_point_nogood compares
    // images
    const float* images;     // [image slot][w*h*3]; frame t's image lives in slot ef_img_slot(A, t)
    // frameEnergyTH per frame [nF]: the set k_ef_linearize classifies with / the set k_ef_select_th writes (one per state_New* set,
    // because FullSystem::setNewFrameEnergyTH moves the newest frame's threshold after EVERY linearizeAll, FullSystemOptimize.cpp:63-97,122)
    const float* frameTH_r;
    float* frameTH_w;
    // CalibHessian float views of the state the kernels work on, in device memory (written by the device-side step of the solve,
    // backend_solve.inc); NULL: the values inside EFConst (host-driven entry points) are used
    const struct CalibDev* calib;
    // diagnostics (SDVGN_DEBUG_FLAGS bit5): wall_clock64() stamps of k_ef_linearize's stages, [workgroup][wave][8]; NULL otherwise
    unsigned long long* dbg_stamps;
    // 1: PointFrameResidual::resetOOB (Residuals.h:69-76) is folded into this linearise + applyRes pair -- every residual they process counts
    // as state IN with state_energy = state_NewEnergy = 0 on entry (the first linearizeAll of FullSystem::optimize, which the reference
    // precedes with a resetOOB loop); what the pair writes is what reset + linearise + apply would have written
    int reset_oob;
    // applyRes FUSED into the linearise (EFResidual::takeDataF + the state / energy hand-over of PointFrameResidual::applyRes, Residuals.cpp:230-254,
    // for every residual the linearise processes): non-NULL = the SECOND copies of the four planes applyRes writes.  The linearise then leaves
    // in them what applyRes would have left in the first ones (flags with the Jacobian buffers swapped, state, energy, JpJd = Jpdxi^T Jpdd), reading
    // the first copies only; the caller makes the second copies the current ones if the step is accepted (a pointer swap) and does nothing if
    // it is not.  Slots the linearise does not process (no residual, or a fixed linearisation) hold the same values in both copies
    // (backend.hip, ef_sync_applied).  NULL: applyRes is a pass of its own (k_ef_apply).
    uint8_t* rflags_w; int8_t* rstate_w; float* renergy_w; float* JpJd_w;
    // sticky error word in pinned host memory (may be NULL): a workgroup that gives up waiting for a word another workgroup of the same
    // launch publishes ORs a code into it (1: the accept verdict, 2: the solution); sdvgn_ef_optimize and the solve check it and return
    // SDVGN_E_STATE instead of carrying on with a partially applied / unstepped window
    unsigned* err;
    // image slot of frame t in nibble t (identity 0x76543210 unless frames were inserted / removed in place, backend_window.inc): a
    // marginalised frame frees its slot, the frames behind it keep theirs -- no image ever moves
    unsigned img_slots;
};
__device__ __host__ __forceinline__ int ef_img_slot(unsigned img_slots, int t) { return (int)((img_slots >> (4 * t)) & 7u); }
__device__ __forceinline__ void ef_raise(unsigned* err, unsigned code) {
    if (err) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct CalibDev { float fxl, fyl, cxl, cyl, fxli, fyli; float cDeltaF[4]; float pad[2]; };
// the EFConst a kernel computes with: calibration floats from device memory when the window's state lives there
__device__ __forceinline__ EFConst ef_const(const EFConst& Cin, const EFArrays& A) {
    EFConst C = Cin;
    if (A.calib) {
        const CalibDev c = *A.calib;
        C.fxl = c.fxl; C.fyl = c.fyl; C.cxl = c.cxl; C.cyl = c.cyl; C.fxli = c.fxli; C.fyli = c.fyli;
#pragma unroll
        for (int i = 0; i < 4; ++i) C.cDeltaF[i] = c.cDeltaF[i];
    }
    return C;
}

// 64-lane double sum (for the energy), result valid in lane 63
__device__ __forceinline__ double wave_sum_double(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// b1: PointFrameResidual::linearize, two waves per 64 residuals ("role split").  One residual per lane costs ~1900 VALU
// instructions and 32 tap loads per lane, and the named size (112 000 residuals = 2000 waves) is then exactly 2 waves per SIMD.
// Here wave 2g+r of a workgroup takes ROLE r of residual group g:
//   role 0: pattern pixels 0..3, the x row of the Jacobian (resF[0], Jpdxi[0], Jpdc[0], Jpdd[0]), the residual's scalars
//   role 1: pattern pixels 4..7, the y row
// The role is wave-uniform (no divergence), every store stays a 256-B coalesced plane segment, the only exchange between the roles is
// 9 floats per residual through LDS (role 1's four energy / wJI2 terms and its alive mask) so that role 0 can replay the reference's
// sequential `+=` order over all 8 pattern pixels (bit-exact sums).
//
// The gather is TRANSPOSED through LDS.  What bounds this kernel is the L1 tag pipeline of the CU (one 128-byte line per clock): with one
// residual per lane the 64 lanes of every tap instruction hit 64-77 different lines (stage stamps + A/B runs, profiles/; with 8
// neighbouring lanes gathering at the same position the kernel drops from 19.1 to 11.9 us -- same-line requests of one instruction are
// merged).  The 8 pattern pixels of a residual land within 6 image rows and ~6 pixels of each other, so the owner lanes only PUBLISH
// where their pixels are ({element offset of the top-left tap, fx, fy} per pixel, 12 B), and the taps are fetched by 16 NEIGHBOURING
// lanes per residual -- lane = (pattern pixel of either role, tap row) -- so that one instruction asks for ~7 lines per residual instead
// of 16.  The fetching lanes also do the bilinear interpolation, in the reference's order: the row-1 lane forms w11 q11 + w01 q01, hands
// it to its row-0 neighbour (DPP), which adds w10 q10 and then w00 q00 -- ((w11 d + w01 c) + w10 b) + w00 a, globalFuncs.h:51-65 -- and
// returns {I, dx, dy} of the pixel to its owner through the same 12 bytes of LDS.  Wave (g, role) fetches for residuals
// [32 role, 32 role + 32) of group g, both roles' pixels; two workgroup barriers frame the exchange.
// grid = (ceil(max np / 128), nF*nF), block = 256 (2 groups x 2 roles).
// energy_partial[pair * gridDim.x + chunk] = sum of the return values of linearize() in this workgroup.
// ------------------------------------------------------------------------------------------------------------
struct LinLane {
    bool todo, oob, wrote;
    float e[4], wj[4];
    unsigned ok;          // bit k: pattern pixel 4*ROLE+k projected inside the image and gathered a finite intensity
    float energyLeft, e_prev;
    uint8_t fl;
};
struct LinIn {            // per-lane inputs and the pattern projections (phase A), kept for the later phases
    uint8_t fl; int st;
    float pu, pv, idz, ids;
    float4 c4, w4;
    float2 m;
    bool inb[4];
};
struct LinGeo { float res0, res1, hwm, Jr[6], Cr[4], dd; };
struct LinRec { unsigned off; float fx, fy; };          // published by the owner: top-left tap (element offset into the target image), fractions
// LDS layout of the gather exchange: per (group, role) 64 lanes x 4 pattern pixels x 12 bytes -- first the records the owner publishes, then
// {I,dx,dy} coming back in the same 12 bytes.  A lane's four records are 14 dwords apart, not 12, and role 1's array starts 16 dwords behind
// a multiple of 32: with the packed layout every access of the exchange hit busy banks (SQ_LDS_BANK_CONFLICT 588 k cycles per launch against 351 k
// SQ_ACTIVE_INST_LDS, profiles/r04_linearize_counters_exact.txt) -- the owner's 8-byte stores and phase C's 8-byte reads are serviced in
// groups of 16 lanes whose 12-dword stride puts lanes l and l + 8 on one bank (14 = 2 x 7: sixteen distinct bank pairs), and the fetching
// lanes of one instruction read both roles' records of a residual, 768 dwords apart = the same bank.
constexpr int kLinLaneDw = 14;
constexpr int kLinRoleDw = 64 * kLinLaneDw + 16;
struct __align__(16) LinSmem {
    unsigned qg[2][2][kLinRoleDw];   // [group][role][lane * kLinLaneDw + 3 * pixel + {0,1,2}]
    float xch[2][15][64];    // role 1 -> role 0: 4 energies, 4 gradient weights, the ok bits | (fused applyRes) its six terms of JpJd
    double s_e[2];
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lin_u32x2 __attribute__((ext_vector_type(2)));
// fused applyRes (EFArrays::rflags_w): JpJd = Jpdxi^T Jpdd (EnergyFunctionalStructs.cpp:20-24), one product per role and their sum, rounded
// operation by operation like k_ef_apply's -- defined here, outside the instantiations' contraction pragma, so that both arithmetic modes of
// the linearise leave what their applyRes pass left
__device__ __forceinline__ float lin_jpjd_term(float jr, float hwm, float dd) { return (jr * hwm) * (dd * hwm); }
__device__ __forceinline__ float lin_jpjd_sum(float tx, float ty) { return tx + ty; }
// ---- the kernel and its phases live in backend_linearize.inc, instantiated twice (see its header) ---------------------------------------
#define LIN_NS lin_exact
#define LIN_DIV(a, b) ((a) / (b))
#define LIN_SQRT(x) sqrtf(x)
#define LIN_FP_CONTRACT
#include "backend_linearize.inc"
#undef LIN_NS
#undef LIN_DIV
#undef LIN_SQRT
#undef LIN_FP_CONTRACT
#undef LIN_STAMP
#define LIN_NS lin_fast
#define LIN_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define LIN_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define LIN_FP_CONTRACT _Pragma("clang fp contract(fast)")
#include "backend_linearize.inc"
#undef LIN_NS
#undef LIN_DIV
#undef LIN_SQRT
#undef LIN_FP_CONTRACT
#undef LIN_STAMP
using lin_exact::k_ef_linearize;   // the name every other launch site uses

// ------------------------------------------------------------------------------------------------------------
// SURVEY 8f-4 (part 2): FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:18-185) with
// ImmaturePoint::linearizeResidual (ImmaturePoint.cpp:410-477), one lane = one immature point.  The lane runs the reference's
// loops in the reference's order (float accumulation of Hdd / bd across residuals and pattern pixels, including the partial
// contributions an out-of-bounds pattern pixel leaves behind), so result codes, inverse depths and residual states are
// bit-identical to the CPU restatement.  Per (pass, target) the 8 pattern projections are computed first and their 96 taps
// loaded as one batch.
// ------------------------------------------------------------------------------------------------------------
struct ImmPrecalc {   // FrameFramePrecalc fields linearizeResidual reads: PRE_RTll, PRE_tTll, PRE_aff_mode, per (host,target)
    float R[9], t[3], aff[2], pad[2];
};
static_assert(sizeof(ImmPrecalc) == 64, "ImmPrecalc is fetched as four 16-byte words");

struct ImmLaneRes { int state_state, state_NewState; double state_energy, state_NewEnergy; };

__device__ __forceinline__ double imm_linearize_residual(const EFConst& C, const float* __restrict__ img, const ImmPrecalc& pc_in, float pu, float pv,
                                                         const float* col, const float* wts, float energyTH, float outlierTHSlack, ImmLaneRes& tmp,
                                                         float& Hdd, float& bd, float idepth) {
    if (tmp.state_state == RS_OOB) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
    // the 16 floats of the (host,target) record in registers, all of them now: read through the reference where they are used, t[] and aff[]
    // were fetched again for every pattern pixel, behind that pixel's isfinite() test -- eight more serial round trips per residual
    ImmPrecalc pc;
    {
        const float4* src = reinterpret_cast<const float4*>(&pc_in);
        float4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = src[k];
        asm volatile("" ::"v"(r[2].y), "v"(r[2].z), "v"(r[2].w), "v"(r[3].x), "v"(r[3].y));
        pc.R[0] = r[0].x; pc.R[1] = r[0].y; pc.R[2] = r[0].z; pc.R[3] = r[0].w; pc.R[4] = r[1].x; pc.R[5] = r[1].y; pc.R[6] = r[1].z; pc.R[7] = r[1].w;
        pc.R[8] = r[2].x; pc.t[0] = r[2].y; pc.t[1] = r[2].z; pc.t[2] = r[2].w; pc.aff[0] = r[3].x; pc.aff[1] = r[3].y; pc.pad[0] = pc.pad[1] = 0;
    }
    const int pat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};
    float us[8], vs[8], dres[8], fx8[8], fy8[8], tp[8][12];
    bool ok[8];
    const float* bp8[8];
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {   // projectPoint (ResidualProjections.h:32-59)
        const float k0 = (pu + pat[idx][0] - C.cxl) * C.fxli, k1 = (pv + pat[idx][1] - C.cyl) * C.fyli;
        const float p0 = ((pc.R[0] * k0 + pc.R[1] * k1) + pc.R[2] * 1.0f) + pc.t[0] * idepth;
        const float p1 = ((pc.R[3] * k0 + pc.R[4] * k1) + pc.R[5] * 1.0f) + pc.t[1] * idepth;
        const float p2 = ((pc.R[6] * k0 + pc.R[7] * k1) + pc.R[8] * 1.0f) + pc.t[2] * idepth;
        const float drescale = 1.0f / p2;
        dres[idx] = drescale;
        bool good = drescale > 0;
        const float u = p0 * drescale, v = p1 * drescale;
        const float Ku = u * C.fxl + C.cxl, Kv = v * C.fyl + C.cyl;
        good = good && (Ku > 1.1f && Kv > 1.1f && Ku < C.wM3G && Kv < C.hM3G);
        us[idx] = u; vs[idx] = v; ok[idx] = good;
        const float x = good ? Ku : 2.0f, y = good ? Kv : 2.0f;
        const int ix = (int)x, iy = (int)y;
        fx8[idx] = x - ix; fy8[idx] = y - iy;
        bp8[idx] = img + 3 * (ix + iy * C.w);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        const float* bq = bp8[idx] + 3 * C.w;
#pragma unroll
        for (int k = 0; k < 6; ++k) { tp[idx][k] = bp8[idx][k]; tp[idx][6 + k] = bq[k]; }
    }
    // (the scheduling barriers alone do not keep the batch together: the loads of the gradient components are only used behind the
    // isfinite() test of their pixel's intensity, and the compiler sinks them there -- two loads and a wait per pixel, eight times in a row;
    // an empty asm that takes the values as inputs pins them here)
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) asm volatile("" ::"v"(tp[idx][4]), "v"(tp[idx][5]), "v"(tp[idx][10]), "v"(tp[idx][11]));
    __builtin_amdgcn_sched_barrier(0);
    float energyLeft = 0;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        if (!ok[idx]) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
        const float dx = fx8[idx], dy = fy8[idx], dxdy = dx * dy;
        const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
        const float* q = tp[idx];
        const float h0 = ((w11 * q[9] + w01 * q[6]) + w10 * q[3]) + w00 * q[0];
        const float h1 = ((w11 * q[10] + w01 * q[7]) + w10 * q[4]) + w00 * q[1];
        const float h2 = ((w11 * q[11] + w01 * q[8]) + w10 * q[5]) + w00 * q[2];
        if (!isfinite(h0)) { tmp.state_NewState = RS_OOB; return tmp.state_energy; }
        const float residual = h0 - (pc.aff[0] * col[idx] + pc.aff[1]);
        float hw = fabsf(residual) < C.huberTH ? 1.0f : C.huberTH / fabsf(residual);
        energyLeft += wts[idx] * wts[idx] * hw * residual * residual * (2 - hw);
        const float dxInterp = h1 * C.fxl, dyInterp = h2 * C.fyl;
        const float d_idepth = (dxInterp * dres[idx] * (pc.t[0] - pc.t[2] * us[idx]) + dyInterp * dres[idx] * (pc.t[1] - pc.t[2] * vs[idx])) * SDVGN_SCALE_IDEPTH;
        hw *= wts[idx] * wts[idx];
        Hdd += (hw * d_idepth) * d_idepth;
        bd += (hw * residual) * d_idepth;
    }
    if (energyLeft > energyTH * outlierTHSlack) { energyLeft = energyTH * outlierTHSlack; tmp.state_NewState = RS_OUTLIER; }
    else tmp.state_NewState = RS_IN;
    tmp.state_NewEnergy = energyLeft;
    return energyLeft;
}

__global__ void __launch_bounds__(64) k_ef_optimize_immature(EFConst C, const float* __restrict__ images, unsigned img_slots, const ImmPrecalc* __restrict__ precalc, int n,
                                                             int minObs, const int* __restrict__ host, const float* __restrict__ u, const float* __restrict__ v,
                                                             const float* __restrict__ idepth_min, const float* __restrict__ idepth_max,
                                                             const float* __restrict__ energyTH, const float4* __restrict__ color,
                                                             const float4* __restrict__ weights, const unsigned char* __restrict__ isFromSensor,
                                                             int* __restrict__ result, float* __restrict__ idepth_out, int* __restrict__ res_state) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int nF = C.nF, hst = host[i];
    const float pu = u[i], pv = v[i], eTH = energyTH[i];
    const float4 c0 = color[2 * i], c1 = color[2 * i + 1], w0 = weights[2 * i], w1 = weights[2 * i + 1];
    const float col[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float wts[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const bool fromSensor = isFromSensor[i] != 0;
    const size_t imgStride = (size_t)C.w * C.h * 3;
    ImmLaneRes res[kMaxFrames];
#pragma unroll
    for (int t = 0; t < kMaxFrames; ++t) { res[t].state_NewEnergy = res[t].state_energy = 0; res[t].state_NewState = RS_OUTLIER; res[t].state_state = RS_IN; }
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (idepth_max[i] + idepth_min[i]) * 0.5f;
    const float trueDepth = currentIdepth;
    int code = 1;
    if (!fromSensor) {
#pragma unroll
        for (int t = 0; t < kMaxFrames; ++t) {
            if (t < nF && t != hst) {
                lastEnergy = (float)((double)lastEnergy + imm_linearize_residual(C, images + ef_img_slot(img_slots, t) * imgStride, precalc[hst * nF + t], pu, pv, col, wts, eTH, 1000.f,
                                                                                 res[t], lastHdd, lastbd, currentIdepth));
                res[t].state_state = res[t].state_NewState;
                res[t].state_energy = res[t].state_NewEnergy;
            }
        }
        if (!isfinite(lastEnergy) || lastHdd < 100.f) code = 0;       // setting_minIdepthH_act (settings.cpp:41)
        float lambda = 0.1f;
        for (int iteration = 0; code == 1 && iteration < 3; iteration++) {   // setting_GNItsOnPointActivation (settings.cpp:133)
            float H = lastHdd;
            H *= 1 + lambda;
            const float step = (float)((1.0 / (double)H) * (double)lastbd);
            const float newIdepth = currentIdepth - step;
            float newHdd = 0, newbd = 0, newEnergy = 0;
#pragma unroll
            for (int t = 0; t < kMaxFrames; ++t)
                if (t < nF && t != hst)
                    newEnergy = (float)((double)newEnergy + imm_linearize_residual(C, images + ef_img_slot(img_slots, t) * imgStride, precalc[hst * nF + t], pu, pv, col, wts, eTH, 1.f,
                                                                                   res[t], newHdd, newbd, newIdepth));
            if (!isfinite(lastEnergy) || newHdd < 100.f) { code = 0; break; }
            if (newEnergy < lastEnergy) {
                currentIdepth = newIdepth;
                lastHdd = newHdd;
                lastbd = newbd;
                lastEnergy = newEnergy;
#pragma unroll
                for (int t = 0; t < kMaxFrames; ++t) { res[t].state_state = res[t].state_NewState; res[t].state_energy = res[t].state_NewEnergy; }
                lambda *= 0.5f;
            } else {
                lambda *= 5;
            }
            if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break;
        }
    }
    if (code == 1) {
        if (!isfinite(currentIdepth)) code = -1;
        else {
            int numGoodRes = 0;
#pragma unroll
            for (int t = 0; t < kMaxFrames; ++t) if (t < nF && t != hst && res[t].state_state == RS_IN) numGoodRes++;
            if (numGoodRes < minObs) code = -1;
            else if (!isfinite(eTH)) code = -1;
        }
    }
    result[i] = code;
    idepth_out[i] = code == 1 ? (fromSensor ? trueDepth : currentIdepth) : NAN;
#pragma unroll
    for (int t = 0; t < kMaxFrames; ++t)
        if (t < nF) res_state[(size_t)i * nF + t] = (t == hst) ? -1 : res[t].state_state;
}

// applyRes(true): one thread per slot.  All per-slot inputs are loaded up front (independent loads, one round trip) and the J
// planes of the buffer the flip would select right behind the flags, so that the kernel is two memory round trips deep instead
// of six; which values are used is decided afterwards.
// cond (may be NULL): the verdict of the device-side accept test; 0 = the step was rejected, nothing is applied.
// verdict (may be NULL): the same verdict as a tagged word (seq << 1 | accept) that ANOTHER workgroup of this launch publishes
// (k_ef_stats_apply): polled after this lane's loads are in flight.
// bak (may be NULL): what this slot's applyRes overwrites -- flags, state, energy, JpJd -- is saved there first, so that a SPECULATIVE
// apply (the sharded loop applies the trial linearisation before the all-reduce that carries the energy of the accept test, backend.hip)
// can be taken back by k_ef_apply_revert; bak.fl[s] = 0xFF marks a slot the apply did not touch.
struct ApplyBackup { uint8_t* fl; int8_t* st; float* en; float* JpJd; };
__device__ __forceinline__ void apply_slot(int nF, int nP, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                           const int* __restrict__ phost, size_t s, const int* __restrict__ cond,
                                           const unsigned* verdict = nullptr, unsigned seq = 0, const ApplyBackup* bak = nullptr) {
    const size_t slots = (size_t)nF * nP;
    if (s >= slots) return;
    if (bak) bak->fl[s] = 0xFF;
    // every input of the slot in two batches of unconditional loads BEFORE the verdict is waited for: the slot's own planes, then (they depend
    // on its flags / host) the freshly linearised Jacobian values and the host's shard flag -- behind the verdict only arithmetic and stores
    // are left (a load under a condition gets a branch and a wait of its own)
    const int go_in = cond ? *cond : 1;
    const int hh = phost[s % nP];
    uint8_t fl = A.rflags[s];
    const int st_in = (int)A.rstate[s];
    const int sn = A.rstate_new[s];
    const float en = A.renergy_new[s];
    __builtin_amdgcn_sched_barrier(0);
    int go = go_in;
    const int st = A.reset_oob ? (int)RS_IN : st_in;
    const int np_h = precalc[hh * nF + hh].np;
    const int buf = (fl & RF_SEL) ? 0 : 1;              // the buffer the residual's freshly linearised J sits in
    const float* Je = A.J + (size_t)buf * kJPlanes * slots + s;
    float jx[6], jy[6];
    const float d0 = Je[22 * slots], d1 = Je[23 * slots];
#pragma unroll
    for (int i = 0; i < 6; ++i) { jx[i] = Je[(2 + i) * slots]; jy[i] = Je[(8 + i) * slots]; }
    __builtin_amdgcn_sched_barrier(0);
    if (verdict) {
        unsigned w = 0;
        int polls = 0;
        for (;;) {
            w = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 1) == seq || ++polls > (1 << 18)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        go = ((w >> 1) == seq) ? (int)(w & 1u) : 0;      // a verdict that did not arrive in time applies nothing here -- and is an error of the
        if ((w >> 1) != seq) ef_raise(A.err, 1u);        // call: the publisher may only be late, other slots may have applied (A.err, sticky)
    }
    if (!go) return;
    if (np_h == 0) return;   // host frame not in this rank's shard
    if (!(fl & RF_EXISTS) || (fl & RF_LINEARIZED)) return;
    if (st == RS_OOB) {          // applyRes leaves an OOB residual alone (:applyRes) -- except for the buffer swap, see below
        if (bak) { bak->fl[s] = fl; bak->st[s] = A.rstate[s]; bak->en[s] = A.renergy[s]; }
        A.rflags[s] = fl ^ RF_SEL;
        return;
    }
    if (bak) {
        bak->fl[s] = fl; bak->st[s] = A.rstate[s]; bak->en[s] = A.renergy[s];
        if ((sn & RS_MASK) == RS_IN) {
#pragma unroll
            for (int i = 0; i < 6; ++i) bak->JpJd[(size_t)i * slots + s] = A.JpJd[(size_t)i * slots + s];
        }
    }
    if ((sn & RS_MASK) == RS_IN) {
        fl |= RF_ACTIVE;                                // takeDataF: swap J with the residual's freshly linearised J (the flip below)
#pragma unroll
        for (int i = 0; i < 6; ++i) A.JpJd[(size_t)i * slots + s] = jx[i] * d0 + jy[i] * d1;
    } else {
        fl &= (uint8_t)~RF_ACTIVE;
    }
    // The buffers swap for EVERY residual that exists and is not fixed (new state IN, OUTLIER or OOB), not only for the ones that stay active:
    // the Jacobian of an inactive residual is never read (it becomes active again only through a later linearise + this swap), and all these
    // residuals then keep the same RF_SEL for the life of the window -- a wave's 64 stores of a Jacobian plane (and the accumulate's loads)
    // go to ONE buffer instead of being split lane by lane between the two after a few accepted steps.
    fl ^= RF_SEL;
    A.rflags[s] = fl;
    A.rstate[s] = (int8_t)(sn & RS_MASK);
    A.renergy[s] = en;
}
__global__ void __launch_bounds__(256) k_ef_apply(int nF, int nP, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                  const int* __restrict__ phost, const int* __restrict__ cond) {
    apply_slot(nF, nP, A, precalc, phost, (size_t)blockIdx.x * blockDim.x + threadIdx.x, cond);
}
__global__ void __launch_bounds__(256) k_ef_apply_backup(int nF, int nP, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                         const int* __restrict__ phost, ApplyBackup bak) {
    apply_slot(nF, nP, A, precalc, phost, (size_t)blockIdx.x * blockDim.x + threadIdx.x, nullptr, nullptr, 0, &bak);
}
// takes a speculative applyRes back (the step was rejected): every touched slot gets its flags, state, energy and -- if the apply had
// swapped the Jacobian buffers -- its JpJd back
__global__ void __launch_bounds__(256) k_ef_apply_revert(int nF, int nP, EFArrays A, ApplyBackup bak) {
    const size_t slots = (size_t)nF * nP, s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    const uint8_t old = bak.fl[s];
    if (old == 0xFF) return;
    const uint8_t cur = A.rflags[s];
    if ((cur & RF_ACTIVE) && ((cur ^ old) & RF_SEL)) {     // the apply made it active with the new Jacobian: JpJd was rewritten (and saved)
#pragma unroll
        for (int i = 0; i < 6; ++i) A.JpJd[(size_t)i * slots + s] = bak.JpJd[(size_t)i * slots + s];
    }
    A.rflags[s] = old; A.rstate[s] = bak.st[s]; A.renergy[s] = bak.en[s];
}

// Per-point sums of addPoint<0> (active, not linearised) and addPoint<1> (active, linearised), then the head of the
// Schur accumulation: HdiF, bdSumF, Hcd.  Workgroup = 64 points x 8 waves; wave t reads the residual slot of target
// frame t (coalesced: consecutive lanes = consecutive points), lanes of wave 0 add the 8 targets in ascending order.
struct PointSmem { float part[kMaxFrames][13][64]; };
struct PointOut { int p; bool nogood; float HddA, bdA, HddL, bdL, HcdA[4], HcdL[4], hdi, bds, Hcd[4]; };
__device__ __forceinline__ void point_store(const EFArrays& A, int nP, const PointOut& o) {
    const int p = o.p;
    A.pHddA[p] = o.HddA; A.pbdA[p] = o.bdA; A.pHddL[p] = o.HddL; A.pbdL[p] = o.bdL;
#pragma unroll
    for (int i = 0; i < 4; ++i) { A.pHcdA[(size_t)i * nP + p] = o.HcdA[i]; A.pHcdL[(size_t)i * nP + p] = o.HcdL[i]; }
    if (o.nogood && A.pnogood) A.pnogood[p] = A.nogood_epoch;
    A.pHdi[p] = o.hdi; A.pbdSum[p] = o.bds;
#pragma unroll
    for (int i = 0; i < 4; ++i) A.pHcd[(size_t)i * nP + p] = o.Hcd[i];
}

// body for one workgroup of 256 threads = the points [p_base, p_base + p_count), p_count <= 64; wave w handles targets w and w + 4.
// MODE 0: solveSystemF (addPoint<0> and <1> + accumulateSCF head, shiftPriorToZero = true).
// MODE 2: marginalizePointsF (EnergyFunctional.cpp:514-549) for the points with mask[p] != 0: priorF *= setting_idepthFixPriorMargFac,
//         addPoint<2> (all active residuals, resApprox = res_toZeroF; sums go to the L fields, the A fields are zeroed,
//         AccumulatedTopHessian.cpp:61-62,99-110) + SC head with shiftPriorToZero = false; other points are left untouched.
template <int MODE = 0>
__device__ __forceinline__ void point_body(const EFConst& C, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                           const int* __restrict__ phost, int p_base, int p_count, PointSmem& S,
                                           const uint8_t* __restrict__ mask = nullptr, float* __restrict__ prior_w = nullptr,
                                           float (*pt_out)[64] = nullptr /* [6][64] LDS: Hcd[4], bdSum, Schur weight of the tile's points */,
                                           int h_tile = -1 /* >= 0: every point of the tile is hosted by this key-frame of the rank's shard: no phost / table loads ahead of the flags */,
                                           const unsigned* verdict = nullptr /* non-NULL: the accept test of the step these sums belong to is being taken by another workgroup of
                                                                               THIS launch (k_ef_acc_stats): the per-point planes are written only if it says accept */,
                                           unsigned verdict_seq = 0) {
    float (*part)[13][64] = S.part;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (pt_out && wave == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) pt_out[i][lane] = 0.0f;
    }
    const int p = p_base + lane;
    const size_t slots = (size_t)C.nF * C.nP;
    int h = 0;
    bool mine = false;
    if (lane < p_count) {
        if (h_tile >= 0) { h = h_tile; mine = true; }
        else { h = phost[p]; mine = precalc[h * C.nF + h].np != 0; }   // host frame in this rank's shard
        if (MODE == 2) mine = mine && mask[p] != 0;
    }
    // this wave's two targets (wave, wave + 4): the flags of both in one round trip, then every Jacobian value of both in one batch of
    // independent loads (the per-target form was flag -> values -> flag -> values: four dependent round trips per workgroup)
    static_assert(kMaxFrames == 8, "two targets per wave");
    uint8_t fl2[2] = {0, 0};
    size_t s2[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int t = wave + 4 * k;
        const bool ex = mine && t < C.nF;
        s2[k] = ex ? (size_t)t * C.nP + p : 0;
        const uint8_t fr = A.rflags[s2[k]];
        fl2[k] = ex ? fr : (uint8_t)0;
    }
    // the per-point inputs of the Schur head below (wave 0 uses them behind the workgroup barrier): in flight with everything else
    const int pq = (wave == 0 && mine) ? p : p_base;
    const float prior_in = A.ppriorF[pq], delta_in = A.pdeltaF[pq];
    const uint8_t sensor_in = A.psensor[pq];
    __builtin_amdgcn_sched_barrier(0);
    // (unconditional loads -- slot s2 = 0 exists when the residual does not --: a load under `on ? ... : 0` is sunk into a branch of its own
    // and the twelve values of a target arrive one round trip after the other)
    float jd[2][2], jc0[2][4], jc1[2][4], jr[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* Je = A.J + (size_t)((fl2[k] & RF_SEL) ? 1 : 0) * kJPlanes * slots + s2[k];
        jd[k][0] = Je[22 * slots]; jd[k][1] = Je[23 * slots];
#pragma unroll
        for (int i = 0; i < 4; ++i) { jc0[k][i] = Je[(14 + i) * slots]; jc1[k][i] = Je[(18 + i) * slots]; }
        jr[k][0] = Je[0]; jr[k][1] = Je[slots];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
    const int t = wave + 4 * k;
    float v[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) v[i] = 0.0f;   // bdA HddA HcdA[4] bdL HddL HcdL[4] ngood
    if (mine && t < C.nF) {
        const size_t s = s2[k];
        const uint8_t fl = fl2[k];
        if ((fl & RF_EXISTS) && (fl & RF_ACTIVE)) {
            v[12] = 1.0f;
            const float* Je = A.J + (size_t)((fl & RF_SEL) ? 1 : 0) * kJPlanes * slots + s;
            const float d0 = jd[k][0], d1 = jd[k][1];
            float c0[4], c1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[i] = jc0[k][i]; c1[i] = jc1[k][i]; }
            if (MODE == 2) {
                const float r0 = A.rres_toZero[s], r1 = A.rres_toZero[slots + s];
                v[6] = r0 * d0 + r1 * d1;
                v[7] = d0 * d0 + d1 * d1;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[8 + i] = c0[i] * d0 + c1[i] * d1;
            } else if (!(fl & RF_LINEARIZED)) {
                const float r0 = jr[k][0], r1 = jr[k][1];
                v[0] = r0 * d0 + r1 * d1;
                v[1] = d0 * d0 + d1 * d1;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[2 + i] = c0[i] * d0 + c1[i] * d1;
            } else {
                const PrecalcDev& pc = precalc[h * C.nF + t];
                const float dd = A.pdeltaF[p];
                float dx = 0, dy = 0, cx = 0, cy = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i) { dx += Je[(2 + i) * slots] * pc.dp[i]; dy += Je[(8 + i) * slots] * pc.dp[i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { cx += c0[i] * C.cDeltaF[i]; cy += c1[i] * C.cDeltaF[i]; }
                const float r0 = A.rres_toZero[s] + (dx + cx + d0 * dd);
                const float r1 = A.rres_toZero[slots + s] + (dy + cy + d1 * dd);
                v[6] = r0 * d0 + r1 * d1;
                v[7] = d0 * d0 + d1 * d1;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[8 + i] = c0[i] * d0 + c1[i] * d1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) part[t][i][lane] = v[i];
    }
    __syncthreads();
    if (wave != 0 || !mine) return;
    float sum[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) sum[i] = 0.0f;
    for (int tt = 0; tt < C.nF; ++tt)
#pragma unroll
        for (int i = 0; i < 13; ++i) sum[i] += part[tt][i][lane];
    // the point's outputs (AccumulatedSCHessian.cpp:12-34 for the last six)
    PointOut o;
    o.p = p;
    o.HddA = sum[1]; o.bdA = sum[0]; o.HddL = sum[7]; o.bdL = sum[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.HcdA[i] = sum[2 + i]; o.HcdL[i] = sum[8 + i]; }
    o.nogood = sum[12] == 0.0f;
    float hdi = 0.0f, bds = 0.0f;
    if (!o.nogood) {
        float prior = prior_in;
        if (MODE == 2) { prior *= 600.0f * 600.0f; prior_w[p] = prior; }   // setting_idepthFixPriorMargFac, EnergyFunctional.cpp:527
        float H = o.HddA + o.HddL + prior;
        if (H < 1e-10) H = 1e-10;
        hdi = (float)(1.0 / H);
        bds = o.bdA + o.bdL;
        if (MODE != 2) bds += prior * delta_in;  // shiftPriorToZero == true in accumulateSCF_MT, false in marginalizePointsF
    }
    o.hdi = hdi; o.bds = bds;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.Hcd[i] = o.nogood ? 0.0f : sum[2 + i] + sum[8 + i];
    bool go = true;
    if (verdict) {   // (everything above ran on the accepted case's inputs; a rejected step must find the planes of the kept state untouched.  Storing later -- behind the
        // workgroup's Gram phase -- was measured: the launch 9.3 -> 10.0 us, the stores then trail the workgroup; by here the word has normally arrived)
        unsigned w = 0;
        int polls = 0;
        for (;;) {
            w = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 1) == verdict_seq || ++polls > (1 << 18)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if ((w >> 1) != verdict_seq) ef_raise(A.err, 1u);
        go = (w >> 1) == verdict_seq && (w & 1u);
    }
    if (go) point_store(A, C.nP, o);
    if (pt_out && !o.nogood) {   // the fused accumulate's Schur Gram takes these from LDS (weight 0 for LiDAR points, AccumulatedSCHessian.cpp:36-37)
#pragma unroll
        for (int i = 0; i < 4; ++i) pt_out[i][lane] = o.Hcd[i];
        pt_out[4][lane] = bds;
        pt_out[5][lane] = sensor_in ? 0.0f : hdi;
    }
}

// ------------------------------------------------------------------------------------------------------------
// MFMA Gram accumulation.  One wave owns 64 rows (K) of a [NT*16 features][64 rows] LDS tile and accumulates the upper
// triangle of tiles of  G += (w .* F) F^T  with v_mfma_f32_16x16x4_f32.
// Fragment maps (cdna guide section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg.
// ------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MFMA round (q, j) takes the rows {4q + j, 16 + 4q + j, 32 + 4q + j, 48 + 4q + j} of the tile: lane (f, kq) then reads four CONSECUTIVE rows
// 16 kq + 4q .. + 3 of its feature with one 16-byte LDS load per feature tile, issued ahead of the four rounds that use them.
__device__ __forceinline__ f32x4 lds_frag4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <int NT>
__device__ __forceinline__ void gram_tile_accumulate(const float* __restrict__ tileF /*16-byte aligned*/, const float* __restrict__ tileW,
                                                     f32x4* acc) {
    const int lane = threadIdx.x & 63;
    const int f = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 frag[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) frag[ti] = lds_frag4(tileF + (ti * 16 + f) * kTileStride + 16 * kq + 4 * q);
        const f32x4 w4 = tileW ? lds_frag4(tileW + 16 * kq + 4 * q) : (f32x4){1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj) {
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ti][j] * w4[j], frag[tj][j], acc[a], 0, 0, 0);
                    ++a;
                }
        }
    }
}

// top Gram: grid = (chunks, nF*nF), block = 256 (4 waves x 64 residual slots).  Features (16, 11 live):
// 0-3 Jpdc, 4-9 Jpdxi, 10 res; two row sets (x and y).  partial: [pair][chunk][256] floats (row-major 16x16).
struct TopGramSmem { alignas(16) float tile[4][2][16 * kTileStride]; float red[4][256]; int s_n[4]; };

// body for workgroup (bx of gx chunks, pair).  MODE 0: addPoint<0>; MODE 2: addPoint<2> over the points with mask[p] != 0 (all
// active residuals, residual feature = res_toZeroF).
template <int MODE = 0>
__device__ __forceinline__ void top_gram_body(const EFConst& C, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                              float* __restrict__ partial, int* __restrict__ nres_partial, int bx, int pair, int gx,
                                              TopGramSmem& S, const uint8_t* __restrict__ mask = nullptr,
                                              const PrecalcDev* __restrict__ ranges = nullptr /* a table whose P0 / np are known without waiting (k_ef_acc_fused) */) {
    float (*tile)[2][16 * kTileStride] = S.tile;
    float (*red)[256] = S.red;
    int* s_n = S.s_n;
    const int h = pair / C.nF, t = pair % C.nF;
    const PrecalcDev& pc = (ranges ? ranges : precalc)[pair];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int P0 = pc.P0, np = pc.np;
    const size_t slots = (size_t)C.nF * C.nP;
    f32x4 acc[1] = {{0, 0, 0, 0}};
    int cnt = 0;
    if (h != t) {
        for (int base = bx * 256 + wave * 64; base < np; base += gx * 256) {
            const int pl = base + lane;
            bool use = false;
            // the slot's flag, then its 22 values in ONE batch of unconditional loads (slot 0 stands in for a lane beyond the host's points): a
            // load written `use ? Je[..] : 0` is sunk under a branch of its own and the values arrive in five or six round trips instead of one
            const bool inr = pl < np;
            const size_t s = inr ? (size_t)t * C.nP + (P0 + pl) : 0;
            const uint8_t fl = A.rflags[s];
            uint8_t mk = 1;
            if (MODE == 2) mk = mask[inr ? P0 + pl : 0];
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) use = inr && (fl & RF_EXISTS) && (fl & RF_ACTIVE) && mk != 0;
            else use = inr && (fl & RF_EXISTS) && (fl & RF_ACTIVE) && !(fl & RF_LINEARIZED);
            const float* Je = A.J + (size_t)((fl & RF_SEL) ? 1 : 0) * kJPlanes * slots + s;
            float vx[11], vy[11];
#pragma unroll
            for (int i = 0; i < 4; ++i) { vx[i] = Je[(14 + i) * slots]; vy[i] = Je[(18 + i) * slots]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) { vx[4 + i] = Je[(2 + i) * slots]; vy[4 + i] = Je[(8 + i) * slots]; }
            if (MODE == 2) { vx[10] = A.rres_toZero[s]; vy[10] = A.rres_toZero[slots + s]; }
            else { vx[10] = Je[0]; vy[10] = Je[slots]; }
            __builtin_amdgcn_sched_barrier(0);
            float* tx = tile[wave][0];
            float* ty = tile[wave][1];
            // feature f of row `lane`
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                tx[i * kTileStride + lane] = use ? vx[i] : 0.0f;
                ty[i * kTileStride + lane] = use ? vy[i] : 0.0f;
            }
#pragma unroll
            for (int i = 11; i < 16; ++i) { tx[i * kTileStride + lane] = 0.0f; ty[i * kTileStride + lane] = 0.0f; }
            cnt += use ? 1 : 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed (wave-private tile)
            __builtin_amdgcn_wave_barrier();
            gram_tile_accumulate<1>(tx, nullptr, acc);
            gram_tile_accumulate<1>(ty, nullptr, acc);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // D fragment -> LDS (row = 4*(lane>>4)+reg, col = lane&15), then 256 threads add the 4 waves
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[0][r];
    int wc = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wc += __shfl_xor(wc, off);
    if (lane == 0) s_n[wave] = wc;
    __syncthreads();
    const size_t o = ((size_t)pair * gx + bx);
    partial[o * 256 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 0) nres_partial[o] = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
}

// Stage 1 of the split accumulate: the top-Gram workgroups and the per-point (Hdd, bd, Hcd, HdiF) workgroups are independent, so
// they share one launch (workgroups [0, n_top) = top Gram as (chunk, pair), the rest = 64 points each) and run side by side.
__global__ void __launch_bounds__(256) k_ef_acc_stage1(EFConst Cin, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                       const int* __restrict__ phost, float* __restrict__ top_partial,
                                                       int* __restrict__ nres_partial, int top_chunks, int n_top) {
    const EFConst C = ef_const(Cin, A);
    __shared__ union U { TopGramSmem t; PointSmem p; __device__ U() {} } S;
    const int b = blockIdx.x;
    if (b < n_top) top_gram_body(C, A, precalc, top_partial, nres_partial, b % top_chunks, b / top_chunks, top_chunks, S.t);
    else point_body(C, A, precalc, phost, (b - n_top) * 64, min(64, C.nP - (b - n_top) * 64), S.p);
}

// SC Gram: grid = (chunks, nF hosts), block = 256.  Features (64, 53 live): 6*t+i = JpJdF of the residual in target t
// (0 if absent/inactive); 48-51 = Hcd; 52 = bdSumF.  Row weight = HdiF (0 for LiDAR points -> excluded from the Schur
// complement, AccumulatedSCHessian.cpp:36-37).  The workgroup walks its points in tiles of 64: all 256 lanes stage the
// [64 feat][64 pts] tile (lane = point, wave = 16-feature group; next tile's values are fetched into registers while the
// current one is multiplied), then each wave accumulates the 2-3 output tiles it owns (10 upper 16x16 tiles over 4
// waves), so no cross-wave reduction is needed.  partial: [host][chunk][10 tiles][256] floats.
template <int A> struct ScTile {   // a-th upper tile of the 4x4 tile grid, row-major: (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
    static constexpr int ti = A < 4 ? 0 : (A < 7 ? 1 : (A < 9 ? 2 : 3));
    static constexpr int tj = A < 4 ? A : (A < 7 ? A - 3 : (A < 9 ? A - 5 : 3));
};

// The Schur-Gram step of one wave on a staged [64 features][64 points] tile: the 2-3 output tiles the wave owns (ScTile), rows taken in the
// order of gram_tile_accumulate (16-byte fragment reads, four MFMA rounds per read).
template <int WAVE>
__device__ __forceinline__ void sc_tile_accumulate(const float* tile, const float* wrow, f32x4* acc) {
    constexpr int NQ = (WAVE + 8 < 10) ? 3 : 2;
    const int lane = threadIdx.x & 63;
    const int f = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 frag[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) frag[ti] = lds_frag4(tile + (ti * 16 + f) * kTileStride + 16 * kq + 4 * q);
        const f32x4 w4 = lds_frag4(wrow + 16 * kq + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ScTile<WAVE>::ti][j] * w4[j], frag[ScTile<WAVE>::tj][j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ScTile<WAVE + 4>::ti][j] * w4[j], frag[ScTile<WAVE + 4>::tj][j], acc[1], 0, 0, 0);
            if (NQ == 3) acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[ScTile<(WAVE + 8) % 10>::ti][j] * w4[j], frag[ScTile<(WAVE + 8) % 10>::tj][j], acc[2], 0, 0, 0);
        }
    }
}

// body of k_ef_sc_gram for one wave; WAVE is a compile-time constant so that feature / tile indices fold
template <int WAVE, int MODE = 0>
__device__ __forceinline__ void sc_gram_wave(const EFConst& C, const EFArrays& A, int P0, int begin, int end, float* tile, float* wrow,
                                             float* __restrict__ out, const uint8_t* __restrict__ mask = nullptr) {
    const int lane = threadIdx.x & 63;
    const size_t slots = (size_t)C.nF * C.nP;
    constexpr int NQ = (WAVE + 8 < 10) ? 3 : 2;   // tiles WAVE, WAVE+4, WAVE+8 (< 10)
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (f32x4){0, 0, 0, 0};
    float stage[16];
    float stage_w = 0.0f;
    auto fetch = [&](int base) {   // features WAVE*16 .. WAVE*16+15 of point base+lane
        const int pl = base + lane;
        bool in = pl < end;
        const int p = P0 + (in ? pl : 0);
        // MODE 2: a point outside the mask has weight 0, and its Hcd / bdSum planes may never have been written (0 * NaN = NaN on the
        // matrix cores): its whole feature row is zero
        if (MODE == 2) in = in && mask[p] != 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            constexpr int f0 = WAVE * 16;
            const int f = f0 + j;
            float v = 0.0f;
            if (f < 48) {
                const int t = f / 6, i = f - 6 * t;
                const bool ok = in && t < C.nF;
                const size_t s = ok ? (size_t)t * C.nP + p : 0;   // flag and value loaded side by side (one round trip), selected afterwards
                const uint8_t fl = A.rflags[s];
                const float jv = A.JpJd[(size_t)i * slots + s];
                if (ok && (fl & RF_EXISTS) && (fl & RF_ACTIVE)) v = jv;
            } else if (f < 52) {
                if (in) v = A.pHcd[(size_t)(f - 48) * C.nP + p];
            } else if (f == 52) {
                if (in) v = A.pbdSum[p];
            }
            stage[j] = v;
        }
        if (WAVE == 3) stage_w = (in && !A.psensor[p] && (MODE != 2 || mask[p] != 0)) ? A.pHdi[p] : 0.0f;
    };
    if (begin < end) fetch(begin);
    for (int base = begin; base < end; base += 64) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[(WAVE * 16 + j) * kTileStride + lane] = stage[j];
        if (WAVE == 3) wrow[lane] = stage_w;
        __syncthreads();
        if (base + 64 < end) fetch(base + 64);   // next tile's loads fly while this one is multiplied
        sc_tile_accumulate<WAVE>(tile, wrow, acc);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int a = WAVE + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[a * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[q][r];
    }
}

struct ScGramSmem { alignas(16) float tile[64 * kTileStride]; alignas(16) float wrow[64]; };

// Single-tile Schur Gram of the fused accumulate (k_ef_acc_fused): the JpJdF / flag loads of the tile are issued BEFORE the per-point
// phase -- they do not depend on it -- and Hcd / bdSum / weight arrive through LDS instead of a store -> load round trip through memory.
template <int WAVE>
__device__ __forceinline__ void sc_fused_prefetch(const EFConst& C, const EFArrays& A, int P0, int begin, int end, float* stage) {
    const int lane = threadIdx.x & 63;
    const size_t slots = (size_t)C.nF * C.nP;
    const int pl = begin + lane;
    const bool in = pl < end;
    const int p = P0 + (in ? pl : 0);
    // flag and value of every feature loaded side by side in ONE batch (clamped addresses: slot 0 always exists), selected behind the
    // scheduling barrier -- written as `if (ok && flags) v = load` the compiler sinks each load under its branch: 16 serial round trips,
    // 5.9 us of the workgroup's 10.7 (stamps, profiles/r03_notes.txt)
    uint8_t fl[16];
    float jv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        constexpr int f0 = WAVE * 16;
        const int f = f0 + j;
        fl[j] = 0; jv[j] = 0.0f;
        if (f < 48) {   // (compile-time)
            const int t = f / 6, i = f - 6 * t;
            const bool ok = in && t < C.nF;
            const size_t s = ok ? (size_t)t * C.nP + p : 0;
            fl[j] = A.rflags[s];
            jv[j] = A.JpJd[(size_t)i * slots + s];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        constexpr int f0 = WAVE * 16;
        const int f = f0 + j;
        const int t = f / 6;
        const bool ok = f < 48 && in && t < C.nF;
        stage[j] = (ok && (fl[j] & RF_EXISTS) && (fl[j] & RF_ACTIVE)) ? jv[j] : 0.0f;
    }
}
template <int WAVE>
__device__ __forceinline__ void sc_fused_finish(float* stage, const float (*pt)[64], float* tile, float* wrow, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        constexpr int f0 = WAVE * 16;
        const int f = f0 + j;
        if (f >= 48 && f < 52) stage[j] = pt[f - 48][lane];
        else if (f == 52) stage[j] = pt[4][lane];
        tile[(WAVE * 16 + j) * kTileStride + lane] = stage[j];
    }
    if (WAVE == 3) wrow[lane] = pt[5][lane];
    __syncthreads();
    constexpr int NQ = (WAVE + 8 < 10) ? 3 : 2;   // tiles WAVE, WAVE+4, WAVE+8 (< 10)
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (f32x4){0, 0, 0, 0};
    sc_tile_accumulate<WAVE>(tile, wrow, acc);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int a = WAVE + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[a * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[q][r];
    }
}



template <int MODE = 0>
__device__ __forceinline__ void sc_gram_body(const EFConst& C, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                             float* __restrict__ partial, int pts_per_block, int bx, int h, int gx, ScGramSmem& S,
                                             const uint8_t* __restrict__ mask = nullptr) {
    const int wave = threadIdx.x >> 6;
    const int P0 = precalc[h * C.nF + h].P0, np = precalc[h * C.nF + h].np;   // np == 0 outside this rank's shard
    const int begin = bx * pts_per_block, end = min(np, begin + pts_per_block);
    float* out = partial + ((size_t)h * gx + bx) * 10 * 256;
    switch (wave) {   // wave-uniform: every wave runs straight-line code specialised for the tiles / features it owns
        case 0: sc_gram_wave<0, MODE>(C, A, P0, begin, end, S.tile, S.wrow, out, mask); break;
        case 1: sc_gram_wave<1, MODE>(C, A, P0, begin, end, S.tile, S.wrow, out, mask); break;
        case 2: sc_gram_wave<2, MODE>(C, A, P0, begin, end, S.tile, S.wrow, out, mask); break;
        default: sc_gram_wave<3, MODE>(C, A, P0, begin, end, S.tile, S.wrow, out, mask); break;
    }
}

__global__ void __launch_bounds__(256) k_ef_sc_gram(EFConst C, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                    float* __restrict__ partial, int pts_per_block) {
    __shared__ ScGramSmem S;
    sc_gram_body(C, A, precalc, partial, pts_per_block, blockIdx.x, blockIdx.y, gridDim.x, S);
}

// The whole accumulate of solveSystemF in ONE launch (windows with <= 64 * kMaxChunks points per host frame): workgroups [0, n_sc) take
// one 64-point tile of one host frame each -- first the per-point sums of that tile (point_body), then, behind a workgroup barrier, its
// Schur Gram (its JpJdF loads already in flight before the per-point phase; Hcd / bdSum / weight through LDS) -- and the remaining
// workgroups the top Grams.  The
// two kinds are independent of each other, so the per-point launch + its kernel boundary leave the critical path of the loop body.
// alt (optimize loop): this accumulate was queued BEHIND the statistics + accept-test launch of a trial step, before the host knew the
// verdict, so that no launch latency separates the two.  The arguments describe the window after an accepted step; when the verdict word
// says "rejected", the state-dependent inputs -- the point copies, the calib floats, the precalc table -- are the kept ones instead.
// skip_on_reject: the solution of the rejected case was computed ahead on the side stream (ef_launch_spec_solve, backend.hip), so a "rejected"
// verdict means this accumulate has no reader -- the planes it would rewrite already hold the kept state's values: every workgroup returns at once
struct AccAlt { const int* verdict; const float* pid; const float* pidz; const float* pdeltaF; const CalibDev* calib; const PrecalcDev* precalc; int skip_on_reject;
                // (fused applyRes: the planes applyRes writes also exist twice; NULL = they do not depend on the verdict)
                uint8_t* rflags; int8_t* rstate; float* renergy; float* JpJd;
                // (k_ef_acc_stats: the accept test is a workgroup of the SAME launch -- the accumulate runs on the accepted case's arguments without looking at
                // `verdict`, its only writes outside scratch, the per-point planes, wait for this tagged word; requires skip_on_reject)
                const unsigned* verdict_word; unsigned verdict_seq; };
// the accumulate's LDS: one union for the three kinds of workgroup -- and for the statistics workgroup k_ef_acc_stats adds (a separate 8 kB array took the
// kernel from 4 to 3 workgroups per CU: 768 places for its 769 workgroups, the statistics workgroup -- the LAST one -- waited for a place)
union AccSmem { TopGramSmem t; PointSmem p; ScGramSmem s; double stats[4][256]; __device__ AccSmem() {} };
// (body for workgroup b of the launch: k_ef_acc_fused launches it for one window, k_lock_acc -- backend_lockstep.inc -- for B windows in one grid)
__device__ __forceinline__ void acc_fused_body(AccSmem& S, const PrecalcDev* __restrict__ precalc, const EFConst& Cin, EFArrays A,
                                               const int* __restrict__ phost, float* __restrict__ top_partial,
                                               int* __restrict__ nres_partial, int top_chunks, float* __restrict__ sc_partial,
                                               int sc_chunks, int n_sc, const AccAlt& alt, const int b) {
    const PrecalcDev* __restrict__ ranges = precalc;   // point ranges / shard flags are the same in both tables: read them without waiting for the verdict
    // the verdict word is FETCHED here and LOOKED AT where the state-dependent inputs are first needed: tested at once it heads the chain
    // verdict -> point range -> flags -> values of every Schur workgroup with a round trip of its own
    const int vd = (alt.verdict && !alt.verdict_word) ? *alt.verdict : 1;
    if (alt.skip_on_reject && vd == 0) return;   // (uniform over the grid: before any barrier)
    // (fused applyRes: which copy of the flags / JpJd planes is current depends on the verdict; the Schur workgroups' first loads of them wait for
    // the point range anyway, which was fetched together with the verdict)
    if (vd == 0 && alt.rflags) { A.rflags = alt.rflags; A.rstate = alt.rstate; A.renergy = alt.renergy; A.JpJd = alt.JpJd; }
    if (b < n_sc) {
        const int h = b / sc_chunks, bx = b - h * sc_chunks;
        const int2 rg = *reinterpret_cast<const int2*>(&ranges[h * Cin.nF + h].P0);   // {P0, np} in one load; np == 0 outside this rank's shard
        const int P0 = rg.x, np = rg.y;
        const int begin = bx * 64, end = min(np, begin + 64);
        __shared__ float pt[6][64];
        const int wave = threadIdx.x >> 6;
        float stage[16];
        switch (wave) {   // wave-uniform
            case 0: sc_fused_prefetch<0>(Cin, A, P0, begin, end, stage); break;
            case 1: sc_fused_prefetch<1>(Cin, A, P0, begin, end, stage); break;
            case 2: sc_fused_prefetch<2>(Cin, A, P0, begin, end, stage); break;
            default: sc_fused_prefetch<3>(Cin, A, P0, begin, end, stage); break;
        }
        if (vd == 0) { A.pid = alt.pid; A.pidz = alt.pidz; A.pdeltaF = alt.pdeltaF; A.calib = alt.calib; precalc = alt.precalc; }
        const EFConst C = ef_const(Cin, A);
        if (begin < np) point_body(C, A, precalc, phost, P0 + begin, end - begin, S.p, nullptr, nullptr, pt, h, alt.verdict_word, alt.verdict_seq);   // (np != 0: the host is this rank's)
        else if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) pt[i][threadIdx.x & 63] = 0.0f;
        }
        __syncthreads();
        float* out = sc_partial + ((size_t)h * sc_chunks + bx) * 10 * 256;
        switch (wave) {
            case 0: sc_fused_finish<0>(stage, pt, S.s.tile, S.s.wrow, out); break;
            case 1: sc_fused_finish<1>(stage, pt, S.s.tile, S.s.wrow, out); break;
            case 2: sc_fused_finish<2>(stage, pt, S.s.tile, S.s.wrow, out); break;
            default: sc_fused_finish<3>(stage, pt, S.s.tile, S.s.wrow, out); break;
        }
    } else {
        if (vd == 0) { A.pid = alt.pid; A.pidz = alt.pidz; A.pdeltaF = alt.pdeltaF; A.calib = alt.calib; precalc = alt.precalc; }
        const EFConst C = ef_const(Cin, A);
        const int q = b - n_sc;
        top_gram_body(C, A, precalc, top_partial, nres_partial, q % top_chunks, q / top_chunks, top_chunks, S.t, nullptr, ranges);
    }
}
__global__ void __launch_bounds__(256) k_ef_acc_fused(const PrecalcDev* __restrict__ precalc, EFConst Cin, EFArrays A,
                                                      const int* __restrict__ phost, float* __restrict__ top_partial,
                                                      int* __restrict__ nres_partial, int top_chunks, float* __restrict__ sc_partial,
                                                      int sc_chunks, int n_sc, AccAlt alt) {
    __shared__ AccSmem S;
    acc_fused_body(S, precalc, Cin, A, phost, top_partial, nres_partial, top_chunks, sc_partial, sc_chunks, n_sc, alt, (int)blockIdx.x);
}

// ---- marginalizePointsF (EnergyFunctional.cpp:514-549): the same three bodies in MODE 2 over the points flagged by `mask` ----
__global__ void __launch_bounds__(256) k_ef_marg_stage1(EFConst C, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                        const int* __restrict__ phost, float* __restrict__ top_partial,
                                                        int* __restrict__ nres_partial, int top_chunks, int n_top,
                                                        const uint8_t* __restrict__ mask, float* __restrict__ prior_w) {
    __shared__ union U { TopGramSmem t; PointSmem p; __device__ U() {} } S;
    const int b = blockIdx.x;
    if (b < n_top) top_gram_body<2>(C, A, precalc, top_partial, nres_partial, b % top_chunks, b / top_chunks, top_chunks, S.t, mask);
    else point_body<2>(C, A, precalc, phost, (b - n_top) * 64, min(64, C.nP - (b - n_top) * 64), S.p, mask, prior_w);
}
__global__ void __launch_bounds__(256) k_ef_marg_sc_gram(EFConst C, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                         float* __restrict__ partial, int pts_per_block, const uint8_t* __restrict__ mask) {
    __shared__ ScGramSmem S;
    sc_gram_body<2>(C, A, precalc, partial, pts_per_block, blockIdx.x, blockIdx.y, gridDim.x, S, mask);
}

// EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:45-55) for the active residuals of the flagged points: one thread per slot.
// res_toZeroF = resF - (Jpdxi . adHTdeltaF + Jpdc . cDeltaF + Jpdd * deltaF), isLinearized = true.
__global__ void __launch_bounds__(256) k_ef_fix_linearization(EFConst C, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                              const int* __restrict__ phost, const uint8_t* __restrict__ mask) {
    const size_t slots = (size_t)C.nF * C.nP;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    const int p = (int)(s % C.nP), t = (int)(s / C.nP);
    if (!mask[p]) return;
    const int h = phost[p];
    const PrecalcDev& pc = precalc[h * C.nF + t];
    if (precalc[h * C.nF + h].np == 0) return;   // host frame not in this rank's shard
    uint8_t fl = A.rflags[s];
    if (!(fl & RF_EXISTS) || !(fl & RF_ACTIVE)) return;
    const float* Je = A.J + (size_t)((fl & RF_SEL) ? 1 : 0) * kJPlanes * slots + s;
    const float dd = A.pdeltaF[p];
    float dx = 0, dy = 0, cx = 0, cy = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { dx += Je[(2 + i) * slots] * pc.dp[i]; dy += Je[(8 + i) * slots] * pc.dp[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { cx += Je[(14 + i) * slots] * C.cDeltaF[i]; cy += Je[(18 + i) * slots] * C.cDeltaF[i]; }
    const float jx = dx + cx + Je[22 * slots] * dd, jy = dy + cy + Je[23 * slots] * dd;
    A.rres_toZero[s] = Je[0] - jx;
    A.rres_toZero[slots + s] = Je[slots] - jy;
    A.rflags[s] = fl | RF_LINEARIZED;
}

// removePoint / dropPointsF for the flagged points: their residual slots cease to exist
__global__ void __launch_bounds__(256) k_ef_remove_points(int nF, int nP, uint8_t* __restrict__ rflags, const uint8_t* __restrict__ marg,
                                                          const uint8_t* __restrict__ drop) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (size_t)nF * nP) return;
    const int p = (int)(s % nP);
    if (marg[p] || (drop && drop[p])) rflags[s] = 0;
}

// Fixed-order fp64 sum of the per-workgroup partial Gram tiles into the PACKED accumulator buffer, all three parts in one launch and one
// memory round trip: top Gram [pairs][121] (the live 11x11), SC Gram [nF][1431] (upper triangle of the live 53x53), resInA.
// Every task sums four adjacent columns of one tile row with 16-byte buffer loads (the chunk stride in the SCALAR offset: one per-lane
// offset register for all loads in flight), all loads first, then the running sums in chunk order:
//   top: task = (pair, row r of 11, column group g of 3), chunks 0 .. top_chunks-1 in order
//   SC : task = (host, tile A of 10, row r of 16, column group g of 4, part p of 4); part p sums chunks [p*per, (p+1)*per), per =
//        ceil(sc_chunks / 4), and four neighbouring lanes combine (p0 + p1) + (p2 + p3) -- a fixed order, the same on every path.
template <int PM>
__device__ __forceinline__ void sum4_chunks_f64(__amdgpu_buffer_rsrc_t rsrc, int voff, int chunk_bytes, int c0, int cn, int c_last, double* acc4) {
    u32x4 v[PM];
#pragma unroll
    for (int j = 0; j < PM; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, min(c0 + j, c_last) * chunk_bytes, 0);
#pragma unroll
    for (int j = 0; j < PM; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[i] += (j < cn) ? (double)__uint_as_float(v[j][i]) : 0.0;   // + 0.0 for the absent chunks is exact
}
__device__ __forceinline__ int sc_packed_index(int a, int b) { return a * 53 - (a * (a - 1)) / 2 + (b - a); }   // upper triangle, a <= b

constexpr int kScTasksPerHost = 10 * 16 * 4 * 4;   // 2560
static inline int acc_reduce_grid(int pairs, int nF) { return (nF * kScTasksPerHost) / 256 + (pairs * 33 + 255) / 256 + 1; }
__device__ __forceinline__ void acc_reduce_body(const float* __restrict__ top_partial, int pairs, int top_chunks,
                                                const float* __restrict__ sc_partial, int nF, int sc_chunks,
                                                const int* __restrict__ nres_partial, double* __restrict__ out, const int b) {
    const int ntop = pairs * 121, nsc = nF * 1431;
    const int nb_sc = (nF * kScTasksPerHost) / 256, nb_top = (pairs * 33 + 255) / 256;
    if (b < nb_sc) {
        const int u = b * 256 + threadIdx.x;
        const int h = u / kScTasksPerHost, uu = u - h * kScTasksPerHost;      // wave-uniform: 2560 % 256 == 0
        const int pq = uu & 3, g = (uu >> 2) & 3, r = (uu >> 4) & 15, A = uu >> 8;
        const float* hb = sc_partial + (size_t)__builtin_amdgcn_readfirstlane(h) * sc_chunks * 2560;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb), 0, sc_chunks * 10240, 0x00020000);
        const int per = (sc_chunks + 3) >> 2;
        const int c_lo = pq * per, c_n = max(0, min(sc_chunks, c_lo + per) - c_lo);
        const int voff = 4 * (A * 256 + r * 16 + 4 * g);
        double s4[4] = {0, 0, 0, 0};
        if (per <= 8) sum4_chunks_f64<8>(rsrc, voff, 10240, c_lo, c_n, sc_chunks - 1, s4);
        else sum4_chunks_f64<16>(rsrc, voff, 10240, c_lo, c_n, sc_chunks - 1, s4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { s4[i] += __shfl_down(s4[i], 1); s4[i] += __shfl_down(s4[i], 2); }
        if (pq == 0) {
            const int ti = A < 4 ? 0 : (A < 7 ? 1 : (A < 9 ? 2 : 3)), tj = A < 4 ? A : (A < 7 ? A - 3 : (A < 9 ? A - 5 : 3));
            const int a = ti * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tj * 16 + 4 * g + i;
                if (a <= c && c < 53) out[(size_t)ntop + (size_t)h * 1431 + sc_packed_index(a, c)] = s4[i];
            }
        }
    } else if (b < nb_sc + nb_top) {
        const int u = (b - nb_sc) * 256 + threadIdx.x;
        const bool on = u < pairs * 33;
        const int uu = on ? u : 0;
        const int pair = uu / 33, e = uu - pair * 33, r = e / 3, g = e - r * 3;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(top_partial), 0, pairs * top_chunks * 1024, 0x00020000);
        const int voff = 4 * (pair * top_chunks * 256 + r * 16 + 4 * g);
        double s4[4] = {0, 0, 0, 0};
        for (int c0 = 0; c0 < top_chunks; c0 += 16) sum4_chunks_f64<16>(rsrc, voff, 1024, c0, on ? min(16, top_chunks - c0) : 0, top_chunks - 1, s4);
        if (on) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = 4 * g + i; if (c < 11) out[(size_t)pair * 121 + r * 11 + c] = s4[i]; }
        }
    } else {   // resInA: integer sum, order-free
        __shared__ int part[4];
        int n = 0;
        for (int i = threadIdx.x; i < pairs * top_chunks; i += 256) n += nres_partial[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
        __syncthreads();
        if (threadIdx.x == 0) out[ntop + nsc] = (double)(part[0] + part[1] + part[2] + part[3]);
    }
}
__global__ void __launch_bounds__(256) k_ef_acc_reduce(const float* __restrict__ top_partial, int pairs, int top_chunks,
                                                       const float* __restrict__ sc_partial, int nF, int sc_chunks,
                                                       const int* __restrict__ nres_partial, double* __restrict__ out,
                                                       const int* __restrict__ skip_verdict = nullptr /* non-NULL: return at once when it reads 0 (AccAlt::skip_on_reject) */) {
    if (skip_verdict && *skip_verdict == 0) return;
    acc_reduce_body(top_partial, pairs, top_chunks, sc_partial, nF, sc_chunks, nres_partial, out, (int)blockIdx.x);
}

// resubstituteFPt (EnergyFunctional.cpp:250-282): workgroup = 64 points x 8 waves; wave t forms xAd[h,t] . JpJdF of the
// residual in target t, wave 0 subtracts the 8 terms in ascending target order.  Also does backupState for the point
// (idepth_backup = idepth, FullSystemOptimize.cpp:300-305) and the per-block partial sums of step^2 and |idepth_backup|
// that doStepFromBackup needs (:236-249).  xAd: [nF(host)][nF(target)][6] floats (index nF*h + t), xc: 4 floats.
// The solution (xc and the nF*nF adjoint-transformed frame steps) is read from device memory, where the device-side solve
// (backend_solve.inc) leaves it: the host need not have seen x when this kernel is launched.  step_fac >= 0 additionally performs doStepFromBackup for the point
// (idepth = idepth_zero = backup + step_fac * step, FullSystemOptimize.cpp:236-249) -- the optimize loop always does both.
struct ResubX { float xc[4]; float xAd[kMaxFrames * kMaxFrames * 6]; };

// A launch may wait for values another workgroup OF THE SAME LAUNCH publishes (k_ef_tail_resub: the factorisation workgroup's solution).
// The solution travels WITHOUT fences: every value is one 64-bit word {number of the solve, payload}, written with one
// relaxed device-scope atomic store (single-copy atomic: a reader sees the tag and its payload together or neither) and polled by the lane
// that needs it until the tag is the current one.  The publisher's release fence (an L2 write-back on this part) and the readers' acquire
// fence + second round trip for the payload are gone: measured 2.9 us from the publisher's last store to the first reader seeing the word
// with the fence pair (tools/exp_solve_stamps.py), see profiles/r03_notes.txt for the tagged form.
constexpr int kXwFloats = 4 + kMaxFrames * kMaxFrames * 6;     // ResubX as tagged words [0, kXwFloats), then x as (hi, lo) pairs
__device__ __forceinline__ void store_tagged_u32(unsigned long long* w, unsigned seq, unsigned payload) {
    __hip_atomic_store(w, ((unsigned long long)seq << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned poll_tagged_u32(const unsigned long long* w, unsigned seq, unsigned* err) {
    unsigned long long v;
    int polls = 0;
    while ((unsigned)((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != seq) {
        if (++polls >= (1 << 20)) { ef_raise(err, 2u); break; }   // gave up: the call must fail (sticky)
        __builtin_amdgcn_s_sleep(4);
    }
    return (unsigned)v;
}

// (body of a workgroup of 512 lanes = 8 waves: it takes the two 64-point blocks blk0 and blk0 + 1; wave t forms the term of target t for both,
// waves 0 and 1 then finish one block each.  The kernel -- k_ef_tail_resub, backend_solve.inc -- runs these workgroups beside the factorisation
// workgroup whose solution they wait for, with their own loads already in flight, plus one workgroup that performs the calib / frame part of
// doStepFromBackup and writes the precalc table of the stepped state)
struct ResubSmem { float part[2][kMaxFrames][2][64]; float sx[4 + kMaxFrames * kMaxFrames * 6]; unsigned xs[2 * (4 + 6 * kMaxFrames)]; };
// xsol / adHostF: the solution x of the device-side solve (doubles, SolveSys::x) and the host adjoints (SolveWindow::adHostF, [h + t nF][36]): every
// workgroup forms xc = (float) x[0..3] and xAd[nF h + t] = x_h AH(h,t) + x_t adTarget (resubstituteF_MT's frame part, EnergyFunctional.cpp:221-240; adTargetF =
// diag(SCALE_XI_TRANS x3, SCALE_XI_ROT x3), :36-48; float arithmetic like the reference) from the 52 doubles itself -- until round 6 the factorisation
// workgroup did that for everybody and published 388 words; now it publishes x the moment it has it.
__device__ __forceinline__ void resubstitute_body(const EFConst& C, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                                  const int* __restrict__ phost, const double* __restrict__ xsol, const float* __restrict__ adHostF, float* __restrict__ backup,
                                                  double* __restrict__ stats_partial, float step_fac,
                                                  float* __restrict__ pid_w, float* __restrict__ pidz_w, float* __restrict__ pdeltaF_w,
                                                  int n_point_blocks, int blk0, ResubSmem& S, const unsigned long long* xw /*NULL: xsol is complete*/, unsigned seq) {
    float* sx = S.sx;
    const int lane = threadIdx.x & 63, t = (threadIdx.x >> 6) & 7;
    const size_t slots = (size_t)C.nF * C.nP;
    // every global input of this thread up front (independent loads; all slots / points have storage behind them), the
    // dependent decisions afterwards: two memory round trips instead of five
    int p[2], h[2];
    bool inP[2];
    uint8_t fl[2];
    float jp[2][6];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        p[q] = (blk0 + q) * 64 + lane;
        inP[q] = p[q] < C.nP;
        const bool slot_ok = inP[q] && t < C.nF;
        const size_t s = slot_ok ? (size_t)t * C.nP + p[q] : 0;
        h[q] = inP[q] ? phost[p[q]] : 0;
        fl[q] = A.rflags[s];
#pragma unroll
        for (int i = 0; i < 6; ++i) jp[q][i] = A.JpJd[(size_t)i * slots + s];
    }
    // the wave that finishes block t (t < 2) needs the point's own terms
    const int qf = t & 1;
    const bool fin = t < 2;
    const int pf = (fin && inP[qf]) ? p[qf] : 0;
    const float bsum = A.pbdSum[pf];
    float hca[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hca[i] = A.pHcdA[(size_t)i * C.nP + pf];
    const float hdi = A.pHdi[pf], pidv = A.pid[pf];
    const uint8_t sens = A.psensor[pf];
    // lane v of the workgroup forms word v of {xc[4], xAd[nF * nF][6]}: its column of the pair's adjoint
    const int nxw = 4 + C.nF * C.nF * 6, nsol = 4 + 6 * C.nF;
    const int v = min((int)threadIdx.x, nxw - 1);
    const int vk = v >= 4 ? (v - 4) / 6 : 0, vc = v >= 4 ? (v - 4) - 6 * vk : 0, vh = vk / C.nF, vt = vk - vh * C.nF;
    float ahc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) ahc[q] = adHostF[(size_t)(vh + vt * C.nF) * 36 + q * 6 + vc];
    // one pass over the argument block into LDS (the per-lane host index below would otherwise turn every use into a
    // vector load from the kernel-argument segment)
    __builtin_amdgcn_sched_barrier(0);
    bool mine[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) mine[q] = inP[q] && precalc[h[q] * C.nF + h[q]].np != 0;
    // this thread's loads above are in flight while the solution is being computed; its words of x arrive tagged (same launch: (hi, lo) pairs behind the
    // kXwFloats words of the earlier layout) or are in memory
    if ((int)threadIdx.x < 2 * nsol) {
        const int i = (int)threadIdx.x >> 1;
        S.xs[threadIdx.x] = xw ? poll_tagged_u32(xw + kXwFloats + threadIdx.x, seq, A.err)
                               : (unsigned)((threadIdx.x & 1) ? __double2loint(xsol[i]) : __double2hiint(xsol[i]));
    }
    __syncthreads();
    if ((int)threadIdx.x < nxw) {
        auto xd = [&](int i) -> double { return __hiloint2double((int)S.xs[2 * i], (int)S.xs[2 * i + 1]); };
        if (threadIdx.x < 4) sx[threadIdx.x] = (float)xd((int)threadIdx.x);
        else {
            const float sTf[6] = {0.5f, 0.5f, 0.5f, 1.0f, 1.0f, 1.0f};
            float a = 0, b = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) { a += (float)xd(4 + 6 * vh + q) * ahc[q]; b += (float)xd(4 + 6 * vt + q) * (q == vc ? sTf[vc] : 0.0f); }
            sx[4 + (size_t)(C.nF * vh + vt) * 6 + vc] = a + b;
        }
    }
    __syncthreads();
    const float* xc = sx;
    const float* xAd = sx + 4;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float dotv = 0.0f, good = 0.0f;
        if (mine[q] && t < C.nF) {
            if ((fl[q] & RF_EXISTS) && (fl[q] & RF_ACTIVE)) {
                good = 1.0f;
                const float* xa = xAd + (size_t)(C.nF * h[q] + t) * 6;
                float sum = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i) sum += xa[i] * jp[q][i];
                dotv = sum;
            }
        }
        S.part[q][t][0][lane] = dotv; S.part[q][t][1][lane] = good;
    }
    __syncthreads();
    if (!fin) return;
    float (*part)[2][64] = S.part[qf];
    const int blk = blk0 + qf;
    double s2 = 0, sa = 0;
    if (mine[qf]) {
        float b = bsum;
        float dot = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) dot += xc[i] * hca[i];
        b -= dot;
        float ngood = 0;
        for (int tt = 0; tt < C.nF; ++tt) {
            if (part[tt][1][lane] != 0.0f) { b -= part[tt][0][lane]; ngood += 1.0f; }
        }
        float step = 0.0f;
        if (ngood > 0 && !sens) step = -b * hdi;
        A.pstep[pf] = step;
        const float idb = pidv * (1.0f / SDVGN_SCALE_IDEPTH);
        backup[pf] = idb;
        if (step_fac >= 0.0f) {
            const float v = idb + step_fac * step;
            pid_w[pf] = SDVGN_SCALE_IDEPTH * v;
            pidz_w[pf] = SDVGN_SCALE_IDEPTH * v;
            pdeltaF_w[pf] = v - v;
        }
        s2 = (double)(step * step);
        sa = (double)fabsf(idb);
    }
    s2 = wave_sum_double(s2); sa = wave_sum_double(sa);
    if (lane == 63 && blk < n_point_blocks) { stats_partial[blk] = s2; stats_partial[n_point_blocks + blk] = sa; }
}

// ------------------------------------------------------------------------------------------------------------
// FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:63-97), run by the reference at the end of EVERY linearizeAll (:122):
//   allResVec = state_NewEnergyWithOutlier of the active (non-linearised) residuals with target == newest frame and value >= 0
//   nth = allResVec[(int)(0.7f * size)] after nth_element ; TH = ((26*0.5 + 1.5*sqrt(nth)*0.5))^2 ; empty -> 12*12*8
// The k-th smallest of <= nP non-negative floats is found EXACTLY by a radix descent on the float bit pattern (monotone for
// non-negative floats): two bits per step, each step one block-wide count of keys below three candidate prefixes.  One workgroup
// of 1024 lanes, keys in registers (16 per lane up to 16384 points, the rest re-read from memory), no atomics: deterministic.
// SRC 0: candidates from the state_NewEnergyWithOutlier plane of the newest target (slots (nF-1)*nP + p) and the slot flags.
// SRC 1: candidates from a double buffer cand[p] = energy + 1 (0 = no candidate), the sum over the ranks of a sharded window.
// th_out[0..nF-2] = th_prev[0..nF-2]; th_out[nF-1] = new threshold; *log_slot (pinned, may be NULL) = new threshold.
// ------------------------------------------------------------------------------------------------------------
constexpr int kSelLanes = 1024, kSelVPT = 16;
// 64-lane integer sum with DPP moves (no LDS crossbar); result valid in lane 63
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_move_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ unsigned wave_sum_dpp_u32(unsigned v) {
    v += dpp_move_u32<0x111, 0xF>(v);   // row_shr:1
    v += dpp_move_u32<0x112, 0xF>(v);   // row_shr:2
    v += dpp_move_u32<0x114, 0xF>(v);   // row_shr:4
    v += dpp_move_u32<0x118, 0xF>(v);   // row_shr:8  -> lane 15 of every row holds the row sum
    v += dpp_move_u32<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v += dpp_move_u32<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
    return v;
}
// inclusive 64-lane prefix sum with DPP moves
__device__ __forceinline__ unsigned wave_scan_dpp_u32(unsigned v) { return wave_sum_dpp_u32(v); }   // the sum IS built as an inclusive scan
struct SelectSmem { unsigned hist[2 * kSelLanes]; unsigned wsum[kSelLanes / 64]; unsigned sel[2]; };
// own0/own1: the point range [own0, own1) hosted by this rank's key-frames (the planes of other points are not written here)
template <int SRC>
__device__ __forceinline__ unsigned sel_key(int p, int nF, int nP, int own0, int own1, const uint8_t* __restrict__ rflags,
                                             const float* __restrict__ wo, const double* __restrict__ cand) {
    if (p >= nP) return 0xFFFFFFFFu;
    float v;
    if (SRC == 0) {
        const size_t s = (size_t)(nF - 1) * nP + p;
        const uint8_t fl = rflags[s];
        v = wo[s];
        if (!(fl & RF_EXISTS) || (fl & RF_LINEARIZED) || p < own0 || p >= own1) return 0xFFFFFFFFu;
    } else {
        const double c = cand[p];
        if (!(c > 0.5)) return 0xFFFFFFFFu;
        v = (float)(c - 1.0);
    }
    if (!(v >= 0.0f)) return 0xFFFFFFFFu;     // -1 (OOB) and NaN are not candidates (`state_NewEnergyWithOutlier >= 0`)
    return __float_as_uint(v) & 0x7FFFFFFFu;  // -0.0f counts as 0
}

// body for one workgroup of LANES lanes (1024: the select kernels and the statistics launches; 512: as a workgroup of the factorisation's launch).
// The result does not depend on LANES (an exact selection).  Lane t owns the 2048 / LANES adjacent histogram bins from t * (2048 / LANES).
template <int SRC, int LANES = kSelLanes>
__device__ __forceinline__ void select_th_body(int nF, int nP, int own0, int own1, const uint8_t* __restrict__ rflags, const float* __restrict__ wo,
                                               const double* __restrict__ cand, const float* __restrict__ th_prev, float* __restrict__ th_out,
                                               float* __restrict__ log_slot, SelectSmem& S,
                                               unsigned long long* __restrict__ thw = nullptr /* the thresholds as tagged words too: a reader in the same launch */, unsigned thseq = 0) {
    constexpr int VPT = kSelLanes * kSelVPT / LANES;      // keys in registers: 16384 points whatever LANES is
    constexpr int BPL = 2 * kSelLanes / LANES;            // histogram bins per lane
    static_assert(LANES % 64 == 0 && LANES <= kSelLanes && (2 * kSelLanes) % LANES == 0, "select_th_body: lane count");
    unsigned* s_hist = S.hist; unsigned* s_wsum = S.wsum; unsigned* s_sel = S.sel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned key[VPT];
    {   // all loads of this lane in ONE batch (clamped addresses), the candidate tests afterwards: one memory round trip, not 16
        uint8_t fl[VPT];
        float v[VPT];
        double c[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int p = min(tid + i * LANES, nP - 1);
            if (SRC == 0) { const size_t s = (size_t)(nF - 1) * nP + p; fl[i] = rflags[s]; v[i] = wo[s]; }
            else c[i] = cand[p];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int p = tid + i * LANES;
            bool ok = p < nP;
            float e;
            if (SRC == 0) { ok = ok && (fl[i] & RF_EXISTS) && !(fl[i] & RF_LINEARIZED) && p >= own0 && p < own1; e = v[i]; }
            else { ok = ok && c[i] > 0.5; e = (float)(c[i] - 1.0); }
            ok = ok && e >= 0.0f;
            key[i] = ok ? (__float_as_uint(e) & 0x7FFFFFFFu) : 0xFFFFFFFFu;
        }
    }
    const int p_tail = LANES * VPT;
    // MSB-first radix select with LDS histograms, 11 + 11 + 9 bits: per pass every key that still matches the prefix adds 1 to its
    // digit's bin (integer LDS atomics: order-free, deterministic), a block scan finds the bin holding rank k.  (A compare-and-count
    // descent, 2 bits per step, costs 16384 keys x 3 compares x 17 steps on ONE CU's vector units: measured 26 us; this: ~5 us.)
    unsigned prefix = 0, pmask = 0x80000000u;   // valid keys have bit 31 clear; 0xFFFFFFFF never matches
    int N = 0, kth = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 20 : (pass == 1 ? 9 : 0);
        const int nb = pass == 2 ? 512 : 2048;
#pragma unroll
        for (int q = 0; q < BPL; ++q) s_hist[tid + q * LANES] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            if ((key[i] & pmask) == prefix) atomicAdd(&s_hist[(key[i] >> shift) & (nb - 1)], 1u);
        for (int p0 = p_tail; p0 < nP; p0 += LANES) {   // windows beyond 16384 points: re-read (uniform trip count)
            const unsigned k = sel_key<SRC>(p0 + tid, nF, nP, own0, own1, rflags, wo, cand);
            if ((k & pmask) == prefix) atomicAdd(&s_hist[(k >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        // lane t owns bins BPL t .. BPL t + BPL - 1: exclusive block scan of the lanes' sums
        unsigned cb[BPL], mysum = 0;
#pragma unroll
        for (int q = 0; q < BPL; ++q) { cb[q] = s_hist[BPL * tid + q]; mysum += cb[q]; }
        const unsigned incl = wave_scan_dpp_u32(mysum);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < LANES / 64; ++w) { const unsigned t = s_wsum[w]; if (w < wave) before += t; total += t; }
        if (pass == 0) { N = (int)total; kth = (int)(0.7f * (float)N); }   // setting_frameEnergyTHN * allResVec.size(): float product, truncated
        if (N == 0) break;
        const unsigned excl = before + incl - mysum;
        if ((unsigned)kth >= excl && (unsigned)kth < excl + mysum) {     // exactly one lane
            unsigned below = excl;
            int bin = 0;
#pragma unroll
            for (int q = 0; q < BPL - 1; ++q) if ((unsigned)kth >= below + cb[q] && bin == q) { below += cb[q]; bin = q + 1; }
            s_sel[0] = BPL * tid + bin;
            s_sel[1] = below;
        }
        __syncthreads();
        prefix |= s_sel[0] << shift;
        kth -= (int)s_sel[1];
        pmask |= (unsigned)(nb - 1) << shift;
    }
    float th;
    if (N == 0) {
        th = 12 * 12 * 8;
    } else {
        const float nthElement = sqrtf(__uint_as_float(prefix));
        th = nthElement * 1.5f;                       // setting_frameEnergyTHFacMedian
        th = 26.0f * 0.5f + th * (1 - 0.5f);          // setting_frameEnergyTHConstWeight
        th = th * th;
        th *= 1.0f * 1.0f;                            // setting_overallEnergyTHWeight^2
    }
    if (tid == 0 && log_slot) *log_slot = th;
    if (tid < nF - 1) { const float tp = th_prev[tid]; if (th_out) th_out[tid] = tp; if (thw) store_tagged_u32(thw + tid, thseq, __float_as_uint(tp)); }
    if (tid == 0) {
        if (th_out) th_out[nF - 1] = th;
        if (thw) store_tagged_u32(thw + nF - 1, thseq, __float_as_uint(th));
    }
}

// arguments of one pending setNewFrameEnergyTH (SRC 0) when it rides in another kernel's launch as an extra workgroup
struct SelArgs { int nF, nP, own0, own1; const uint8_t* rflags; const float* wo; const float* th_prev; float* th_out; float* log_slot; };

template <int SRC>
__global__ void __launch_bounds__(kSelLanes) k_ef_select_th(int nF, int nP, int own0, int own1, const uint8_t* __restrict__ rflags,
                                                            const float* __restrict__ wo, const double* __restrict__ cand,
                                                            const float* __restrict__ th_prev, float* __restrict__ th_out, float* __restrict__ log_slot) {
    __shared__ SelectSmem S;
    select_th_body<SRC>(nF, nP, own0, own1, rflags, wo, cand, th_prev, th_out, log_slot, S);
}

// sharded windows: this rank's candidates of the quantile above as doubles (energy + 1, 0 = none), summed over the ranks by the
// same all-reduce that carries the four linearize statistics
__global__ void __launch_bounds__(256) k_ef_pack_th_candidates(int nF, int nP, int own0, int own1, const uint8_t* __restrict__ rflags,
                                                               const float* __restrict__ wo, double* __restrict__ cand) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nP) return;
    const unsigned k = sel_key<0>(p, nF, nP, own0, own1, rflags, wo, nullptr);
    cand[p] = (k == 0xFFFFFFFFu) ? 0.0 : (double)__uint_as_float(k) + 1.0;
}

// After a REJECTED step the reference re-linearises the restored state (FullSystemOptimize.cpp:446-449).  Everything that
// re-linearisation computes is what the kept state_New* set already holds, EXCEPT the IN / OUTLIER classification and the clamped
// state_NewEnergy: those are taken under the threshold the trial's linearizeAll just set (th), not the one the kept set was built
// with.  This kernel repeats exactly that decision (Residuals.cpp:210-222) for the residuals that involve the newest frame (the
// only threshold that moves): lanes [0, nP) = target == newest, then np_last x (nF-1) lanes = host == newest.
struct ReclArgs {
    int nF, nP, P0_last, np_last;
    const uint8_t* rflags; const float* wo; int8_t* rstate_new; float* renergy_new; const int* phost; const PrecalcDev* precalc; const float* th;
    const unsigned long long* thw = nullptr; unsigned thseq = 0;   // non-NULL: the thresholds are being selected in THIS launch and arrive as tagged words (select_th_body)
    unsigned* err = nullptr;   // the handle's sticky error word (EFArrays::err): a threshold that did not arrive in time fails the call like every other intra-launch wait
};
__device__ __forceinline__ int reclassify_count(const ReclArgs& a) { return a.nP + a.np_last * (a.nF - 1); }
__device__ __forceinline__ void reclassify_slot(const ReclArgs& a, int i) {
    const int nF = a.nF, nP = a.nP;
    int p, t;
    if (i < nP) { p = i; t = nF - 1; }
    else { const int j = i - nP; if (j >= a.np_last * (nF - 1)) return; t = j / a.np_last; p = a.P0_last + j % a.np_last; }
    const int h = a.phost[p];
    if (h == t || a.precalc[h * nF + h].np == 0) return;
    const size_t s = (size_t)t * nP + p;
    const uint8_t fl = a.rflags[s];
    const int sn = a.rstate_new[s];
    const float e = a.wo[s];
    if (!(fl & RF_EXISTS) || (fl & RF_LINEARIZED) || (sn & RS_MASK) == RS_OOB) return;
    float ta, tb;
    if (a.thw) { ta = __uint_as_float(poll_tagged_u32(a.thw + h, a.thseq, a.err)); tb = __uint_as_float(poll_tagged_u32(a.thw + t, a.thseq, a.err)); }
    else { ta = a.th[h]; tb = a.th[t]; }
    const float frameTH = ta < tb ? tb : ta;
    const bool wjlow = (sn & RS_WJLOW) != 0;
    if (e > frameTH || wjlow) { a.renergy_new[s] = frameTH; a.rstate_new[s] = (int8_t)(RS_OUTLIER | (wjlow ? RS_WJLOW : 0)); }
    else { a.renergy_new[s] = e; a.rstate_new[s] = RS_IN; }
}
__global__ void __launch_bounds__(256) k_ef_reclassify(ReclArgs a) {
    reclassify_slot(a, blockIdx.x * 256 + threadIdx.x);
}

// linearizeAll(true)'s per-residual epilogue (FullSystemOptimize.cpp:32-52, 136-155) after linearize + applyRes: one lane per point.
// For every residual of the point that was in activeResiduals (exists, not linearised): still active -> relBS (the relative baseline,
// 0.01 * pixel distance between the projections at infinite and at real depth) feeds the point's maxRelBaseline, numGoodResiduals++;
// not active -> toRemove: the slot ceases to exist.
__global__ void __launch_bounds__(256) k_ef_finish_points(EFConst C, EFArrays A, const PrecalcDev* __restrict__ precalc, const int* __restrict__ phost,
                                                          float* __restrict__ relbs_max, int* __restrict__ ngood_inc, uint8_t* __restrict__ removed) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= C.nP) return;
    const int h = phost[p];
    const bool mine = precalc[h * C.nF + h].np != 0;
    const float pu = A.pu[p], pv = A.pv[p], ids = A.pid[p];
    float mx = 0.0f;
    int ng = 0;
    for (int t = 0; t < C.nF; ++t) {
        const size_t s = (size_t)t * C.nP + p;
        uint8_t rm = 0;
        const uint8_t fl = A.rflags[s];
        if (mine && (fl & RF_EXISTS) && !(fl & RF_LINEARIZED)) {
            if (fl & RF_ACTIVE) {
                const PrecalcDev& pc = precalc[h * C.nF + t];
                const float i0 = (pc.KRKi[0] * pu + pc.KRKi[1] * pv) + pc.KRKi[2] * 1.0f;
                const float i1 = (pc.KRKi[3] * pu + pc.KRKi[4] * pv) + pc.KRKi[5] * 1.0f;
                const float i2 = (pc.KRKi[6] * pu + pc.KRKi[7] * pv) + pc.KRKi[8] * 1.0f;
                const float q0 = i0 + pc.Kt[0] * ids, q1 = i1 + pc.Kt[1] * ids, q2 = i2 + pc.Kt[2] * ids;
                const float dx = i0 / i2 - q0 / q2, dy = i1 / i2 - q1 / q2;
                const float relBS = (float)(0.01 * (double)sqrtf(dx * dx + dy * dy));
                if (relBS > mx) mx = relBS;
                ng++;
            } else {
                rm = 1;
                A.rflags[s] = 0;
            }
        }
        removed[s] = rm;
    }
    relbs_max[p] = mx;
    ngood_inc[p] = ng;
}

}  // namespace sdvgn
