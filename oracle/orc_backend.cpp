// oracle/orc_backend.cpp -- TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY PINNED against the reference's own translation units (oracle/_ref/libref.so, oracle/README.md; tests/test_ref_pin*.py).
//
// Plain C++ restatement of the reference's sliding-window back end, SURVEY.md section 8 rows b1-b7, on a
// flattened copy of the EnergyFunctional graph (frames / points / residuals in the reference's iteration
// order: `for f in frames, for p in f->points, for r in p->residualsAll`):
//   b1 PointFrameResidual::linearize / applyRes      src/FullSystem/Residuals.cpp:60-224, 252-275
//      projectPoint helpers                          src/FullSystem/ResidualProjections.h:20-59
//      FrameFramePrecalc::set                        src/FullSystem/HessianBlocks.cpp:169-195
//   b2 AccumulatedTopHessianSSE::addPoint<0/1>       src/OptimizationBackend/AccumulatedTopHessian.cpp:14-112
//   b3 AccumulatorApprox                             src/OptimizationBackend/MatrixAccumulators.h:560-932
//   b4 stitchDoubleInternal / stitchDoubleMT         AccumulatedTopHessian.cpp:181-242, .h:63-114
//   b5 setAdjointsF / setDeltaF                      src/OptimizationBackend/EnergyFunctional.cpp:21-71,131-156
//   b6 solveSystemF / resubstituteF_MT               EnergyFunctional.cpp:650-759, 221-282
//   b7 AccumulatedSCHessianSSE::addPoint / stitch    src/OptimizationBackend/AccumulatedSCHessian.cpp:10-135
//      AccumulatorXX / AccumulatorX                  MatrixAccumulators.h:13-66,148-208
// The reference's own translation units are compiled unmodified into oracle/_ref/libref.so (oracle/Makefile target `ref`);
// tests/test_ref_pin_backend.py runs every function of this file against that build (bit-identical where the reference writes its
// arithmetic out).
// Float32 / float64 usage, operand order and the 1k/1M tiered accumulation follow the reference.
#include "orc_math.hpp"
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>

namespace orcb {
using namespace orc;

static const int CPARS = 4;
static const float SCALE_XI_ROT = 1.0f, SCALE_XI_TRANS = 0.5f, SCALE_F = 50.0f, SCALE_C = 50.0f, SCALE_A = 10.0f,
                   SCALE_B = 1000.0f, SCALE_IDEPTH = 1.0f;
// settings.cpp:21-27,65,101
static const float setting_idepthFixPrior = 50 * 50, setting_initialRotPrior = 1e11, setting_initialTransPrior = 1e10,
                   setting_initialCalibHessian = 5e9, setting_outlierTHSumComponent = 50 * 50, setting_huberTH = 6;
static const int patternNum = 8;
// settings.cpp:108-111
static const float setting_frameEnergyTHConstWeight = 0.5f, setting_frameEnergyTHN = 0.7f, setting_frameEnergyTHFacMedian = 1.5f,
                   setting_overallEnergyTHWeight = 1;
static const int patternP[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};  // settings.cpp:250

enum ResState { IN = 0, OOB = 1, OUTLIER = 2 };

// ---- b3: AccumulatorApprox (MatrixAccumulators.h:560-932) ---------------------------------------------
struct AccumulatorApprox {
    float Data[60], Data1k[60], Data1m[60];
    float TR[32], TR1k[32], TR1m[32];
    float BR[8], BR1k[8], BR1m[8];
    float numIn1, numIn1k, numIn1m;
    size_t num;
    float H[13][13];
    void initialize() {
        std::memset(this, 0, sizeof(*this));
    }
    void shiftUp(bool force) {
        if (numIn1 > 1000 || force) {
            for (int i = 0; i < 60; ++i) Data1k[i] = Data[i] + Data1k[i];
            for (int i = 0; i < 32; ++i) TR1k[i] = TR[i] + TR1k[i];
            for (int i = 0; i < 8; ++i) BR1k[i] = BR[i] + BR1k[i];
            numIn1k += numIn1; numIn1 = 0;
            std::memset(Data, 0, sizeof(Data)); std::memset(TR, 0, sizeof(TR)); std::memset(BR, 0, sizeof(BR));
        }
        if (numIn1k > 1000 || force) {
            for (int i = 0; i < 60; ++i) Data1m[i] = Data1k[i] + Data1m[i];
            for (int i = 0; i < 32; ++i) TR1m[i] = TR1k[i] + TR1m[i];
            for (int i = 0; i < 8; ++i) BR1m[i] = BR1k[i] + BR1m[i];
            numIn1m += numIn1k; numIn1k = 0;
            std::memset(Data1k, 0, sizeof(Data1k)); std::memset(TR1k, 0, sizeof(TR1k)); std::memset(BR1k, 0, sizeof(BR1k));
        }
    }
    // update(x4,x6,y4,y6,a,b,c) :715-808 with x = [x4;x6], y = [y4;y6]
    void update(const float* x4, const float* x6, const float* y4, const float* y6, float a, float b, float c) {
        float x[10], y[10];
        for (int i = 0; i < 4; ++i) { x[i] = x4[i]; y[i] = y4[i]; }
        for (int i = 0; i < 6; ++i) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
        int idx = 0;
        for (int r = 0; r < 10; ++r)
            for (int cc = r; cc < 10; ++cc) {
                Data[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
                idx++;
            }
        num++; numIn1++;
        shiftUp(false);
    }
    void updateTopRight(const float* x4, const float* x6, const float* y4, const float* y6, float TR00, float TR10,
                        float TR01, float TR11, float TR02, float TR12) {  // :810-859
        float x[10], y[10];
        for (int i = 0; i < 4; ++i) { x[i] = x4[i]; y[i] = y4[i]; }
        for (int i = 0; i < 6; ++i) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
        for (int i = 0; i < 10; ++i) {
            TR[3 * i + 0] += x[i] * TR00 + y[i] * TR10;
            TR[3 * i + 1] += x[i] * TR01 + y[i] * TR11;
            TR[3 * i + 2] += x[i] * TR02 + y[i] * TR12;
        }
    }
    void updateBotRight(float a00, float a01, float a02, float a11, float a12, float a22) {  // :861-875
        BR[0] += a00; BR[1] += a01; BR[2] += a02; BR[3] += a11; BR[4] += a12; BR[5] += a22;
    }
    void finish() {  // :584-616
        std::memset(H, 0, sizeof(H));
        shiftUp(true);
        int idx = 0;
        for (int r = 0; r < 10; ++r)
            for (int c = r; c < 10; ++c) { H[r][c] = H[c][r] = Data1m[idx]; idx++; }
        idx = 0;
        for (int r = 0; r < 10; ++r)
            for (int c = 0; c < 3; ++c) { H[r][c + 10] = H[c + 10][r] = TR1m[idx]; idx++; }
        H[10][10] = BR1m[0];
        H[10][11] = H[11][10] = BR1m[1];
        H[10][12] = H[12][10] = BR1m[2];
        H[11][11] = BR1m[3];
        H[11][12] = H[12][11] = BR1m[4];
        H[12][12] = BR1m[5];
        num = (size_t)(numIn1 + numIn1k + numIn1m);
    }
};

// ---- Accumulator11 (MatrixAccumulators.h:69-146): one float in SSE lane 0, three tiers shifted up every 1000 entries ----------------
struct Accumulator11 {
    float S[4], S1k[4], S1m[4];
    float numIn1, numIn1k, numIn1m;
    size_t num;
    float A;
    void initialize() { std::memset(this, 0, sizeof(*this)); }
    void shiftUp(bool force) {  // :130-145
        if (numIn1 > 1000 || force) {
            for (int i = 0; i < 4; ++i) S1k[i] = S[i] + S1k[i];
            numIn1k += numIn1; numIn1 = 0;
            std::memset(S, 0, sizeof(S));
        }
        if (numIn1k > 1000 || force) {
            for (int i = 0; i < 4; ++i) S1m[i] = S1k[i] + S1m[i];
            numIn1m += numIn1k; numIn1k = 0;
            std::memset(S1k, 0, sizeof(S1k));
        }
    }
    void updateSingle(float val) { S[0] += val; num++; numIn1++; shiftUp(false); }          // :92-98
    void updateSingleNoShift(float val) { S[0] += val; num++; numIn1++; }                    // :109-114
    void finish() { shiftUp(true); A = S1m[0] + S1m[1] + S1m[2] + S1m[3]; }                  // :86-90
};

// ---- the small dot products Eigen itself evaluates (`a.dot(b)`, `row * col`) ------------------------------------------------
// calcLEnergyPt, fixLinearizationF, addPoint<1> and resubstituteFPt contain 4- and 6-term float dot products written as Eigen expressions.
// The order in which Eigen adds the terms depends on its version and on whether it vectorises the expression, and Eigen is not available
// here to settle it (oracle/README.md).  The restatement adds left to right (order 0, what the kernels do as well); orders 1 and 2 exist
// for the sensitivity test (tests/test_oracle_backend.py): 1 = the halving unroller of Eigen's scalar reductions
// (redux_novec_unroller: f(0, n/2) + f(n/2, n - n/2), recursively), 2 = SSE packets (4 lanes multiplied at once, horizontal add
// (p0 + p2) + (p1 + p3), a scalar tail of 2 added as (p4 + p5)).
static int g_redux_order = 0;
static float dot_tree(const float* a, const float* b, int n) {
    if (n == 1) return a[0] * b[0];
    const int h = n / 2;
    return dot_tree(a, b, h) + dot_tree(a + h, b + h, n - h);
}
static inline float dot_small(const float* a, const float* b, int n) {   // n = 4 or 6
    if (g_redux_order == 1) return dot_tree(a, b, n);
    if (g_redux_order == 3) {   // what Eigen 3.2.8 built for SSE2 does per operand type (oracle/ref_shim/Eigen/Core, header): a 4-float vector is one
        // aligned packet (product, then the SSE2 horizontal add (p0 + p2) + (p1 + p3)); a 6-float vector has no packet access (24 bytes) and
        // goes through the halving unroller.  With this order the oracle matches the reference-built library bit for bit (tests/test_ref_pin_backend.py)
        if (n == 4) { const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2], p3 = a[3] * b[3]; return (p0 + p2) + (p1 + p3); }
        return dot_tree(a, b, n);
    }
    if (g_redux_order == 2) {
        const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2], p3 = a[3] * b[3];
        float s = (p0 + p2) + (p1 + p3);
        if (n == 6) s = s + (a[4] * b[4] + a[5] * b[5]);
        return s;
    }
    float s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

// ---- AccumulatorXX<i,j> / AccumulatorX<i> (MatrixAccumulators.h:13-66,148-208), i,j <= 8 ---------------
struct AccXX {
    int I, J;
    float A[64], A1k[64], A1m[64];
    float numIn1, numIn1k, numIn1m;
    size_t num;
    void initialize(int i, int j) { std::memset(this, 0, sizeof(*this)); I = i; J = j; }
    void shiftUp(bool force) {
        if (numIn1 > 1000 || force) {
            for (int k = 0; k < I * J; ++k) { A1k[k] += A[k]; A[k] = 0; }
            numIn1k += numIn1; numIn1 = 0;
        }
        if (numIn1k > 1000 || force) {
            for (int k = 0; k < I * J; ++k) { A1m[k] += A1k[k]; A1k[k] = 0; }
            numIn1m += numIn1k; numIn1k = 0;
        }
    }
    void update(const float* L, const float* R, float w) {  // A += w*L*R^T
        for (int r = 0; r < I; ++r) {
            const float wl = w * L[r];
            for (int c = 0; c < J; ++c) A[r * J + c] += wl * R[c];
        }
        numIn1++;
        shiftUp(false);
    }
    void updateVec(const float* L, float w) {  // AccumulatorX::update: A += w*L   (J == 1)
        for (int r = 0; r < I; ++r) A[r] += w * L[r];
        numIn1++;
        shiftUp(false);
    }
    void finish() { shiftUp(true); num = (size_t)(numIn1 + numIn1k + numIn1m); }
};

struct Precalc {  // FrameFramePrecalc, HessianBlocks.h:51-79
    float PRE_RTll[9], PRE_KRKiTll[9], PRE_RKiTll[9], PRE_RTll_0[9];
    float PRE_aff_mode[2], PRE_b0_mode;
    float PRE_tTll[3], PRE_KtTll[3], PRE_tTll_0[3];
};

struct RawJ {  // live part of RawResidualJacobian (RawResidualJacobian.h:7-36)
    float resF[2];
    float Jpdxi[2][6];
    float Jpdc[2][4];
    float Jpdd[2];
};

struct Frame {
    SE3 evalPT, PRE_worldToCam, PRE_camToWorld;
    double state[10], state_zero[10], state_scaled[10];
    double prior[6], delta_prior[6], delta[6];
    double step[10], state_backup[10];
    int frameID;
    float ab_exposure, frameEnergyTH;
    std::vector<float> dI;  // level-0 AoS {I,dx,dy}
};

struct Point {
    int host;
    float u, v, idepth, idepth_zero, idepth_scaled, idepth_zero_scaled;
    float color[8], weights[8];
    bool hasDepthPrior, isFromSensor;
    int r0, r1;  // residual range
    // EFPoint
    float priorF, deltaF, bdSumF, HdiF, Hdd_accLF, Hcd_accLF[4], bd_accLF, Hdd_accAF, Hcd_accAF[4], bd_accAF;
    float step, idepth_hessian, idepth_backup;
};

struct Residual {
    int point, host, target;
    int state_state, state_NewState;
    double state_energy, state_NewEnergy, state_NewEnergyWithOutlier;
    bool hasMatcher;
    double matcher[2];
    RawJ Jnew;   // PointFrameResidual::J
    RawJ Jef;    // EFResidual::J
    float res_toZeroF[2];
    float JpJdF[8];
    bool isLinearized, isActive;
    bool dropped;   // removed by the tail of optimize (linearizeAll(true)'s toRemove list)
    float centerProjectedTo[3];
};

struct EF {
    int w, h, nF;
    int resInM = 0;
    bool fixedIts = false;   // bench only: run exactly mnumOptIts loop bodies (no early break, no nF-dependent iteration count), like flags bit0 of sdvgn_ef_optimize
    int nThreads = 1;   // 1 = the reference's default (multiThreading = false); > 1 = its IndexThreadReduce paths, see solve_system
    // CalibHessian
    double value_scaled[4], value_minus_value_zero[4], value[4], value_zero[4], value_backup[4];
    float fxl, fyl, cxl, cyl, fxli, fyli, cxli, cyli;
    float wM3G, hM3G;
    std::vector<Frame> frames;
    std::vector<Point> points;
    std::vector<Residual> res;
    std::vector<Precalc> precalc;            // [host*nF + target]
    std::vector<double> adHost, adTarget;    // [h + t*nF][36]
    std::vector<float> adHostF, adTargetF;
    std::vector<float> adHTdeltaF;           // [h + t*nF][6]
    float cDeltaF[4];
    double cPrior[4];
    float cPriorF[4];
    std::vector<double> HM, bM;
    std::vector<std::vector<double>> nullspaces;
    // outputs of the last solve
    std::vector<double> HA, bA, Hsc, bsc, HFinal, bFinal, lastX;
    std::vector<AccumulatorApprox> accA, accL;
    int resInA, resInL;
    std::vector<float> scE, scEB, scD, scHcc, scbc;   // finished SC accumulators of the last solve
    double calibStep[4];
};

static void mat3f_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}
static void mat3f_vec(const float* A, const float* v, float* o) {
    for (int i = 0; i < 3; ++i) o[i] = (A[i * 3] * v[0] + A[i * 3 + 1] * v[1]) + A[i * 3 + 2] * v[2];
}

// FrameHessian::setState (HessianBlocks.h:131-143)
static void frame_set_state(Frame& f, const double* state) {
    for (int i = 0; i < 10; ++i) f.state[i] = state[i];
    for (int i = 0; i < 3; ++i) f.state_scaled[i] = SCALE_XI_TRANS * state[i];
    for (int i = 3; i < 6; ++i) f.state_scaled[i] = SCALE_XI_ROT * state[i];
    f.state_scaled[6] = SCALE_A * state[6]; f.state_scaled[7] = SCALE_B * state[7];
    f.state_scaled[8] = SCALE_A * state[8]; f.state_scaled[9] = SCALE_B * state[9];
    f.PRE_worldToCam = se3_mul(se3_exp(f.state_scaled), f.evalPT);
    f.PRE_camToWorld = se3_inverse(f.PRE_worldToCam);
}

// CalibHessian::setValueScaled-equivalent for the float views (HessianBlocks.h:302-330)
static void calib_update(EF* E) {
    E->fxl = (float)E->value_scaled[0]; E->fyl = (float)E->value_scaled[1];
    E->cxl = (float)E->value_scaled[2]; E->cyl = (float)E->value_scaled[3];
    E->fxli = 1.0f / E->fxl; E->fyli = 1.0f / E->fyl;
    E->cxli = -E->cxl / E->fxl; E->cyli = -E->cyl / E->fyl;
}

// FrameFramePrecalc::set (HessianBlocks.cpp:169-195) for every (host,target) + EnergyFunctional::setDeltaF
static void set_precalc(EF* E) {
    const int nF = E->nF;
    E->precalc.resize((size_t)nF * nF);
    float K[9] = {E->fxl, 0, E->cxl, 0, E->fyl, E->cyl, 0, 0, 1};
    float Ki[9];
    inv3f(K, Ki);
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            Precalc& P = E->precalc[(size_t)h * nF + t];
            const Frame& host = E->frames[h];
            const Frame& target = E->frames[t];
            SE3 l0 = se3_mul(target.evalPT, se3_inverse(host.evalPT));
            double R[9];
            quat_to_R(l0.q, R);
            for (int i = 0; i < 9; ++i) P.PRE_RTll_0[i] = (float)R[i];
            for (int i = 0; i < 3; ++i) P.PRE_tTll_0[i] = (float)l0.t[i];
            SE3 l = se3_mul(target.PRE_worldToCam, host.PRE_camToWorld);
            quat_to_R(l.q, R);
            for (int i = 0; i < 9; ++i) P.PRE_RTll[i] = (float)R[i];
            for (int i = 0; i < 3; ++i) P.PRE_tTll[i] = (float)l.t[i];
            float KR[9];
            mat3f_mul(K, P.PRE_RTll, KR);
            mat3f_mul(KR, Ki, P.PRE_KRKiTll);
            mat3f_mul(P.PRE_RTll, Ki, P.PRE_RKiTll);
            mat3f_vec(K, P.PRE_tTll, P.PRE_KtTll);
            double ab[2];
            aff_from_to(host.ab_exposure, target.ab_exposure, host.state_scaled[6], host.state_scaled[7],
                        target.state_scaled[6], target.state_scaled[7], ab);
            P.PRE_aff_mode[0] = (float)ab[0]; P.PRE_aff_mode[1] = (float)ab[1];
            P.PRE_b0_mode = (float)(host.state_zero[7] * SCALE_B);
        }
}

// EnergyFunctional::setAdjointsF (EnergyFunctional.cpp:21-71)
static void set_adjoints(EF* E) {
    const int nF = E->nF;
    E->adHost.assign((size_t)nF * nF * 36, 0); E->adTarget.assign((size_t)nF * nF * 36, 0);
    E->adHostF.assign((size_t)nF * nF * 36, 0); E->adTargetF.assign((size_t)nF * nF * 36, 0);
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            SE3 hostToTarget = se3_mul(E->frames[t].evalPT, se3_inverse(E->frames[h].evalPT));
            double Adj[36];
            se3_adj(hostToTarget, Adj);
            double* AH = &E->adHost[(size_t)(h + t * nF) * 36];
            double* AT = &E->adTarget[(size_t)(h + t * nF) * 36];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) { AH[r * 6 + c] = -Adj[c * 6 + r]; AT[r * 6 + c] = (r == c) ? 1.0 : 0.0; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) { AH[r * 6 + c] *= SCALE_XI_TRANS; AT[r * 6 + c] *= SCALE_XI_TRANS; }
            for (int r = 3; r < 6; ++r) for (int c = 0; c < 6; ++c) { AH[r * 6 + c] *= SCALE_XI_ROT; AT[r * 6 + c] *= SCALE_XI_ROT; }
            for (int i = 0; i < 36; ++i) {
                E->adHostF[(size_t)(h + t * nF) * 36 + i] = (float)AH[i];
                E->adTargetF[(size_t)(h + t * nF) * 36 + i] = (float)AT[i];
            }
        }
    for (int i = 0; i < 4; ++i) { E->cPrior[i] = setting_initialCalibHessian; E->cPriorF[i] = (float)E->cPrior[i]; }
}

// EnergyFunctional::setDeltaF (EnergyFunctional.cpp:131-156) + EFFrame::takeData / EFPoint::takeData
static void set_delta(EF* E) {
    const int nF = E->nF;
    E->adHTdeltaF.assign((size_t)nF * nF * 6, 0);
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            const int idx = h + t * nF;
            float dh[6], dt[6];
            for (int i = 0; i < 6; ++i) {
                dh[i] = (float)(E->frames[h].state[i] - E->frames[h].state_zero[i]);
                dt[i] = (float)(E->frames[t].state[i] - E->frames[t].state_zero[i]);
            }
            const float* AH = &E->adHostF[(size_t)idx * 36];
            const float* AT = &E->adTargetF[(size_t)idx * 36];
            for (int c = 0; c < 6; ++c) {
                float a = 0, b = 0;
                for (int k = 0; k < 6; ++k) { a += dh[k] * AH[k * 6 + c]; b += dt[k] * AT[k * 6 + c]; }
                E->adHTdeltaF[(size_t)idx * 6 + c] = a + b;
            }
        }
    for (int i = 0; i < 4; ++i) E->cDeltaF[i] = (float)E->value_minus_value_zero[i];
    for (Frame& f : E->frames)
        for (int i = 0; i < 6; ++i) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i] - 0.0; }
    for (Point& p : E->points) p.deltaF = p.idepth - p.idepth_zero;
}

// a9 again (globalFuncs.h:51-65)
static inline void interp33(const float* mat, float x, float y, int width, float out[3]) {
    int ix = (int)x, iy = (int)y;
    float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    for (int k = 0; k < 3; ++k)
        out[k] = ((w11 * bp[3 * (1 + width) + k] + w01 * bp[3 * width + k]) + w10 * bp[3 + k]) + w00 * bp[k];
}

// ---- b1: PointFrameResidual::linearize (Residuals.cpp:60-224) -------------------------------------------
static double linearize(EF* E, Residual& r) {
    r.state_NewEnergyWithOutlier = -1;
    if (r.state_state == OOB) { r.state_NewState = OOB; return r.state_energy; }
    const Point& pt = E->points[r.point];
    const Frame& host = E->frames[r.host];
    const Frame& target = E->frames[r.target];
    const Precalc& pc = E->precalc[(size_t)r.host * E->nF + r.target];
    float energyLeft = 0;
    const float* KRKi = pc.PRE_KRKiTll;
    const float* Kt = pc.PRE_KtTll;
    const float* R0 = pc.PRE_RTll_0;
    const float* t0 = pc.PRE_tTll_0;
    const float* dIl = target.dI.data();
    const float affLL0 = pc.PRE_aff_mode[0], affLL1 = pc.PRE_aff_mode[1];
    const float fxl = E->fxl, fyl = E->fyl, cxl = E->cxl, cyl = E->cyl, fxli = E->fxli, fyli = E->fyli;

    float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x, d_d_y, Ku, Kv;
    {
        if (!r.hasMatcher) { r.state_NewState = OOB; return r.state_energy; }
        // projectPoint(u,v,idepth_zero_scaled,0,0,HCalib,R0,t0,...)  ResidualProjections.h:32-59
        float KliP[3] = {(pt.u + 0 - cxl) * fxli, (pt.v + 0 - cyl) * fyli, 1};
        float ptp[3];
        for (int i = 0; i < 3; ++i) ptp[i] = ((R0[i * 3] * KliP[0] + R0[i * 3 + 1] * KliP[1]) + R0[i * 3 + 2] * KliP[2]) + t0[i] * pt.idepth_zero_scaled;
        float drescale = 1.0f / ptp[2];
        float new_idepth = pt.idepth_zero_scaled * drescale;
        bool ok = true;
        float u = 0, v = 0;
        if (!(drescale > 0)) ok = false;
        else {
            u = ptp[0] * drescale; v = ptp[1] * drescale;
            Ku = u * fxl + cxl; Kv = v * fyl + cyl;
            ok = Ku > 1.1f && Kv > 1.1f && Ku < E->wM3G && Kv < E->hM3G;
        }
        if (!ok) { r.state_NewState = OOB; return r.state_energy; }
        r.centerProjectedTo[0] = Ku; r.centerProjectedTo[1] = Kv; r.centerProjectedTo[2] = new_idepth;

        d_d_x = drescale * (t0[0] - t0[2] * u) * SCALE_IDEPTH * fxl;
        d_d_y = drescale * (t0[1] - t0[2] * v) * SCALE_IDEPTH * fyl;

        d_C_x[2] = drescale * (R0[2 * 3 + 0] * u - R0[0 * 3 + 0]);
        d_C_x[3] = fxl * drescale * (R0[2 * 3 + 1] * u - R0[0 * 3 + 1]) * fyli;
        d_C_x[0] = KliP[0] * d_C_x[2];
        d_C_x[1] = KliP[1] * d_C_x[3];

        d_C_y[2] = fyl * drescale * (R0[2 * 3 + 0] * v - R0[1 * 3 + 0]) * fxli;
        d_C_y[3] = drescale * (R0[2 * 3 + 1] * v - R0[1 * 3 + 1]);
        d_C_y[0] = KliP[0] * d_C_y[2];
        d_C_y[1] = KliP[1] * d_C_y[3];

        d_C_x[0] = (d_C_x[0] + u) * SCALE_F;
        d_C_x[1] *= SCALE_F;
        d_C_x[2] = (d_C_x[2] + 1) * SCALE_C;
        d_C_x[3] *= SCALE_C;

        d_C_y[0] *= SCALE_F;
        d_C_y[1] = (d_C_y[1] + v) * SCALE_F;
        d_C_y[2] *= SCALE_C;
        d_C_y[3] = (d_C_y[3] + 1) * SCALE_C;

        d_xi_x[0] = new_idepth * fxl;
        d_xi_x[1] = 0;
        d_xi_x[2] = -new_idepth * u * fxl;
        d_xi_x[3] = -u * v * fxl;
        d_xi_x[4] = (1 + u * u) * fxl;
        d_xi_x[5] = -v * fxl;

        d_xi_y[0] = 0;
        d_xi_y[1] = new_idepth * fyl;
        d_xi_y[2] = -new_idepth * v * fyl;
        d_xi_y[3] = -(1 + v * v) * fyl;
        d_xi_y[4] = u * v * fyl;
        d_xi_y[5] = u * fyl;
    }
    RawJ& J = r.Jnew;
    for (int i = 0; i < 6; ++i) { J.Jpdxi[0][i] = d_xi_x[i]; J.Jpdxi[1][i] = d_xi_y[i]; }
    for (int i = 0; i < 4; ++i) { J.Jpdc[0][i] = d_C_x[i]; J.Jpdc[1][i] = d_C_y[i]; }
    J.Jpdd[0] = d_d_x; J.Jpdd[1] = d_d_y;

    float wJI2_sum = 0;
    float energyLeft2 = 0.0;
    for (int idx = 0; idx < patternNum; idx++) {
        // projectPoint(u+dx, v+dy, idepth_scaled, KRKi, Kt, Ku2, Kv2)  ResidualProjections.h:20-30
        const float up = pt.u + patternP[idx][0], vp = pt.v + patternP[idx][1];
        float ptp[3];
        for (int i = 0; i < 3; ++i) ptp[i] = ((KRKi[i * 3] * up + KRKi[i * 3 + 1] * vp) + KRKi[i * 3 + 2] * 1.0f) + Kt[i] * pt.idepth_scaled;
        float Ku2 = ptp[0] / ptp[2];
        float Kv2 = ptp[1] / ptp[2];
        if (!(Ku2 > 1.1f && Kv2 > 1.1f && Ku2 < E->wM3G && Kv2 < E->hM3G)) break;
        float hit[3];
        interp33(dIl, Ku2, Kv2, E->w, hit);
        float residual = hit[0] - (float)(affLL0 * pt.color[idx] + affLL1);
        if (!std::isfinite(hit[0])) break;
        float w = sqrtf(setting_outlierTHSumComponent / (setting_outlierTHSumComponent + (hit[1] * hit[1] + hit[2] * hit[2])));
        w = 0.5f * (w + pt.weights[idx]);
        float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
        energyLeft2 += w * w * hw * residual * residual * (2 - hw);
        {
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * w;
            hit[1] *= hw;
            hit[2] *= hw;
            wJI2_sum += hw * hw * (hit[1] * hit[1] + hit[2] * hit[2]);
        }
    }
    float res0 = Ku - (float)r.matcher[0];
    float res1 = Kv - (float)r.matcher[1];
    float nrm = std::sqrt(res0 * res0 + res1 * res1);
    float hw = fabsf(nrm) < setting_huberTH ? 1 : setting_huberTH / fabsf(nrm);
    energyLeft = hw * (res0 * res0 + res1 * res1) * (2 - hw);
    if (hw < 1) hw = sqrtf(hw);
    J.resF[0] = res0 * hw; J.resF[1] = res1 * hw;
    for (int i = 0; i < 6; ++i) { J.Jpdxi[0][i] = J.Jpdxi[0][i] * hw; J.Jpdxi[1][i] = J.Jpdxi[1][i] * hw; }
    for (int i = 0; i < 4; ++i) { J.Jpdc[0][i] = J.Jpdc[0][i] * hw; J.Jpdc[1][i] = J.Jpdc[1][i] * hw; }
    J.Jpdd[0] = J.Jpdd[0] * hw; J.Jpdd[1] = J.Jpdd[1] * hw;

    r.state_NewEnergyWithOutlier = energyLeft2;
    const float th = std::max<float>(host.frameEnergyTH, target.frameEnergyTH);
    if (energyLeft2 > th || wJI2_sum < 2) { energyLeft2 = th; r.state_NewState = OUTLIER; }
    else r.state_NewState = IN;
    r.state_NewEnergy = energyLeft2;
    return energyLeft;
}

// PointFrameResidual::applyRes(true) (Residuals.cpp:252-275) + EFResidual::takeDataF (EnergyFunctionalStructs.cpp:15-25)
static void apply_res(Residual& r) {
    if (r.state_state == OOB) return;
    if (r.state_NewState == IN) {
        r.isActive = true;
        std::swap(r.Jef, r.Jnew);
        for (int i = 0; i < 6; i++) r.JpJdF[i] = r.Jef.Jpdxi[0][i] * r.Jef.Jpdd[0] + r.Jef.Jpdxi[1][i] * r.Jef.Jpdd[1];
        r.JpJdF[6] = r.JpJdF[7] = 0;
    } else {
        r.isActive = false;
    }
    r.state_state = r.state_NewState;
    r.state_energy = r.state_NewEnergy;
}

// ---- b2: AccumulatedTopHessianSSE::addPoint<mode> (AccumulatedTopHessian.cpp:14-112), mode 0 / 1 ---------
static void add_point_top(EF* E, std::vector<AccumulatorApprox>& acc, Point& p, int mode, int& nres) {
    const float* dc = E->cDeltaF;
    const float dd = p.deltaF;
    float bd_acc = 0, Hdd_acc = 0, Hcd_acc[4] = {0, 0, 0, 0};
    for (int ri = p.r0; ri < p.r1; ++ri) {
        Residual& r = E->res[ri];
        if (mode == 0) { if (r.isLinearized || !r.isActive) continue; }
        if (mode == 1) { if (!r.isLinearized || !r.isActive) continue; }
        if (mode == 2) { if (!r.isActive) continue; }   // marginalise: all active residuals (the reference asserts isLinearized, :40)
        const RawJ& rJ = r.Jef;
        const int htIDX = r.host + r.target * E->nF;
        const float* dp = &E->adHTdeltaF[(size_t)htIDX * 6];
        float resApprox[2];
        if (mode == 0) { resApprox[0] = rJ.resF[0]; resApprox[1] = rJ.resF[1]; }
        else if (mode == 2) { resApprox[0] = r.res_toZeroF[0]; resApprox[1] = r.res_toZeroF[1]; }   // :61-62
        else {
            const float dx = dot_small(rJ.Jpdxi[0], dp, 6), dy = dot_small(rJ.Jpdxi[1], dp, 6);
            const float cx = dot_small(rJ.Jpdc[0], dc, 4), cy = dot_small(rJ.Jpdc[1], dc, 4);
            const float Jp_delta_x = dx + cx + rJ.Jpdd[0] * dd;
            const float Jp_delta_y = dy + cy + rJ.Jpdd[1] * dd;
            resApprox[0] = r.res_toZeroF[0] + Jp_delta_x;
            resApprox[1] = r.res_toZeroF[1] + Jp_delta_y;
        }
        const float rr = resApprox[0] * resApprox[0] + resApprox[1] * resApprox[1];
        acc[htIDX].update(rJ.Jpdc[0], rJ.Jpdxi[0], rJ.Jpdc[1], rJ.Jpdxi[1], 1, 0, 1);
        acc[htIDX].updateBotRight(0, 0, 0, 0, 0, rr);
        acc[htIDX].updateTopRight(rJ.Jpdc[0], rJ.Jpdxi[0], rJ.Jpdc[1], rJ.Jpdxi[1], 0, 0, 0, 0, resApprox[0], resApprox[1]);
        bd_acc += resApprox[0] * rJ.Jpdd[0] + resApprox[1] * rJ.Jpdd[1];
        Hdd_acc += rJ.Jpdd[0] * rJ.Jpdd[0] + rJ.Jpdd[1] * rJ.Jpdd[1];
        for (int i = 0; i < 4; ++i) Hcd_acc[i] += rJ.Jpdc[0][i] * rJ.Jpdd[0] + rJ.Jpdc[1][i] * rJ.Jpdd[1];
        nres++;
    }
    if (mode == 0) { p.Hdd_accAF = Hdd_acc; p.bd_accAF = bd_acc; for (int i = 0; i < 4; ++i) p.Hcd_accAF[i] = Hcd_acc[i]; }
    else { p.Hdd_accLF = Hdd_acc; p.bd_accLF = bd_acc; for (int i = 0; i < 4; ++i) p.Hcd_accLF[i] = Hcd_acc[i]; }
    if (mode == 2) { p.Hdd_accAF = 0; p.bd_accAF = 0; for (int i = 0; i < 4; ++i) p.Hcd_accAF[i] = 0; }   // :105-110
}

static inline double* blk(std::vector<double>& H, int n, int r, int c) { return &H[(size_t)r * n + c]; }

// C(6x6) += A(6x6) * M(6x6 stored with leading dim ldm) * B^T(6x6)
static void add_AMBt(std::vector<double>& H, int n, int r0, int c0, const double* A, const double* M, int ldm, const double* B) {
    double T[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * M[k * ldm + j];
            T[i * 6 + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += T[i * 6 + k] * B[j * 6 + k];
            H[(size_t)(r0 + i) * n + c0 + j] += s;
        }
}

// ---- b4: stitchDoubleInternal(tid=-1) + tail of stitchDoubleMT ------------------------------------------
// `extra`: the accumulators of worker threads 1..T-1 (stitchDoubleMT sums the finished per-thread matrices in double,
// AccumulatedTopHessian.h:70-98); empty in the single-thread configuration.
static void stitch_top(EF* E, std::vector<AccumulatorApprox>& acc, std::vector<double>& H, std::vector<double>& b, bool usePrior,
                       std::vector<std::vector<AccumulatorApprox>>* extra = nullptr) {
    const int nF = E->nF, n = CPARS + 6 * nF;
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    for (int k = 0; k < nF * nF; ++k) {
        const int h = k % nF, t = k / nF;
        const int hIdx = CPARS + h * 6, tIdx = CPARS + t * 6, aidx = h + nF * t;
        double accH[13 * 13];
        std::memset(accH, 0, sizeof(accH));
        acc[aidx].finish();
        if (acc[aidx].num != 0)
            for (int r = 0; r < 13; ++r) for (int c = 0; c < 13; ++c) accH[r * 13 + c] += (double)acc[aidx].H[r][c];
        if (extra)
            for (auto& ex : *extra) {
                ex[aidx].finish();
                if (ex[aidx].num == 0) continue;
                for (int r = 0; r < 13; ++r) for (int c = 0; c < 13; ++c) accH[r * 13 + c] += (double)ex[aidx].H[r][c];
            }
        const double* AH = &E->adHost[(size_t)aidx * 36];
        const double* AT = &E->adTarget[(size_t)aidx * 36];
        const double* A66 = &accH[CPARS * 13 + CPARS];
        add_AMBt(H, n, hIdx, hIdx, AH, A66, 13, AH);
        add_AMBt(H, n, tIdx, tIdx, AT, A66, 13, AT);
        add_AMBt(H, n, hIdx, tIdx, AH, A66, 13, AT);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < CPARS; ++j) {
                double sh = 0, st = 0;
                for (int kk = 0; kk < 6; ++kk) { sh += AH[i * 6 + kk] * accH[(CPARS + kk) * 13 + j]; st += AT[i * 6 + kk] * accH[(CPARS + kk) * 13 + j]; }
                H[(size_t)(hIdx + i) * n + j] += sh;
                H[(size_t)(tIdx + i) * n + j] += st;
            }
        for (int i = 0; i < CPARS; ++i) for (int j = 0; j < CPARS; ++j) H[(size_t)i * n + j] += accH[i * 13 + j];
        for (int i = 0; i < 6; ++i) {
            double sh = 0, st = 0;
            for (int kk = 0; kk < 6; ++kk) { sh += AH[i * 6 + kk] * accH[(CPARS + kk) * 13 + (CPARS + 8)]; st += AT[i * 6 + kk] * accH[(CPARS + kk) * 13 + (CPARS + 8)]; }
            b[hIdx + i] += sh;
            b[tIdx + i] += st;
        }
        for (int i = 0; i < CPARS; ++i) b[i] += accH[i * 13 + (CPARS + 8)];
    }
    if (usePrior) {
        for (int i = 0; i < CPARS; ++i) { H[(size_t)i * n + i] += E->cPrior[i]; b[i] += E->cPrior[i] * (double)E->cDeltaF[i]; }
        for (int h = 0; h < nF; ++h)
            for (int i = 0; i < 6; ++i) {
                H[(size_t)(CPARS + h * 6 + i) * n + (CPARS + h * 6 + i)] += E->frames[h].prior[i];
                b[CPARS + h * 6 + i] += E->frames[h].prior[i] * E->frames[h].delta_prior[i];
            }
    }
    // stitchDoubleMT tail (.h:100-113)
    for (int h = 0; h < nF; ++h) {
        const int hIdx = CPARS + h * 6;
        for (int i = 0; i < CPARS; ++i) for (int j = 0; j < 6; ++j) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i];
        for (int t = h + 1; t < nF; ++t) {
            const int tIdx = CPARS + t * 6;
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[(size_t)(hIdx + i) * n + tIdx + j] += H[(size_t)(tIdx + j) * n + hIdx + i];
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[(size_t)(tIdx + i) * n + hIdx + j] = H[(size_t)(hIdx + j) * n + tIdx + i];
        }
    }
}

// ---- b7: AccumulatedSCHessianSSE::addPoint + stitchDoubleInternal(tid=-1) + MT tail ------------------------
struct SCAcc {
    int nF;
    std::vector<AccXX> accE, accEB, accD;
    AccXX accHcc, accbc;
    void setZero(int n) {
        nF = n;
        accE.resize((size_t)n * n); accEB.resize((size_t)n * n); accD.resize((size_t)n * n * n);
        for (auto& a : accE) a.initialize(8, CPARS);
        for (auto& a : accEB) a.initialize(8, 1);
        for (auto& a : accD) a.initialize(8, 8);
        accHcc.initialize(CPARS, CPARS); accbc.initialize(CPARS, 1);
    }
};

static void add_point_sc(EF* E, SCAcc& S, Point& p, bool shiftPriorToZero) {
    int ngoodres = 0;
    for (int ri = p.r0; ri < p.r1; ++ri) if (E->res[ri].isActive) ngoodres++;
    if (ngoodres == 0) { p.HdiF = 0; p.bdSumF = 0; p.idepth_hessian = 0; return; }
    float H = p.Hdd_accAF + p.Hdd_accLF + p.priorF;
    if (H < 1e-10) H = 1e-10;
    p.idepth_hessian = H;
    p.HdiF = 1.0 / H;
    p.bdSumF = p.bd_accAF + p.bd_accLF;
    if (shiftPriorToZero) p.bdSumF += p.priorF * p.deltaF;
    float Hcd[4];
    for (int i = 0; i < 4; ++i) Hcd[i] = p.Hcd_accAF[i] + p.Hcd_accLF[i];
    if (p.isFromSensor) return;
    S.accHcc.update(Hcd, Hcd, p.HdiF);
    S.accbc.updateVec(Hcd, p.bdSumF * p.HdiF);
    const int nF = S.nF, nFrames2 = nF * nF;
    for (int r1i = p.r0; r1i < p.r1; ++r1i) {
        Residual& r1 = E->res[r1i];
        if (!r1.isActive) continue;
        const int r1ht = r1.host + r1.target * nF;
        for (int r2i = p.r0; r2i < p.r1; ++r2i) {
            Residual& r2 = E->res[r2i];
            if (!r2.isActive) continue;
            S.accD[r1ht + r2.target * nFrames2].update(r1.JpJdF, r2.JpJdF, p.HdiF);
        }
        S.accE[r1ht].update(r1.JpJdF, Hcd, p.HdiF);
        S.accEB[r1ht].updateVec(r1.JpJdF, p.HdiF * p.bdSumF);
    }
}

static void add_AMBt6(std::vector<double>& H, int n, int r0, int c0, const double* A, const double* M8, const double* B) {
    add_AMBt(H, n, r0, c0, A, M8, 8, B);
}

static void stitch_sc(EF* E, SCAcc& S, std::vector<double>& H, std::vector<double>& b, std::vector<SCAcc>* extra = nullptr) {
    const int nF = S.nF, n = CPARS + 6 * nF, nframes2 = nF * nF;
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    for (int k = 0; k < nframes2; ++k) {
        const int i = k % nF, j = k / nF;
        const int iIdx = CPARS + i * 6, jIdx = CPARS + j * 6, ijIdx = i + nF * j;
        S.accE[ijIdx].finish(); S.accEB[ijIdx].finish();
        double Hpc[8 * CPARS], bp[8];
        for (int q = 0; q < 8 * CPARS; ++q) Hpc[q] = (double)S.accE[ijIdx].A1m[q];
        for (int q = 0; q < 8; ++q) bp[q] = (double)S.accEB[ijIdx].A1m[q];
        if (extra)
            for (SCAcc& X : *extra) {   // AccumulatedSCHessian.cpp:80-90 (tid loop)
                X.accE[ijIdx].finish(); X.accEB[ijIdx].finish();
                for (int q = 0; q < 8 * CPARS; ++q) Hpc[q] += (double)X.accE[ijIdx].A1m[q];
                for (int q = 0; q < 8; ++q) bp[q] += (double)X.accEB[ijIdx].A1m[q];
            }
        const double* AH = &E->adHost[(size_t)ijIdx * 36];
        const double* AT = &E->adTarget[(size_t)ijIdx * 36];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < CPARS; ++c) {
                double sh = 0, st = 0;
                for (int q = 0; q < 6; ++q) { sh += AH[r * 6 + q] * Hpc[q * CPARS + c]; st += AT[r * 6 + q] * Hpc[q * CPARS + c]; }
                H[(size_t)(iIdx + r) * n + c] += sh;
                H[(size_t)(jIdx + r) * n + c] += st;
            }
            double sh = 0, st = 0;
            for (int q = 0; q < 6; ++q) { sh += AH[r * 6 + q] * bp[q]; st += AT[r * 6 + q] * bp[q]; }
            b[iIdx + r] += sh;
            b[jIdx + r] += st;
        }
        for (int kk = 0; kk < nF; ++kk) {
            const int kIdx = CPARS + kk * 6, ijkIdx = ijIdx + kk * nframes2, ikIdx = i + nF * kk;
            double accDM[64];
            std::memset(accDM, 0, sizeof(accDM));
            S.accD[ijkIdx].finish();
            if (S.accD[ijkIdx].num != 0) for (int q = 0; q < 64; ++q) accDM[q] += (double)S.accD[ijkIdx].A1m[q];
            if (extra)
                for (SCAcc& X : *extra) {
                    X.accD[ijkIdx].finish();
                    if (X.accD[ijkIdx].num != 0) for (int q = 0; q < 64; ++q) accDM[q] += (double)X.accD[ijkIdx].A1m[q];
                }
            const double* AHk = &E->adHost[(size_t)ikIdx * 36];
            const double* ATk = &E->adTarget[(size_t)ikIdx * 36];
            add_AMBt6(H, n, iIdx, iIdx, AH, accDM, AHk);
            add_AMBt6(H, n, jIdx, kIdx, AT, accDM, ATk);
            add_AMBt6(H, n, jIdx, iIdx, AT, accDM, AHk);
            add_AMBt6(H, n, iIdx, kIdx, AH, accDM, ATk);
        }
    }
    S.accHcc.finish(); S.accbc.finish();
    for (int r = 0; r < CPARS; ++r) {
        for (int c = 0; c < CPARS; ++c) H[(size_t)r * n + c] += (double)S.accHcc.A1m[r * CPARS + c];
        b[r] += (double)S.accbc.A1m[r];
    }
    if (extra)
        for (SCAcc& X : *extra) {
            X.accHcc.finish(); X.accbc.finish();
            for (int r = 0; r < CPARS; ++r) {
                for (int c = 0; c < CPARS; ++c) H[(size_t)r * n + c] += (double)X.accHcc.A1m[r * CPARS + c];
                b[r] += (double)X.accbc.A1m[r];
            }
        }
    for (int h = 0; h < nF; ++h) {
        const int hIdx = CPARS + h * 6;
        for (int i = 0; i < CPARS; ++i) for (int j = 0; j < 6; ++j) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i];
    }
}

// One-sided Jacobi SVD of an m x k (k <= m) matrix: A = U diag(s) V^T  (stands in for Eigen::JacobiSVD in
// EnergyFunctional::orthogonalize, EnergyFunctional.cpp:615-648).
static void jacobi_svd(int m, int k, std::vector<double> A, std::vector<double>& U, std::vector<double>& s, std::vector<double>& V) {
    V.assign((size_t)k * k, 0);
    for (int i = 0; i < k; ++i) V[(size_t)i * k + i] = 1;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                double a = 0, bb = 0, c = 0;
                for (int i = 0; i < m; ++i) { a += A[(size_t)i * k + p] * A[(size_t)i * k + p]; bb += A[(size_t)i * k + q] * A[(size_t)i * k + q]; c += A[(size_t)i * k + p] * A[(size_t)i * k + q]; }
                off = std::max(off, std::fabs(c) / std::sqrt(std::max(a * bb, 1e-300)));
                if (std::fabs(c) < 1e-300) continue;
                const double zeta = (bb - a) / (2 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) {
                    const double x = A[(size_t)i * k + p], y = A[(size_t)i * k + q];
                    A[(size_t)i * k + p] = cs * x - sn * y; A[(size_t)i * k + q] = sn * x + cs * y;
                }
                for (int i = 0; i < k; ++i) {
                    const double x = V[(size_t)i * k + p], y = V[(size_t)i * k + q];
                    V[(size_t)i * k + p] = cs * x - sn * y; V[(size_t)i * k + q] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    s.assign(k, 0); U.assign((size_t)m * k, 0);
    for (int j = 0; j < k; ++j) {
        double nn = 0;
        for (int i = 0; i < m; ++i) nn += A[(size_t)i * k + j] * A[(size_t)i * k + j];
        s[j] = std::sqrt(nn);
        for (int i = 0; i < m; ++i) U[(size_t)i * k + j] = s[j] > 0 ? A[(size_t)i * k + j] / s[j] : 0;
    }
}

// EnergyFunctional::orthogonalize(VecX* b, 0)  (EnergyFunctional.cpp:615-648), setting_solverModeDelta=1e-5
static void orthogonalize_x(EF* E, std::vector<double>& x) {
    const int n = (int)x.size(), k = (int)E->nullspaces.size();
    if (k == 0) return;
    std::vector<double> N((size_t)n * k);
    for (int j = 0; j < k; ++j) {
        double nn = 0;
        for (int i = 0; i < n; ++i) nn += E->nullspaces[j][i] * E->nullspaces[j][i];
        nn = std::sqrt(nn);
        for (int i = 0; i < n; ++i) N[(size_t)i * k + j] = E->nullspaces[j][i] / nn;
    }
    std::vector<double> U, s, V;
    jacobi_svd(n, k, N, U, s, V);
    double maxSv = 0;
    for (double v : s) maxSv = std::max(maxSv, v);
    for (double& v : s) v = (v > 1e-5 * maxSv) ? 1.0 / v : 0;
    // Npi = U diag(s) V^T ; NNpiT = N Npi^T ; NNpiTS = 0.5 (NNpiT + NNpiT^T)
    std::vector<double> Npi((size_t)n * k, 0), M((size_t)n * n, 0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < k; ++j) { double a = 0; for (int q = 0; q < k; ++q) a += U[(size_t)i * k + q] * s[q] * V[(size_t)j * k + q]; Npi[(size_t)i * k + j] = a; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double a = 0; for (int q = 0; q < k; ++q) a += N[(size_t)i * k + q] * Npi[(size_t)j * k + q]; M[(size_t)i * n + j] = a; }
    std::vector<double> y(n, 0);
    for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < n; ++j) a += 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]) * x[j]; y[i] = a; }
    for (int i = 0; i < n; ++i) x[i] -= y[i];
}

// ---- b6: EnergyFunctional::solveSystemF (EnergyFunctional.cpp:650-759), default solver mode -------------
static void solve_system(EF* E, int iteration, double lambda) {
    const int nF = E->nF, n = CPARS + 6 * nF;
    // accumulateAF_MT (MT=false)
    E->accA.resize((size_t)nF * nF);
    for (auto& a : E->accA) a.initialize();
    E->resInA = 0;
    const int T = E->nThreads, nPts = (int)E->points.size();
    SCAcc S;
    S.setZero(nF);
    if (T <= 1) {
    for (Point& p : E->points) add_point_top(E, E->accA, p, 0, E->resInA);
    stitch_top(E, E->accA, E->HA, E->bA, true);
    // accumulateLF_MT: result discarded, but it (re)sets p->*_accLF
    E->accL.resize((size_t)nF * nF);
    for (auto& a : E->accL) a.initialize();
    E->resInL = 0;
    for (Point& p : E->points) add_point_top(E, E->accL, p, 1, E->resInL);
    // accumulateSCF_MT
    for (Point& p : E->points) add_point_sc(E, S, p, true);
    stitch_sc(E, S, E->Hsc, E->bsc);
    } else {
        // multiThreading = true (settings.cpp:164 off by default): the *_MT calls hand blocks of 50 points to the IndexThreadReduce
        // workers, each with its own accumulator set (EnergyFunctional.cpp:158-219); here the blocks go to the T workers round-robin
        // (static schedule) so that a run is reproducible.  Timing baseline only -- sums differ from T = 1 in the last float bits.
        std::vector<std::vector<AccumulatorApprox>> exA(T - 1, std::vector<AccumulatorApprox>((size_t)nF * nF)), accLT(T, std::vector<AccumulatorApprox>((size_t)nF * nF));
        for (auto& v : exA) for (auto& a : v) a.initialize();
        for (auto& v : accLT) for (auto& a : v) a.initialize();
        std::vector<SCAcc> exS(T - 1);
        for (SCAcc& X : exS) X.setZero(nF);
        std::vector<int> nA(T, 0), nL(T, 0);
#pragma omp parallel num_threads(T)
        {
            const int tid = omp_get_thread_num();
            std::vector<AccumulatorApprox>& mineA = tid == 0 ? E->accA : exA[tid - 1];
#pragma omp for schedule(static, 50)
            for (int i = 0; i < nPts; ++i) add_point_top(E, mineA, E->points[i], 0, nA[tid]);
#pragma omp for schedule(static, 50)
            for (int i = 0; i < nPts; ++i) add_point_top(E, accLT[tid], E->points[i], 1, nL[tid]);
            SCAcc& mineS = tid == 0 ? S : exS[tid - 1];
#pragma omp for schedule(static, 50)
            for (int i = 0; i < nPts; ++i) add_point_sc(E, mineS, E->points[i], true);
        }
        for (int t = 0; t < T; ++t) E->resInA += nA[t];
        stitch_top(E, E->accA, E->HA, E->bA, true, &exA);
        stitch_sc(E, S, E->Hsc, E->bsc, &exS);
        // fold the workers' (finished) accumulators into set 0 so that the getters below see the totals
        for (auto& v : exA) for (int k = 0; k < nF * nF; ++k) for (int r = 0; r < 13; ++r) for (int c = 0; c < 13; ++c) E->accA[k].H[r][c] += v[k].H[r][c];
        for (SCAcc& X : exS) {
            for (int k = 0; k < nF * nF; ++k) { for (int q = 0; q < 32; ++q) S.accE[k].A1m[q] += X.accE[k].A1m[q]; for (int q = 0; q < 8; ++q) S.accEB[k].A1m[q] += X.accEB[k].A1m[q]; }
            for (int k = 0; k < nF * nF * nF; ++k) for (int q = 0; q < 64; ++q) S.accD[k].A1m[q] += X.accD[k].A1m[q];
            for (int q = 0; q < 16; ++q) S.accHcc.A1m[q] += X.accHcc.A1m[q];
            for (int q = 0; q < 4; ++q) S.accbc.A1m[q] += X.accbc.A1m[q];
        }
    }
    {
        E->scE.assign((size_t)nF * nF * 32, 0); E->scEB.assign((size_t)nF * nF * 8, 0); E->scD.assign((size_t)nF * nF * nF * 64, 0);
        for (int k = 0; k < nF * nF; ++k) { std::memcpy(&E->scE[(size_t)k * 32], S.accE[k].A1m, 32 * 4); std::memcpy(&E->scEB[(size_t)k * 8], S.accEB[k].A1m, 8 * 4); }
        for (int k = 0; k < nF * nF * nF; ++k) std::memcpy(&E->scD[(size_t)k * 64], S.accD[k].A1m, 64 * 4);
        E->scHcc.assign(S.accHcc.A1m, S.accHcc.A1m + 16); E->scbc.assign(S.accbc.A1m, S.accbc.A1m + 4);
    }
    // bM_top = bM + HM * delta
    std::vector<double> d(n), bM_top(n);
    for (int i = 0; i < CPARS; ++i) d[i] = (double)E->cDeltaF[i];
    for (int h = 0; h < nF; ++h) for (int i = 0; i < 6; ++i) d[CPARS + 6 * h + i] = E->frames[h].delta[i];
    for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < n; ++j) a += E->HM[(size_t)i * n + j] * d[j]; bM_top[i] = E->bM[i] + a; }
    E->HFinal.assign((size_t)n * n, 0); E->bFinal.assign(n, 0);
    for (size_t i = 0; i < (size_t)n * n; ++i) E->HFinal[i] = E->HA[i] + E->HM[i] - E->Hsc[i];
    for (int i = 0; i < n; ++i) E->bFinal[i] = E->bA[i] + bM_top[i] - E->bsc[i];
    std::vector<double> Hd = E->HFinal;   // lastHS keeps the undamped matrix
    for (int i = 0; i < n; ++i) Hd[(size_t)i * n + i] *= (1 + lambda);
    std::vector<double> SVecI(n), Hs((size_t)n * n), bs(n), xs(n);
    for (int i = 0; i < n; ++i) SVecI[i] = 1.0 / std::sqrt(Hd[(size_t)i * n + i] + 10);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) Hs[(size_t)i * n + j] = SVecI[i] * Hd[(size_t)i * n + j] * SVecI[j]; bs[i] = SVecI[i] * E->bFinal[i]; }
    ldlt_solve(n, Hs.data(), bs.data(), xs.data());
    E->lastX.assign(n, 0);
    for (int i = 0; i < n; ++i) E->lastX[i] = SVecI[i] * xs[i];
    if (iteration >= 2) orthogonalize_x(E, E->lastX);  // SOLVER_ORTHOGONALIZE_X_LATER (settings.cpp:34)
    // resubstituteF_MT (:221-282)
    std::vector<float> xF(n);
    for (int i = 0; i < n; ++i) xF[i] = (float)E->lastX[i];
    for (int i = 0; i < CPARS; ++i) E->calibStep[i] = -E->lastX[i];
    std::vector<float> xAd((size_t)nF * nF * 6);
    for (int h = 0; h < nF; ++h) {
        for (int i = 0; i < 6; ++i) E->frames[h].step[i] = -E->lastX[CPARS + 6 * h + i];
        for (int i = 6; i < 10; ++i) E->frames[h].step[i] = 0;
        for (int t = 0; t < nF; ++t) {
            const float* AH = &E->adHostF[(size_t)(h + nF * t) * 36];
            const float* AT = &E->adTargetF[(size_t)(h + nF * t) * 36];
            for (int c = 0; c < 6; ++c) {
                float a = 0, bb = 0;
                for (int k = 0; k < 6; ++k) { a += xF[CPARS + 6 * h + k] * AH[k * 6 + c]; bb += xF[CPARS + 6 * t + k] * AT[k * 6 + c]; }
                xAd[(size_t)(nF * h + t) * 6 + c] = a + bb;
            }
        }
    }
#pragma omp parallel for schedule(static) num_threads(T) if (T > 1)
    for (int pi = 0; pi < nPts; ++pi) {
        Point& p = E->points[pi];
        int ngoodres = 0;
        for (int ri = p.r0; ri < p.r1; ++ri) if (E->res[ri].isActive) ngoodres++;
        if (ngoodres == 0) { p.step = 0; continue; }
        float b = p.bdSumF;
        b -= dot_small(xF.data(), p.Hcd_accAF, 4);
        for (int ri = p.r0; ri < p.r1; ++ri) {
            const Residual& r = E->res[ri];
            if (!r.isActive) continue;
            const float* xa = &xAd[(size_t)(r.host * nF + r.target) * 6];
            b -= dot_small(xa, r.JpJdF, 6);
        }
        p.step = p.isFromSensor ? 0 : -b * p.HdiF;
    }
}


// CalibHessian::setValue (HessianBlocks.h:302-316)
static void calib_set_value(EF* E, const double* v) {
    for (int i = 0; i < 4; ++i) E->value[i] = v[i];
    E->value_scaled[0] = SCALE_F * v[0]; E->value_scaled[1] = SCALE_F * v[1];
    E->value_scaled[2] = SCALE_C * v[2]; E->value_scaled[3] = SCALE_C * v[3];
    calib_update(E);
    for (int i = 0; i < 4; ++i) E->value_minus_value_zero[i] = E->value[i] - E->value_zero[i];
}

// EnergyFunctional::calcLEnergyF_MT (:333-351) + calcLEnergyPt (:297-331)
static double calc_L_energy(EF* E) {
    double En = 0;
    for (const Frame& f : E->frames) for (int i = 0; i < 6; ++i) En += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
    { float a = 0; for (int i = 0; i < 4; ++i) a += E->cDeltaF[i] * E->cPriorF[i] * E->cDeltaF[i]; En += a; }
    Accumulator11 acc;   // float, tiered: residual terms enter without a shift check, every point's prior term with one (:322,:325)
    acc.initialize();
    for (const Point& p : E->points) {
        for (int ri = p.r0; ri < p.r1; ++ri) {
            const Residual& r = E->res[ri];
            if (!r.isLinearized || !r.isActive) continue;
            const float* dp = &E->adHTdeltaF[(size_t)(r.host + E->nF * r.target) * 6];
            const float dx = dot_small(r.Jef.Jpdxi[0], dp, 6), dy = dot_small(r.Jef.Jpdxi[1], dp, 6);
            const float cx = dot_small(r.Jef.Jpdc[0], E->cDeltaF, 4), cy = dot_small(r.Jef.Jpdc[1], E->cDeltaF, 4);
            const float jx = dx + cx + r.Jef.Jpdd[0] * p.deltaF, jy = dy + cy + r.Jef.Jpdd[1] * p.deltaF;
            acc.updateSingleNoShift((r.res_toZeroF[0] * jx + r.res_toZeroF[1] * jy) + (jx * r.res_toZeroF[0] + jy * r.res_toZeroF[1]) + (jx * jx + jy * jy));
        }
        acc.updateSingle(p.deltaF * p.deltaF * p.priorF);
    }
    acc.finish();
    return En + acc.A;
}
// EnergyFunctional::calcMEnergyF (:284-295)
static double calc_M_energy(EF* E) {
    const int n = CPARS + 6 * E->nF;
    std::vector<double> d(n);
    for (int i = 0; i < CPARS; ++i) d[i] = (double)E->cDeltaF[i];
    for (int h = 0; h < E->nF; ++h) for (int i = 0; i < 6; ++i) d[CPARS + 6 * h + i] = E->frames[h].delta[i];
    double s = 0;
    for (int i = 0; i < n; ++i) { double a = 2 * E->bM[i]; for (int j = 0; j < n; ++j) a += E->HM[(size_t)i * n + j] * d[j]; s += d[i] * a; }
    return s;
}
// FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:63-97): after every linearizeAll the outlier threshold of the NEWEST
// key-frame becomes a function of the 70th percentile of state_NewEnergyWithOutlier over the active (= non-linearised) residuals
// that target it; the next linearize classifies IN / OUTLIER with it (Residuals.cpp:212-214).
static void set_new_frame_energy_th(EF* E) {
    std::vector<float> allResVec;                      // FullSystem.h:299, a vector<float>: the double energies are narrowed on push_back
    allResVec.reserve(E->res.size());
    const int newFrame = E->nF - 1;                    // frameHessians.back()
    for (const Residual& r : E->res)
        if (!r.isLinearized && r.state_NewEnergyWithOutlier >= 0 && r.target == newFrame) allResVec.push_back((float)r.state_NewEnergyWithOutlier);
    Frame& nf = E->frames[newFrame];
    if (allResVec.size() == 0) { nf.frameEnergyTH = 12 * 12 * patternNum; return; }
    const int nthIdx = setting_frameEnergyTHN * allResVec.size();   // float * size_t -> float product, truncated (:85)
    std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
    const float nthElement = sqrtf(allResVec[nthIdx]);
    nf.frameEnergyTH = nthElement * setting_frameEnergyTHFacMedian;
    nf.frameEnergyTH = 26.0f * setting_frameEnergyTHConstWeight + nf.frameEnergyTH * (1 - setting_frameEnergyTHConstWeight);
    nf.frameEnergyTH = nf.frameEnergyTH * nf.frameEnergyTH;
    nf.frameEnergyTH *= setting_overallEnergyTHWeight * setting_overallEnergyTHWeight;
}

static double linearize_all_energy(EF* E);
// FullSystem::linearizeAll(false) (FullSystemOptimize.cpp:99-123): the reductor over activeResiduals, then setNewFrameEnergyTH
static double linearize_all(EF* E) {
    const double s = linearize_all_energy(E);
    set_new_frame_energy_th(E);
    return s;
}
static double linearize_all_energy(EF* E) {
    double s = 0;
    const int T = E->nThreads;
    if (T <= 1) {
        for (Residual& r : E->res) if (!r.isLinearized) s += linearize(E, r);
        return s;
    }
    // linearizeAll_Reductor over T contiguous blocks (treadReduce.reduce(.., 0, n, 0), FullSystemOptimize.cpp:120-129); per-worker
    // energies added in worker order
    const int nR = (int)E->res.size();
    std::vector<double> part(T, 0.0);
#pragma omp parallel num_threads(T)
    {
        const int tid = omp_get_thread_num();
        double a = 0;
#pragma omp for schedule(static)
        for (int i = 0; i < nR; ++i) if (!E->res[i].isLinearized) a += linearize(E, E->res[i]);
        part[tid] = a;
    }
    for (int t = 0; t < T; ++t) s += part[t];
    return s;
}

// FullSystem::optimize loop (FullSystemOptimize.cpp:344-458): backupState / solveSystem / doStepFromBackup /
// linearizeAll / accept-reject.  trace rows: {iteration, lambda, accepted, E_new, EL_new, EM_new, canbreak, x[n]...}
static int optimize(EF* E, int mnumOptIts, double* trace, int trace_stride, int trace_cap) {
    const int nF = E->nF, n = CPARS + 6 * nF;
    if (nF < 2) return 0;
    if (!E->fixedIts && nF < 3) mnumOptIts = 100;
    if (!E->fixedIts && nF < 4) mnumOptIts = 75;
    for (Residual& r : E->res) if (!r.isLinearized) { r.state_NewEnergy = r.state_energy = 0; r.state_NewState = OUTLIER; r.state_state = IN; }  // resetOOB
    double lastEnergy = linearize_all(E);
    double lastEnergyL = calc_L_energy(E), lastEnergyM = calc_M_energy(E);
    for (Residual& r : E->res) if (!r.isLinearized) apply_res(r);
    double lambda = 1e-1;
    const float stepsize = 1;
    const float thOpt = 1.2f;  // setting_thOptIterations
    int it = 0;
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
        // backupState (non-momentum branch :297-306)
        for (int i = 0; i < 4; ++i) E->value_backup[i] = E->value[i];
        for (Frame& f : E->frames) for (int i = 0; i < 10; ++i) f.state_backup[i] = f.state[i];
        for (Point& p : E->points) p.idepth_backup = p.idepth;
        solve_system(E, iteration, lambda);
        // doStepFromBackup(stepsize x5)  :212-262
        float sumT = 0, sumR = 0, sumID = 0, numID = 0, sumNID = 0;
        double v[4];
        for (int i = 0; i < 4; ++i) v[i] = E->value_backup[i] + stepsize * E->calibStep[i];
        calib_set_value(E, v);
        for (Frame& f : E->frames) {
            double st[10];
            for (int i = 0; i < 10; ++i) st[i] = f.state_backup[i] + (double)stepsize * f.step[i];
            frame_set_state(f, st);
            for (int i = 0; i < 3; ++i) sumT += (float)(f.step[i] * f.step[i]);
            for (int i = 3; i < 6; ++i) sumR += (float)(f.step[i] * f.step[i]);
        }
        for (Point& p : E->points) {
            const float nid = p.idepth_backup + stepsize * p.step;
            p.idepth = nid; p.idepth_scaled = SCALE_IDEPTH * nid;
            sumID += p.step * p.step;
            sumNID += fabsf(p.idepth_backup);
            numID++;
            p.idepth_zero = nid; p.idepth_zero_scaled = SCALE_IDEPTH * nid;
        }
        sumR /= nF; sumT /= nF; sumID /= numID; sumNID /= numID;
        set_precalc(E); set_delta(E);
        const bool canbreak = sqrtf(sumR) < 0.00005 * thOpt && sqrtf(sumT) * sumNID < 0.00005 * thOpt;
        const double newEnergy = linearize_all(E);
        const double newEnergyL = calc_L_energy(E), newEnergyM = calc_M_energy(E);
        const bool accept = newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM;
        if (trace && iteration < trace_cap) {
            double* tr = trace + (size_t)iteration * trace_stride;
            tr[0] = iteration; tr[1] = lambda; tr[2] = accept; tr[3] = newEnergy; tr[4] = newEnergyL; tr[5] = newEnergyM; tr[6] = canbreak;
            for (int i = 0; i < n && 7 + i < trace_stride; ++i) tr[7 + i] = E->lastX[i];
            if (7 + n < trace_stride) tr[7 + n] = E->frames[nF - 1].frameEnergyTH;   // as set by the trial linearizeAll
        }
        it = iteration + 1;
        if (accept) {
            for (Residual& r : E->res) if (!r.isLinearized) apply_res(r);
            lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
            lambda *= 0.25;
        } else {
            // loadSateBackup :264-282
            calib_set_value(E, E->value_backup);
            for (Frame& f : E->frames) frame_set_state(f, f.state_backup);
            for (Point& p : E->points) {
                p.idepth = p.idepth_backup; p.idepth_scaled = SCALE_IDEPTH * p.idepth_backup;
                p.idepth_zero = p.idepth_backup; p.idepth_zero_scaled = SCALE_IDEPTH * p.idepth_backup;
            }
            set_precalc(E); set_delta(E);
            lastEnergy = linearize_all(E);
            lastEnergyL = calc_L_energy(E); lastEnergyM = calc_M_energy(E);
            lambda *= 1e2;
        }
        if (!E->fixedIts && canbreak && iteration >= 1) break;   // setting_minOptIterations = 1
    }
    return it;
}

// ---- tail of FullSystem::optimize (FullSystemOptimize.cpp:460-470): the newest frame's linearisation point moves to its optimised
// pose (setEvalPT with a zero state except the affine part), adjoints / precalc are rebuilt, then linearizeAll(true):
// linearize + applyRes(true) per active residual, the isNew bookkeeping of its point (:34-47; isNew is never cleared by the reference),
// setNewFrameEnergyTH, and every residual that is not active afterwards is dropped (:136-155).
// relbs_max[nP]: max over the point's surviving residuals of relBS (caller: maxRelBaseline = max(maxRelBaseline, relbs_max));
// ngood_inc[nP]: numGoodResiduals increments; removed[nR] (input order): 1 = in toRemove.  Returns lastEnergy[0].
static double optimize_finish(EF* E, float* relbs_max, int* ngood_inc, uint8_t* removed) {
    Frame& nf = E->frames[E->nF - 1];
    double newStateZero[10] = {0, 0, 0, 0, 0, 0, nf.state[6], nf.state[7], 0, 0};
    nf.evalPT = nf.PRE_worldToCam;                      // setEvalPT (HessianBlocks.h:170-176): evalPT, setState, setStateZero
    frame_set_state(nf, newStateZero);
    for (int i = 0; i < 10; ++i) nf.state_zero[i] = newStateZero[i];
    set_adjoints(E);
    set_precalc(E); set_delta(E);
    double lastEnergyP = 0;
    for (size_t pi = 0; pi < E->points.size(); ++pi) { if (relbs_max) relbs_max[pi] = 0; if (ngood_inc) ngood_inc[pi] = 0; }
    for (size_t ri = 0; ri < E->res.size(); ++ri) {
        Residual& r = E->res[ri];
        if (removed) removed[ri] = 0;
        if (r.isLinearized) continue;                   // not in activeResiduals
        lastEnergyP += linearize(E, r);
        apply_res(r);
        if (r.isActive) {
            const Point& p = E->points[r.point];
            const Precalc& pc = E->precalc[(size_t)r.host * E->nF + r.target];
            float v3[3] = {p.u, p.v, 1}, inf[3];
            mat3f_vec(pc.PRE_KRKiTll, v3, inf);                                  // projected point assuming infinite depth
            const float ptp[3] = {inf[0] + pc.PRE_KtTll[0] * p.idepth_scaled, inf[1] + pc.PRE_KtTll[1] * p.idepth_scaled,
                                  inf[2] + pc.PRE_KtTll[2] * p.idepth_scaled};   // with real depth
            const float dx = inf[0] / inf[2] - ptp[0] / ptp[2], dy = inf[1] / inf[2] - ptp[1] / ptp[2];
            const float relBS = 0.01 * sqrtf(dx * dx + dy * dy);                 // 0.01 (double) * float norm -> float
            if (relbs_max && relBS > relbs_max[r.point]) relbs_max[r.point] = relBS;
            if (ngood_inc) ngood_inc[r.point]++;
        } else if (removed) removed[ri] = 1;
    }
    set_new_frame_energy_th(E);
    for (size_t ri = 0; ri < E->res.size(); ++ri) {     // ef->dropResidual + deleteOut: the residual no longer exists
        Residual& r = E->res[ri];
        if (!r.isLinearized && !r.isActive) { r.isLinearized = true; r.dropped = true; }
    }
    return lastEnergyP;
}

// ---- EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:45-55) for the active residuals of the flagged points -------------
static void fix_linearization(EF* E, const uint8_t* mask) {
    for (size_t pi = 0; pi < E->points.size(); ++pi) {
        if (!mask[pi]) continue;
        Point& p = E->points[pi];
        for (int ri = p.r0; ri < p.r1; ++ri) {
            Residual& r = E->res[ri];
            if (!r.isActive) continue;
            const float* dp = &E->adHTdeltaF[(size_t)(r.host + E->nF * r.target) * 6];
            const float dx = dot_small(r.Jef.Jpdxi[0], dp, 6), dy = dot_small(r.Jef.Jpdxi[1], dp, 6);
            const float cx = dot_small(r.Jef.Jpdc[0], E->cDeltaF, 4), cy = dot_small(r.Jef.Jpdc[1], E->cDeltaF, 4);
            const float Jp_delta_x = dx + cx + r.Jef.Jpdd[0] * p.deltaF;
            const float Jp_delta_y = dy + cy + r.Jef.Jpdd[1] * p.deltaF;
            r.res_toZeroF[0] = r.Jef.resF[0] - Jp_delta_x;
            r.res_toZeroF[1] = r.Jef.resF[1] - Jp_delta_y;
            r.isLinearized = true;
        }
    }
}

// ---- EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:514-576), setting_solverMode = SOLVER_ORTHOGONALIZE_X_LATER (no
// ORTHOGONALIZE_POINTMARG / _FULL branch).  marg[p]: stateFlag == PS_MARGINALIZE; drop[p]: PS_DROP (dropPointsF, :578-597).  Removed
// points keep their slots but all their residuals become inactive + linearised, which takes them out of every later loop.
static const float setting_idepthFixPriorMargFac = 600 * 600;   // settings.cpp:22
static const float setting_margWeightFac = 0.5f * 0.5f;         // settings.cpp:71
static void marginalize_points(EF* E, const uint8_t* marg, const uint8_t* drop) {
    const int nF = E->nF, n = CPARS + 6 * nF;
    std::vector<AccumulatorApprox> acc((size_t)nF * nF);
    for (auto& a : acc) a.initialize();
    SCAcc S;
    S.setZero(nF);
    int nres = 0;
    for (size_t pi = 0; pi < E->points.size(); ++pi) {
        if (!marg[pi]) continue;
        Point& p = E->points[pi];
        p.priorF *= setting_idepthFixPriorMargFac;
        add_point_top(E, acc, p, 2, nres);
        add_point_sc(E, S, p, false);
    }
    std::vector<double> M, Mb, Msc, Mbsc;
    stitch_top(E, acc, M, Mb, false);
    stitch_sc(E, S, Msc, Mbsc);
    E->resInM += nres;
    if ((int)E->HM.size() != n * n) { E->HM.assign((size_t)n * n, 0); E->bM.assign(n, 0); }
    for (size_t i = 0; i < (size_t)n * n; ++i) E->HM[i] += (double)setting_margWeightFac * (M[i] - Msc[i]);
    for (int i = 0; i < n; ++i) E->bM[i] += (double)setting_margWeightFac * (Mb[i] - Mbsc[i]);
    for (size_t pi = 0; pi < E->points.size(); ++pi) {
        if (!marg[pi] && !(drop && drop[pi])) continue;
        Point& p = E->points[pi];
        for (int ri = p.r0; ri < p.r1; ++ri) { E->res[ri].isActive = false; E->res[ri].isLinearized = true; }
    }
}

// ---- EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:434-512), the algebra on HM / bM: frame idx goes to the end, its prior is
// added, the Schur complement on the (preconditioned) last 6x6 block is taken.  Outputs are (n-6) x (n-6) and n-6.
static void inverse6(const double* A, double* Ainv) {   // Gauss-Jordan with partial pivoting (stands in for Eigen's Mat66::inverse)
    double a[6][12];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j); }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 12; ++j) std::swap(a[c][j], a[piv][j]);
        const double d = a[c][c];
        for (int j = 0; j < 12; ++j) a[c][j] /= d;
        for (int r = 0; r < 6; ++r) if (r != c) { const double f = a[r][c]; if (f != 0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = a[i][6 + j];
}
static void marginalize_frame(const EF* E, int idx, double* HM_out, double* bM_out) {
    const int nF = E->nF, odim = CPARS + 6 * nF, ndim = odim - 6;
    std::vector<double> H = E->HM, b = E->bM;
    if ((int)H.size() != odim * odim) { H.assign((size_t)odim * odim, 0); b.assign(odim, 0); }
    // permutation: frame idx to the end (:446-466)
    std::vector<int> perm;
    for (int i = 0; i < odim; ++i) if (i < CPARS + 6 * idx || i >= CPARS + 6 * idx + 6) perm.push_back(i);
    for (int i = 0; i < 6; ++i) perm.push_back(CPARS + 6 * idx + i);
    std::vector<double> Hp((size_t)odim * odim), bp(odim);
    for (int i = 0; i < odim; ++i) { bp[i] = b[perm[i]]; for (int j = 0; j < odim; ++j) Hp[(size_t)i * odim + j] = H[(size_t)perm[i] * odim + perm[j]]; }
    const Frame& f = E->frames[idx];
    for (int i = 0; i < 6; ++i) { Hp[(size_t)(ndim + i) * odim + ndim + i] += f.prior[i]; bp[ndim + i] += f.prior[i] * f.delta_prior[i]; }   // :468-469
    std::vector<double> SVec(odim), SVecI(odim);
    for (int i = 0; i < odim; ++i) { SVec[i] = std::sqrt(std::fabs(Hp[(size_t)i * odim + i]) + 10); SVecI[i] = 1.0 / SVec[i]; }
    std::vector<double> Hs((size_t)odim * odim), bs(odim);
    for (int i = 0; i < odim; ++i) { bs[i] = SVecI[i] * bp[i]; for (int j = 0; j < odim; ++j) Hs[(size_t)i * odim + j] = SVecI[i] * Hp[(size_t)i * odim + j] * SVecI[j]; }
    double hp[36], hpi[36];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) hp[i * 6 + j] = Hs[(size_t)(ndim + i) * odim + ndim + j];
    for (int i = 0; i < 36; ++i) hp[i] = 0.5f * (hp[i] + hp[i]);          // `hpi = 0.5f*(hpi+hpi)` :478 (a no-op kept as written)
    inverse6(hp, hpi);
    for (int i = 0; i < 36; ++i) hpi[i] = 0.5f * (hpi[i] + hpi[i]);
    // bli = bottomLeft^T * hpi  (ndim x 6); H_tl -= bli * bottomLeft ; b_head -= bli * b_tail
    std::vector<double> bli((size_t)ndim * 6);
    for (int r = 0; r < ndim; ++r) for (int c = 0; c < 6; ++c) { double a = 0; for (int k = 0; k < 6; ++k) a += Hs[(size_t)(ndim + k) * odim + r] * hpi[k * 6 + c]; bli[(size_t)r * 6 + c] = a; }
    for (int r = 0; r < ndim; ++r) {
        for (int c = 0; c < ndim; ++c) { double a = 0; for (int k = 0; k < 6; ++k) a += bli[(size_t)r * 6 + k] * Hs[(size_t)(ndim + k) * odim + c]; Hs[(size_t)r * odim + c] -= a; }
        double a = 0; for (int k = 0; k < 6; ++k) a += bli[(size_t)r * 6 + k] * bs[ndim + k];
        bs[r] -= a;
    }
    // unscale, symmetrise (:489-493)
    for (int i = 0; i < odim; ++i) { bs[i] = SVec[i] * bs[i]; for (int j = 0; j < odim; ++j) Hs[(size_t)i * odim + j] = SVec[i] * Hs[(size_t)i * odim + j] * SVec[j]; }
    for (int r = 0; r < ndim; ++r) { bM_out[r] = bs[r]; for (int c = 0; c < ndim; ++c) HM_out[(size_t)r * ndim + c] = 0.5 * (Hs[(size_t)r * odim + c] + Hs[(size_t)c * odim + r]); }
}

}  // namespace orcb

using namespace orcb;
// ---- 8f-4 (part 2): ImmaturePoint::linearizeResidual (ImmaturePoint.cpp:410-477) and FullSystem::optimizeImmaturePoint
// (FullSystemOptPoint.cpp:18-185).  The tail of optimizeImmaturePoint that allocates the PointHessian / PointFrameResidual objects
// stays with the caller; this returns what it needs: result code (0 = "return 0" not well-constrained, -1 = the (PointHessian*)-1
// outlier code, 1 = activate), the optimised inverse depth and the per-target residual states.
struct ImmRes { int state_state, state_NewState; double state_energy, state_NewEnergy; int target; };
struct ImmPt { int host; float u, v, idepth_min, idepth_max, energyTH; float color[8], weights[8]; bool isFromSensor; };
static const float setting_minIdepthH_act = 100;           // settings.cpp:41
static const int setting_GNItsOnPointActivation = 3;       // settings.cpp:133
static const float setting_huberTH_imm = 6;                // settings.cpp:101

static double imm_linearize_residual(const EF* E, const ImmPt& pt, float outlierTHSlack, ImmRes* tmp, float& Hdd, float& bd, float idepth) {
    if (tmp->state_state == 1 /*OOB*/) { tmp->state_NewState = 1; return tmp->state_energy; }
    const Precalc& pc = E->precalc[(size_t)pt.host * E->nF + tmp->target];
    float energyLeft = 0;
    const float* dIl = E->frames[tmp->target].dI.data();
    const float* R = pc.PRE_RTll;
    const float* t = pc.PRE_tTll;
    const float affLL[2] = {pc.PRE_aff_mode[0], pc.PRE_aff_mode[1]};
    for (int idx = 0; idx < patternNum; idx++) {
        const int dx = patternP[idx][0], dy = patternP[idx][1];
        // projectPoint (ResidualProjections.h:32-59)
        const float KliP[3] = {(pt.u + dx - E->cxl) * E->fxli, (pt.v + dy - E->cyl) * E->fyli, 1};
        const float ptp[3] = {((R[0] * KliP[0] + R[1] * KliP[1]) + R[2] * KliP[2]) + t[0] * idepth,
                              ((R[3] * KliP[0] + R[4] * KliP[1]) + R[5] * KliP[2]) + t[1] * idepth,
                              ((R[6] * KliP[0] + R[7] * KliP[1]) + R[8] * KliP[2]) + t[2] * idepth};
        const float drescale = 1.0f / ptp[2];
        bool ok = drescale > 0;
        float u = 0, v = 0, Ku = 0, Kv = 0;
        if (ok) {
            u = ptp[0] * drescale;
            v = ptp[1] * drescale;
            Ku = u * E->fxl + E->cxl;
            Kv = v * E->fyl + E->cyl;
            ok = Ku > 1.1f && Kv > 1.1f && Ku < E->wM3G && Kv < E->hM3G;
        }
        if (!ok) { tmp->state_NewState = 1; return tmp->state_energy; }
        float hit[3];
        interp33(dIl, Ku, Kv, E->w, hit);
        if (!std::isfinite(hit[0])) { tmp->state_NewState = 1; return tmp->state_energy; }
        const float residual = hit[0] - (affLL[0] * pt.color[idx] + affLL[1]);
        float hw = fabsf(residual) < setting_huberTH_imm ? 1 : setting_huberTH_imm / fabsf(residual);
        energyLeft += pt.weights[idx] * pt.weights[idx] * hw * residual * residual * (2 - hw);
        const float dxInterp = hit[1] * E->fxl, dyInterp = hit[2] * E->fyl;
        const float d_idepth = (dxInterp * drescale * (t[0] - t[2] * u) + dyInterp * drescale * (t[1] - t[2] * v)) * SCALE_IDEPTH;   // derive_idepth
        hw *= pt.weights[idx] * pt.weights[idx];
        Hdd += (hw * d_idepth) * d_idepth;
        bd += (hw * residual) * d_idepth;
    }
    if (energyLeft > pt.energyTH * outlierTHSlack) { energyLeft = pt.energyTH * outlierTHSlack; tmp->state_NewState = 2 /*OUTLIER*/; }
    else tmp->state_NewState = 0 /*IN*/;
    tmp->state_NewEnergy = energyLeft;
    return energyLeft;
}

static int imm_optimize(const EF* E, const ImmPt& pt, int minObs, ImmRes* res, float* idepth_out) {
    int nres = 0;
    for (int f = 0; f < E->nF; ++f)
        if (f != pt.host) {
            res[nres].state_NewEnergy = res[nres].state_energy = 0;
            res[nres].state_NewState = 2;
            res[nres].state_state = 0;
            res[nres].target = f;
            nres++;
        }
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (pt.idepth_max + pt.idepth_min) * 0.5f;
    const float trueDepth = currentIdepth;
    if (!pt.isFromSensor) {
        for (int i = 0; i < nres; i++) {
            lastEnergy += imm_linearize_residual(E, pt, 1000, res + i, lastHdd, lastbd, currentIdepth);
            res[i].state_state = res[i].state_NewState;
            res[i].state_energy = res[i].state_NewEnergy;
        }
        if (!std::isfinite(lastEnergy) || lastHdd < setting_minIdepthH_act) return 0;
        float lambda = 0.1f;
        for (int iteration = 0; iteration < setting_GNItsOnPointActivation; iteration++) {
            float H = lastHdd;
            H *= 1 + lambda;
            const float step = (float)((1.0 / H) * lastbd);
            const float newIdepth = currentIdepth - step;
            float newHdd = 0, newbd = 0, newEnergy = 0;
            for (int i = 0; i < nres; i++) newEnergy += imm_linearize_residual(E, pt, 1, res + i, newHdd, newbd, newIdepth);
            if (!std::isfinite(lastEnergy) || newHdd < setting_minIdepthH_act) return 0;
            if (newEnergy < lastEnergy) {
                currentIdepth = newIdepth;
                lastHdd = newHdd;
                lastbd = newbd;
                lastEnergy = newEnergy;
                for (int i = 0; i < nres; i++) { res[i].state_state = res[i].state_NewState; res[i].state_energy = res[i].state_NewEnergy; }
                lambda *= 0.5f;
            } else {
                lambda *= 5;
            }
            if (fabsf(step) < 0.0001 * currentIdepth) break;
        }
    }
    if (!std::isfinite(currentIdepth)) return -1;
    int numGoodRes = 0;
    for (int i = 0; i < nres; i++) if (res[i].state_state == 0) numGoodRes++;
    if (numGoodRes < minObs) return -1;
    if (!std::isfinite(pt.energyTH)) return -1;            // `new PointHessian(point)` copies energyTH (:139-140)
    *idepth_out = pt.isFromSensor ? trueDepth : currentIdepth;
    return 1;
}

extern "C" {
// ---- hooks with the signatures of oracle/ref_glue.cpp (tests/test_ref_pin.py) ----
void orc_kat_acc_approx(int n, const float* in, float* H169, double* num) {
    AccumulatorApprox* acc = new AccumulatorApprox();
    acc->initialize();
    for (int i = 0; i < n; ++i) {
        const float* p = in + (size_t)35 * i;
        acc->update(p, p + 4, p + 10, p + 14, p[20], p[21], p[22]);
        acc->updateTopRight(p, p + 4, p + 10, p + 14, p[23], p[24], p[25], p[26], p[27], p[28]);
        acc->updateBotRight(p[29], p[30], p[31], p[32], p[33], p[34]);
    }
    acc->finish();
    for (int a = 0; a < 13; ++a) for (int b = 0; b < 13; ++b) H169[a * 13 + b] = acc->H[a][b];
    *num = (double)acc->num;
    delete acc;
}
// 0: left to right (default); 1 / 2: the two orders Eigen may use for the small dot products; 3: the per-type mix Eigen 3.2.8 / SSE2 uses
// (process-wide; the reference-pin tests run with 3)
void orc_set_redux_order(int order) { g_redux_order = order; }
void orc_kat_acc11(int n, const float* vals, float* A) {
    Accumulator11 acc;
    acc.initialize();
    for (int i = 0; i < n; ++i) acc.updateSingle(vals[i]);
    acc.finish();
    *A = acc.A;
}
void orc_kat_interp33_backend(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
    for (int i = 0; i < n; ++i) interp33(img3, x[i], y[i], width, out3 + 3 * i);
}


void* orc_ef_create(int w, int h) {
    EF* E = new EF();
    E->w = w; E->h = h; E->nF = 0; E->wM3G = w - 3; E->hM3G = h - 3;
    return E;
}
void orc_ef_destroy(void* e) { delete (EF*)e; }

void orc_ef_set_calib(void* e, const double value_scaled[4], const double value_minus_value_zero[4]) {
    EF* E = (EF*)e;
    for (int i = 0; i < 4; ++i) { E->value_scaled[i] = value_scaled[i]; E->value_minus_value_zero[i] = value_minus_value_zero[i]; }
    // CalibHessian::setValueScaled (HessianBlocks.h:318-330): value = SCALE_*_INVERSE * value_scaled
    E->value[0] = (1.0f / SCALE_F) * value_scaled[0]; E->value[1] = (1.0f / SCALE_F) * value_scaled[1];
    E->value[2] = (1.0f / SCALE_C) * value_scaled[2]; E->value[3] = (1.0f / SCALE_C) * value_scaled[3];
    for (int i = 0; i < 4; ++i) E->value_zero[i] = E->value[i] - value_minus_value_zero[i];
    calib_update(E);
}

void orc_ef_set_frames(void* e, int nF, const double* evalPT7, const double* state10, const double* state_zero10,
                       const int* frameID, const float* ab_exposure, const float* frameEnergyTH) {
    EF* E = (EF*)e;
    E->nF = nF;
    E->frames.resize(nF);
    for (int i = 0; i < nF; ++i) {
        Frame& f = E->frames[i];
        std::memcpy(f.evalPT.q, evalPT7 + 7 * i, 32); std::memcpy(f.evalPT.t, evalPT7 + 7 * i + 4, 24);
        for (int k = 0; k < 10; ++k) f.state_zero[k] = state_zero10[10 * i + k];
        frame_set_state(f, state10 + 10 * i);
        f.frameID = frameID[i]; f.ab_exposure = ab_exposure[i]; f.frameEnergyTH = frameEnergyTH[i];
        // EFFrame::takeData / FrameHessian::getPrior (HessianBlocks.h:220-250)
        for (int k = 0; k < 6; ++k) f.prior[k] = 0;
        if (f.frameID == 0) { for (int k = 0; k < 3; ++k) f.prior[k] = setting_initialTransPrior; for (int k = 3; k < 6; ++k) f.prior[k] = setting_initialRotPrior; }
        for (int k = 0; k < 10; ++k) f.step[k] = 0;
    }
    const int n = CPARS + 6 * nF;
    E->HM.assign((size_t)n * n, 0); E->bM.assign(n, 0);
}
void orc_ef_set_frame_state(void* e, int idx, const double* state10) { frame_set_state(((EF*)e)->frames[idx], state10); }
void orc_ef_set_frame_image(void* e, int idx, const float* dI) {
    EF* E = (EF*)e;
    E->frames[idx].dI.assign(dI, dI + (size_t)E->w * E->h * 3);
}
void orc_ef_set_points(void* e, int nP, const int* host, const float* u, const float* v, const float* idepth, const float* idepth_zero,
                       const float* color8, const float* weights8, const uint8_t* hasDepthPrior, const uint8_t* isFromSensor) {
    EF* E = (EF*)e;
    E->points.resize(nP);
    for (int i = 0; i < nP; ++i) {
        Point& p = E->points[i];
        std::memset(&p, 0, sizeof(p));
        p.host = host[i]; p.u = u[i]; p.v = v[i];
        p.idepth = idepth[i]; p.idepth_scaled = SCALE_IDEPTH * idepth[i];
        p.idepth_zero = idepth_zero[i]; p.idepth_zero_scaled = SCALE_IDEPTH * idepth_zero[i];
        for (int k = 0; k < 8; ++k) { p.color[k] = color8[8 * i + k]; p.weights[k] = weights8[8 * i + k]; }
        p.hasDepthPrior = hasDepthPrior[i]; p.isFromSensor = isFromSensor[i];
        p.priorF = p.hasDepthPrior ? setting_idepthFixPrior * SCALE_IDEPTH * SCALE_IDEPTH : 0;  // EFPoint::takeData
        p.r0 = p.r1 = 0;
    }
}
void orc_ef_set_point_idepth(void* e, const float* idepth, const float* idepth_zero) {
    EF* E = (EF*)e;
    for (size_t i = 0; i < E->points.size(); ++i) {
        Point& p = E->points[i];
        p.idepth = idepth[i]; p.idepth_scaled = SCALE_IDEPTH * idepth[i];
        p.idepth_zero = idepth_zero[i]; p.idepth_zero_scaled = SCALE_IDEPTH * idepth_zero[i];
    }
}
// residuals sorted by point (a point's residualsAll order = order given here)
void orc_ef_set_residuals(void* e, int nR, const int* point, const int* target, const int* state_state, const uint8_t* hasMatcher,
                          const double* matcher2, const uint8_t* isLinearized, const uint8_t* isActive) {
    EF* E = (EF*)e;
    E->res.resize(nR);
    for (Point& p : E->points) p.r0 = p.r1 = 0;
    for (int i = 0; i < nR; ++i) {
        Residual& r = E->res[i];
        std::memset(&r, 0, sizeof(r));
        r.point = point[i]; r.host = E->points[point[i]].host; r.target = target[i];
        r.state_state = state_state[i]; r.state_NewState = OUTLIER;
        r.hasMatcher = hasMatcher[i]; r.matcher[0] = matcher2[2 * i]; r.matcher[1] = matcher2[2 * i + 1];
        r.isLinearized = isLinearized[i]; r.isActive = isActive[i];
        Point& p = E->points[point[i]];
        if (p.r1 == 0 && p.r0 == 0) { p.r0 = i; p.r1 = i + 1; }
        else p.r1 = i + 1;
    }
}
void orc_ef_set_marg_prior(void* e, const double* HM, const double* bM) {
    EF* E = (EF*)e; const int n = CPARS + 6 * E->nF;
    E->HM.assign(HM, HM + (size_t)n * n); E->bM.assign(bM, bM + n);
}
// FrameHessian::setStateZero, null-space columns (HessianBlocks.cpp:57-76) + FullSystem::getNullspaces (FullSystemOptimize.cpp:548-588):
// lastNullspaces_pose (6) followed by lastNullspaces_scale (1), the set EnergyFunctional::orthogonalize projects out (:727-735)
int orc_ef_compute_nullspaces(void* e, double* out) {
    EF* E = (EF*)e; const int nF = E->nF, n = CPARS + 6 * nF;
    E->nullspaces.assign(7, std::vector<double>(n, 0.0));
    const float SCALE_XI_TRANS_INVERSE = 1.0f / SCALE_XI_TRANS, SCALE_XI_ROT_INVERSE = 1.0f / SCALE_XI_ROT;
    for (int h = 0; h < nF; ++h) {
        const SE3 T = E->frames[h].evalPT;                 // get_worldToCam_evalPT()
        double pose[6][6], scale[6];
        for (int i = 0; i < 6; ++i) {
            double eps[6] = {0, 0, 0, 0, 0, 0}, meps[6] = {0, 0, 0, 0, 0, 0};
            eps[i] = 1e-3; meps[i] = -1e-3;
            const SE3 EepsP = se3_exp(eps), EepsM = se3_exp(meps);
            const SE3 P = se3_mul(se3_mul(T, EepsP), se3_inverse(T));
            const SE3 M = se3_mul(se3_mul(T, EepsM), se3_inverse(T));
            double lp[6], lm[6];
            se3_log(P, lp); se3_log(M, lm);
            for (int r = 0; r < 6; ++r) pose[i][r] = (lp[r] - lm[r]) / (2e-3);
        }
        {
            SE3 P = T, M = T;
            for (int r = 0; r < 3; ++r) { P.t[r] *= 1.00001; M.t[r] /= 1.00001; }
            P = se3_mul(P, se3_inverse(T)); M = se3_mul(M, se3_inverse(T));
            double lp[6], lm[6];
            se3_log(P, lp); se3_log(M, lm);
            for (int r = 0; r < 6; ++r) scale[r] = (lp[r] - lm[r]) / (2e-3);
        }
        for (int i = 0; i < 7; ++i)
            for (int r = 0; r < 6; ++r) {
                double v = i < 6 ? pose[i][r] : scale[r];
                v *= r < 3 ? SCALE_XI_TRANS_INVERSE : SCALE_XI_ROT_INVERSE;
                E->nullspaces[i][CPARS + 6 * h + r] = v;
            }
    }
    if (out) for (int j = 0; j < 7; ++j) std::memcpy(out + (size_t)j * n, E->nullspaces[j].data(), sizeof(double) * n);
    return 7;
}
void orc_ef_set_nullspaces(void* e, int k, const double* ns) {
    EF* E = (EF*)e; const int n = CPARS + 6 * E->nF;
    E->nullspaces.clear();
    for (int j = 0; j < k; ++j) E->nullspaces.emplace_back(ns + (size_t)j * n, ns + (size_t)(j + 1) * n);
}
void orc_ef_set_precalc(void* e) { set_precalc((EF*)e); set_delta((EF*)e); }
void orc_ef_set_adjoints(void* e) { set_adjoints((EF*)e); }
double orc_ef_linearize_all(void* e) { return linearize_all((EF*)e); }   // FullSystem::linearizeAll(false), incl. setNewFrameEnergyTH
double orc_ef_optimize_finish(void* e, float* relbs_max, int* ngood_inc, uint8_t* removed) { return optimize_finish((EF*)e, relbs_max, ngood_inc, removed); }
void orc_ef_get_frame_energy_th(void* e, float* th) { EF* E = (EF*)e; for (int i = 0; i < E->nF; ++i) th[i] = E->frames[i].frameEnergyTH; }
void orc_ef_get_evalPT(void* e, int idx, double* q4t3, double* state_zero10) {
    const Frame& f = ((EF*)e)->frames[idx];
    for (int i = 0; i < 4; ++i) q4t3[i] = f.evalPT.q[i];
    for (int i = 0; i < 3; ++i) q4t3[4 + i] = f.evalPT.t[i];
    for (int i = 0; i < 10; ++i) state_zero10[i] = f.state_zero[i];
}
void orc_ef_apply_res(void* e) { for (Residual& r : ((EF*)e)->res) if (!r.isLinearized) apply_res(r); }
void orc_ef_solve_system(void* e, int iteration, double lambda) { solve_system((EF*)e, iteration, lambda); }

int orc_ef_dim(void* e) { return CPARS + 6 * ((EF*)e)->nF; }
void orc_ef_get_system(void* e, double* HA, double* bA, double* Hsc, double* bsc, double* HFinal, double* bFinal, double* x) {
    EF* E = (EF*)e; const int n = CPARS + 6 * E->nF;
    if (HA) std::memcpy(HA, E->HA.data(), sizeof(double) * n * n);
    if (bA) std::memcpy(bA, E->bA.data(), sizeof(double) * n);
    if (Hsc) std::memcpy(Hsc, E->Hsc.data(), sizeof(double) * n * n);
    if (bsc) std::memcpy(bsc, E->bsc.data(), sizeof(double) * n);
    if (HFinal) std::memcpy(HFinal, E->HFinal.data(), sizeof(double) * n * n);
    if (bFinal) std::memcpy(bFinal, E->bFinal.data(), sizeof(double) * n);
    if (x) std::memcpy(x, E->lastX.data(), sizeof(double) * n);
}
// per residual: J (24 floats: resF2, Jpdxi 12, Jpdc 8, Jpdd 2), which = 0 new / 1 EF ; states
void orc_ef_get_residual_J(void* e, int which, float* out24) {
    EF* E = (EF*)e;
    for (size_t i = 0; i < E->res.size(); ++i) {
        const RawJ& J = which ? E->res[i].Jef : E->res[i].Jnew;
        float* o = out24 + 24 * i;
        o[0] = J.resF[0]; o[1] = J.resF[1];
        for (int k = 0; k < 6; ++k) { o[2 + k] = J.Jpdxi[0][k]; o[8 + k] = J.Jpdxi[1][k]; }
        for (int k = 0; k < 4; ++k) { o[14 + k] = J.Jpdc[0][k]; o[18 + k] = J.Jpdc[1][k]; }
        o[22] = J.Jpdd[0]; o[23] = J.Jpdd[1];
    }
}
void orc_ef_get_residual_state(void* e, int* state_state, int* state_new, double* energy_new, double* energy_with_outlier, uint8_t* isActive) {
    EF* E = (EF*)e;
    for (size_t i = 0; i < E->res.size(); ++i) {
        const Residual& r = E->res[i];
        if (state_state) state_state[i] = r.state_state;
        if (state_new) state_new[i] = r.state_NewState;
        if (energy_new) energy_new[i] = r.state_NewEnergy;
        if (energy_with_outlier) energy_with_outlier[i] = r.state_NewEnergyWithOutlier;
        if (isActive) isActive[i] = r.isActive;
    }
}
// per point: [Hdd_accAF, bd_accAF, Hcd_accAF(4), HdiF, bdSumF, step] = 9 floats
void orc_ef_get_points(void* e, float* out9) {
    EF* E = (EF*)e;
    for (size_t i = 0; i < E->points.size(); ++i) {
        const Point& p = E->points[i];
        float* o = out9 + 9 * i;
        o[0] = p.Hdd_accAF; o[1] = p.bd_accAF;
        for (int k = 0; k < 4; ++k) o[2 + k] = p.Hcd_accAF[k];
        o[6] = p.HdiF; o[7] = p.bdSumF; o[8] = p.step;
    }
}
void orc_ef_get_frame_steps(void* e, double* steps6, double* calibStep4) {
    EF* E = (EF*)e;
    for (int h = 0; h < E->nF; ++h) for (int i = 0; i < 6; ++i) steps6[6 * h + i] = E->frames[h].step[i];
    for (int i = 0; i < 4; ++i) calibStep4[i] = E->calibStep[i];
}
// top accumulators after the last solve: nF*nF x 13 x 13 floats (index h + nF*t)
void orc_ef_get_top_acc(void* e, float* out) {
    EF* E = (EF*)e;
    for (size_t k = 0; k < E->accA.size(); ++k)
        for (int r = 0; r < 13; ++r) for (int c = 0; c < 13; ++c) out[k * 169 + r * 13 + c] = E->accA[k].H[r][c];
}
void orc_ef_get_precalc(void* e, int h, int t, float* out) {  // KRKi9, Kt3, R0 9, t0 3, aff2, b0 = 27
    EF* E = (EF*)e; const Precalc& P = E->precalc[(size_t)h * E->nF + t];
    std::memcpy(out, P.PRE_KRKiTll, 36); std::memcpy(out + 9, P.PRE_KtTll, 12); std::memcpy(out + 12, P.PRE_RTll_0, 36);
    std::memcpy(out + 21, P.PRE_tTll_0, 12); out[24] = P.PRE_aff_mode[0]; out[25] = P.PRE_aff_mode[1]; out[26] = P.PRE_b0_mode;
}
void orc_ef_get_adjoints(void* e, double* adHost, double* adTarget) {
    EF* E = (EF*)e;
    std::memcpy(adHost, E->adHost.data(), sizeof(double) * E->adHost.size());
    std::memcpy(adTarget, E->adTarget.data(), sizeof(double) * E->adTarget.size());
}
int orc_ef_res_in_A(void* e) { return ((EF*)e)->resInA; }
// SC accumulators after the last solve: accE [nF*nF][8][4], accEB [nF*nF][8], accD [nF^3][8][8], accHcc [4][4], accbc [4]
void orc_ef_get_sc_acc(void* e, float* accE, float* accEB, float* accD, float* Hcc, float* bc) {
    EF* E = (EF*)e;
    std::memcpy(accE, E->scE.data(), 4 * E->scE.size()); std::memcpy(accEB, E->scEB.data(), 4 * E->scEB.size());
    std::memcpy(accD, E->scD.data(), 4 * E->scD.size()); std::memcpy(Hcc, E->scHcc.data(), 64); std::memcpy(bc, E->scbc.data(), 16);
}
void orc_ef_fix_linearization(void* e, const uint8_t* mask) { fix_linearization((EF*)e, mask); }
void orc_ef_reset_oob(void* e, const uint8_t* mask) {   // PointFrameResidual::resetOOB (Residuals.h:70-76) for the flagged points (NULL: all)
    EF* E = (EF*)e;
    for (Residual& r : E->res) {
        if (mask && !mask[r.point]) continue;
        if (r.isLinearized) continue;
        r.state_NewEnergy = r.state_energy = 0; r.state_NewState = OUTLIER; r.state_state = IN;
    }
}
void orc_ef_marginalize_points(void* e, const uint8_t* marg, const uint8_t* drop) { marginalize_points((EF*)e, marg, drop); }
void orc_ef_marginalize_frame(void* e, int idx, double* HM_out, double* bM_out) { marginalize_frame((const EF*)e, idx, HM_out, bM_out); }
void orc_ef_get_marg_prior(void* e, double* HM, double* bM) {
    EF* E = (EF*)e; const int n = CPARS + 6 * E->nF;
    for (size_t i = 0; i < (size_t)n * n; ++i) HM[i] = E->HM[i];
    for (int i = 0; i < n; ++i) bM[i] = E->bM[i];
}
void orc_ef_get_res_toZero(void* e, float* out2, uint8_t* isLinearized) {
    EF* E = (EF*)e;
    for (size_t i = 0; i < E->res.size(); ++i) { out2[2 * i] = E->res[i].res_toZeroF[0]; out2[2 * i + 1] = E->res[i].res_toZeroF[1]; isLinearized[i] = E->res[i].isLinearized; }
}
void orc_ef_get_adHTdeltaF(void* e, float* out) { EF* E = (EF*)e; for (size_t i = 0; i < E->adHTdeltaF.size(); ++i) out[i] = E->adHTdeltaF[i]; }
void orc_ef_get_frame_prior(void* e, int idx, double* prior6, double* delta_prior6) {
    const Frame& f = ((EF*)e)->frames[idx];
    for (int i = 0; i < 6; ++i) { prior6[i] = f.prior[i]; delta_prior6[i] = f.delta_prior[i]; }
}
void orc_ef_set_fixed_its(void* e, int on) { ((EF*)e)->fixedIts = on != 0; }
void orc_ef_set_threads(void* e, int n) { ((EF*)e)->nThreads = n < 1 ? 1 : n; }
int orc_ef_optimize(void* e, int its, double* trace, int stride, int cap) { return optimize((EF*)e, its, trace, stride, cap); }
double orc_ef_calc_L_energy(void* e) { return calc_L_energy((EF*)e); }
double orc_ef_calc_M_energy(void* e) { return calc_M_energy((EF*)e); }
void orc_ef_get_state(void* e, double* value_scaled4, double* state10, float* idepth) {
    EF* E = (EF*)e;
    for (int i = 0; i < 4; ++i) value_scaled4[i] = E->value_scaled[i];
    for (int h = 0; h < E->nF; ++h) for (int i = 0; i < 10; ++i) state10[10 * h + i] = E->frames[h].state[i];
    for (size_t i = 0; i < E->points.size(); ++i) idepth[i] = E->points[i].idepth;
}

// optimizeImmaturePoint for n points against the frames / precalc currently set on the handle (orc_ef_set_precalc first).
// res_state[n][nF]: state_state of the residual towards frame t (-1 for t == host); result[n]: 0 / -1 / 1; idepth[n] valid if 1.
void orc_ef_optimize_immature(void* e, int n, const int* host, const float* u, const float* v, const float* idepth_min, const float* idepth_max,
                              const float* energyTH, const float* color8, const float* weights8, const uint8_t* isFromSensor, int minObs,
                              int* result, float* idepth, int* res_state) {
    const EF* E = (const EF*)e;
    std::vector<ImmRes> res(E->nF);
    for (int i = 0; i < n; ++i) {
        ImmPt pt;
        pt.host = host[i]; pt.u = u[i]; pt.v = v[i]; pt.idepth_min = idepth_min[i]; pt.idepth_max = idepth_max[i]; pt.energyTH = energyTH[i];
        for (int k = 0; k < 8; ++k) { pt.color[k] = color8[8 * i + k]; pt.weights[k] = weights8[8 * i + k]; }
        pt.isFromSensor = isFromSensor[i] != 0;
        float id = NAN;
        result[i] = imm_optimize(E, pt, minObs, res.data(), &id);
        idepth[i] = id;
        for (int t = 0; t < E->nF; ++t) res_state[(size_t)i * E->nF + t] = -1;
        int k = 0;
        for (int t = 0; t < E->nF; ++t) if (t != pt.host) res_state[(size_t)i * E->nF + t] = res[k++].state_state;
    }
}
}
