// oracle/orc_tracker.cpp -- TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY PINNED against the reference's own translation units (oracle/_ref/libref.so, oracle/README.md; tests/test_ref_pin*.py).
//
// Plain C++ restatement of the reference's photometric coarse tracker, SURVEY.md section 8 rows a1-a9:
//   a1 FrameHessian::makeImages          src/FullSystem/HessianBlocks.cpp:107-167
//   a2 CoarseTracker::makeK              src/FullSystem/CoarseTracker.cpp:77-106
//   a4 CoarseTracker::calcRes            src/FullSystem/CoarseTracker.cpp:486-634
//   a5 CoarseTracker::calcGSSSE          src/FullSystem/CoarseTracker.cpp:427-484
//   a6 Accumulator9                      src/OptimizationBackend/MatrixAccumulators.h:934-1115,1273-1292
//   a7 CoarseTracker::trackNewestCoarse  src/FullSystem/CoarseTracker.cpp:662-838
//   a8 AffLight::fromToVecExposure       src/util/NumType.h:149-158
//   a9 getInterpolatedElement33          src/util/globalFuncs.h:51-65
// The reference's third-party dependencies are absent here, but its own source files compile unmodified against the stand-ins of
// oracle/ref_shim (oracle/Makefile target `ref`); tests/test_ref_pin_tracker.py runs every function of this file against that build.
//
// Arithmetic follows the reference operation by operation: float32 where the reference uses float,
// the same operand order, no FMA contraction (build with -ffp-contract=off, baseline SSE2 like the
// reference's `-O3` x86-64 build), 4-lane SSE-shaped accumulation with the 1k / 1M tiered shift-up.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this library; the
// product (libsdvgn) never links or loads it.
#include "orc_math.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace orc {

static const int kMaxLvl = 6;  // PYR_LEVELS, src/util/settings.h

// HessianBlocks.h:33-40
static const float SCALE_XI_ROT = 1.0f, SCALE_XI_TRANS = 0.5f, SCALE_A = 10.0f, SCALE_B = 1000.0f;

// ---- a6: Accumulator9 (MatrixAccumulators.h:934-1292) ------------------------------------------
struct Accumulator9 {
    float S[4 * 45], S1k[4 * 45], S1m[4 * 45];
    float numIn1, numIn1k, numIn1m;
    size_t num;
    float H[9][9];
    void initialize() {
        std::memset(S, 0, sizeof(S)); std::memset(S1k, 0, sizeof(S1k)); std::memset(S1m, 0, sizeof(S1m));
        num = 0; numIn1 = numIn1k = numIn1m = 0;
        std::memset(H, 0, sizeof(H));
    }
    void shiftUp(bool force) {  // :1273-1292
        if (numIn1 > 1000 || force) {
            for (int i = 0; i < 4 * 45; ++i) S1k[i] = S[i] + S1k[i];
            numIn1k += numIn1; numIn1 = 0;
            std::memset(S, 0, sizeof(S));
        }
        if (numIn1k > 1000 || force) {
            for (int i = 0; i < 4 * 45; ++i) S1m[i] = S1k[i] + S1m[i];
            numIn1m += numIn1k; numIn1k = 0;
            std::memset(S1k, 0, sizeof(S1k));
        }
    }
    // updateSSE_eighted :1040-1115 -- J[k][lane], w[lane]
    void updateWeighted(const float J[9][4], const float w[4]) {
        int idx = 0;
        for (int r = 0; r < 9; ++r) {
            float Jw[4];
            for (int l = 0; l < 4; ++l) Jw[l] = J[r][l] * w[l];
            for (int c = r; c < 9; ++c) {
                for (int l = 0; l < 4; ++l) S[idx + l] = S[idx + l] + Jw[l] * J[c][l];
                idx += 4;
            }
        }
        num += 4; numIn1++;
        shiftUp(false);
    }
    void finish() {  // :953-970
        std::memset(H, 0, sizeof(H));
        shiftUp(true);
        int idx = 0;
        for (int r = 0; r < 9; ++r)
            for (int c = r; c < 9; ++c) {
                float d = S1m[idx + 0] + S1m[idx + 1] + S1m[idx + 2] + S1m[idx + 3];
                H[r][c] = H[c][r] = d;
                idx += 4;
            }
    }
};

struct Tracker {
    int levels;
    int w[kMaxLvl], h[kMaxLvl];
    float fx[kMaxLvl], fy[kMaxLvl], cx[kMaxLvl], cy[kMaxLvl];
    float K[kMaxLvl][9], Ki[kMaxLvl][9];
    std::vector<float> pc_u[kMaxLvl], pc_v[kMaxLvl], pc_idepth[kMaxLvl], pc_color[kMaxLvl];
    int pc_n[kMaxLvl];
    // lastRef: exposure + aff_g2l ; newFrame: exposure + dIp pyramid
    float ref_exposure, new_exposure;
    double ref_a, ref_b;
    std::vector<float> dIp[kMaxLvl];  // AoS {I,dx,dy}
    // warped buffers (CoarseTracker.h:93-102)
    std::vector<float> bw_idepth, bw_u, bw_v, bw_dx, bw_dy, bw_res, bw_w, bw_ref;
    int bw_n;
    Accumulator9 acc;
    // settings (src/util/settings.cpp:93-94,101,112; launch/run.launch mode=1 -> affine modes 0)
    float huberTH, coarseCutoffTH, affineOptModeA, affineOptModeB;
    // side outputs (CoarseTracker.h:62-65)
    double lastResiduals[5];
    double lastFlowIndicators[3];
};

// ---- a9: getInterpolatedElement33 (globalFuncs.h:51-65) ----------------------------------------
static inline void interp33(const float* mat, float x, float y, int width, float out[3]) {
    int ix = (int)x;
    int iy = (int)y;
    float dx = x - ix;
    float dy = y - iy;
    float dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    for (int k = 0; k < 3; ++k)
        out[k] = ((w11 * bp[3 * (1 + width) + k] + w01 * bp[3 * width + k]) + w10 * bp[3 + k]) + w00 * bp[k];
}

// ---- a1: makeImages (HessianBlocks.cpp:107-167) ------------------------------------------------
// out[lvl] = AoS float3 {I,dx,dy}.  Gradient rows 0 and hl-1 are uninitialised memory in the
// reference (loop bounds :145); the oracle fills them with NaN as a canary -- no consumer may read them.
static void make_images(const float* color, int w0, int h0, int levels, std::vector<float>* out) {
    for (int l = 0; l < levels; ++l) {
        int wl = w0 >> l, hl = h0 >> l;
        out[l].assign((size_t)wl * hl * 3, std::numeric_limits<float>::quiet_NaN());
    }
    for (int i = 0; i < w0 * h0; ++i) out[0][3 * i] = color[i];
    for (int lvl = 0; lvl < levels; ++lvl) {
        int wl = w0 >> lvl, hl = h0 >> lvl;
        float* d = out[lvl].data();
        if (lvl > 0) {
            int wlm1 = w0 >> (lvl - 1);
            const float* dm = out[lvl - 1].data();
            for (int y = 0; y < hl; ++y)
                for (int x = 0; x < wl; ++x)
                    d[3 * (x + y * wl)] = 0.25f * (((dm[3 * (2 * x + 2 * y * wlm1)] + dm[3 * (2 * x + 1 + 2 * y * wlm1)]) +
                                                    dm[3 * (2 * x + 2 * y * wlm1 + wlm1)]) +
                                                   dm[3 * (2 * x + 1 + 2 * y * wlm1 + wlm1)]);
        }
        for (int idx = wl; idx < wl * (hl - 1); ++idx) {
            float dx = 0.5f * (d[3 * (idx + 1)] - d[3 * (idx - 1)]);
            float dy = 0.5f * (d[3 * (idx + wl)] - d[3 * (idx - wl)]);
            if (!std::isfinite(dx)) dx = 0;
            if (!std::isfinite(dy)) dy = 0;
            d[3 * idx + 1] = dx;
            d[3 * idx + 2] = dy;
        }
    }
}

// ---- a2: makeK (CoarseTracker.cpp:77-106) ------------------------------------------------------
static void make_K(Tracker* T, float fx0, float fy0, float cx0, float cy0) {
    T->fx[0] = fx0; T->fy[0] = fy0; T->cx[0] = cx0; T->cy[0] = cy0;
    for (int level = 1; level < T->levels; ++level) {
        T->fx[level] = T->fx[level - 1] * 0.5;            // float * double -> float
        T->fy[level] = T->fy[level - 1] * 0.5;
        T->cx[level] = (T->cx[0] + 0.5) / ((int)1 << level) - 0.5;  // evaluated in double, stored float
        T->cy[level] = (T->cy[0] + 0.5) / ((int)1 << level) - 0.5;
    }
    for (int level = 0; level < T->levels; ++level) {
        float* K = T->K[level];
        K[0] = T->fx[level]; K[1] = 0; K[2] = T->cx[level];
        K[3] = 0; K[4] = T->fy[level]; K[5] = T->cy[level];
        K[6] = 0; K[7] = 0; K[8] = 1;
        inv3f(K, T->Ki[level]);
    }
}

// ---- a4: calcRes (CoarseTracker.cpp:486-634) ---------------------------------------------------
static void calc_res(Tracker* T, int lvl, const SE3& refToNew, double aff_a, double aff_b, float cutoffTH, double rs[6]) {
    float E = 0;
    int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
    const int wl = T->w[lvl], hl = T->h[lvl];
    const float* dINewl = T->dIp[lvl].data();
    const float fxl = T->fx[lvl], fyl = T->fy[lvl], cxl = T->cx[lvl], cyl = T->cy[lvl];
    const float* Ki = T->Ki[lvl];

    double Rd[9];
    quat_to_R(refToNew.q, Rd);
    float Rf[9], RKi[9], t[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Rd[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            RKi[i * 3 + j] = (Rf[i * 3 + 0] * Ki[0 * 3 + j] + Rf[i * 3 + 1] * Ki[1 * 3 + j]) + Rf[i * 3 + 2] * Ki[2 * 3 + j];
    for (int i = 0; i < 3; ++i) t[i] = (float)refToNew.t[i];
    double affd[2];
    aff_from_to(T->ref_exposure, T->new_exposure, T->ref_a, T->ref_b, aff_a, aff_b, affd);
    const float affLL0 = (float)affd[0], affLL1 = (float)affd[1];

    float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
    const float huber = T->huberTH;
    const float maxEnergy = 2 * huber * cutoffTH - huber * huber;

    const int nl = T->pc_n[lvl];
    const float* lpc_u = T->pc_u[lvl].data();
    const float* lpc_v = T->pc_v[lvl].data();
    const float* lpc_idepth = T->pc_idepth[lvl].data();
    const float* lpc_color = T->pc_color[lvl].data();

    const size_t cap = (size_t)nl + 4;
    if (T->bw_u.size() < cap) {
        T->bw_idepth.resize(cap); T->bw_u.resize(cap); T->bw_v.resize(cap); T->bw_dx.resize(cap);
        T->bw_dy.resize(cap); T->bw_res.resize(cap); T->bw_w.resize(cap); T->bw_ref.resize(cap);
    }

    auto mv = [](const float* M, float x, float y, const float* tt, float id, float sgn, float* o) {
        for (int r = 0; r < 3; ++r) {
            float m = (M[r * 3 + 0] * x + M[r * 3 + 1] * y) + M[r * 3 + 2] * 1.0f;
            o[r] = (sgn > 0) ? (m + tt[r] * id) : (m - tt[r] * id);
        }
    };

    for (int i = 0; i < nl; ++i) {
        float id = lpc_idepth[i];
        float x = lpc_u[i];
        float y = lpc_v[i];
        float pt[3];
        mv(RKi, x, y, t, id, 1, pt);
        float u = pt[0] / pt[2];
        float v = pt[1] / pt[2];
        float Ku = fxl * u + cxl;
        float Kv = fyl * v + cyl;
        float new_idepth = id / pt[2];

        if (lvl == 0 && i % 32 == 0) {  // :538-566
            float ptT[3], ptT2[3], pt3[3];
            mv(Ki, x, y, t, id, 1, ptT);
            float uT = ptT[0] / ptT[2], vT = ptT[1] / ptT[2];
            float KuT = fxl * uT + cxl, KvT = fyl * vT + cyl;
            mv(Ki, x, y, t, id, -1, ptT2);
            float uT2 = ptT2[0] / ptT2[2], vT2 = ptT2[1] / ptT2[2];
            float KuT2 = fxl * uT2 + cxl, KvT2 = fyl * vT2 + cyl;
            mv(RKi, x, y, t, id, -1, pt3);
            float u3 = pt3[0] / pt3[2], v3 = pt3[1] / pt3[2];
            float Ku3 = fxl * u3 + cxl, Kv3 = fyl * v3 + cyl;
            sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
            sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
            sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
            sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
            sumSquaredShiftNum += 2;
        }

        if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;

        float refColor = lpc_color[i];
        float hit[3];
        interp33(dINewl, Ku, Kv, wl, hit);
        if (!std::isfinite(hit[0])) continue;
        float residual = hit[0] - (float)(affLL0 * refColor + affLL1);
        float hw = std::fabs(residual) < huber ? 1 : huber / std::fabs(residual);

        if (std::fabs(residual) > cutoffTH) {
            E += maxEnergy;
            numTermsInE++;
            numSaturated++;
        } else {
            E += hw * residual * residual * (2 - hw);
            numTermsInE++;
            T->bw_idepth[numTermsInWarped] = new_idepth;
            T->bw_u[numTermsInWarped] = u;
            T->bw_v[numTermsInWarped] = v;
            T->bw_dx[numTermsInWarped] = hit[1];
            T->bw_dy[numTermsInWarped] = hit[2];
            T->bw_res[numTermsInWarped] = residual;
            T->bw_w[numTermsInWarped] = hw;
            T->bw_ref[numTermsInWarped] = lpc_color[i];
            numTermsInWarped++;
        }
    }
    while (numTermsInWarped % 4 != 0) {  // :603-615
        T->bw_idepth[numTermsInWarped] = 0; T->bw_u[numTermsInWarped] = 0; T->bw_v[numTermsInWarped] = 0;
        T->bw_dx[numTermsInWarped] = 0; T->bw_dy[numTermsInWarped] = 0; T->bw_res[numTermsInWarped] = 0;
        T->bw_w[numTermsInWarped] = 0; T->bw_ref[numTermsInWarped] = 0;
        numTermsInWarped++;
    }
    T->bw_n = numTermsInWarped;

    rs[0] = E;
    rs[1] = numTermsInE;
    rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
    rs[3] = 0;
    rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
    rs[5] = numSaturated / (float)numTermsInE;
}

// ---- a5: calcGSSSE (CoarseTracker.cpp:427-484) -------------------------------------------------
static void calc_gs(Tracker* T, int lvl, double aff_a, double aff_b, double H_out[64], double b_out[8]) {
    Accumulator9& acc = T->acc;
    acc.initialize();
    const float fxl = T->fx[lvl], fyl = T->fy[lvl];
    const float b0 = (float)T->ref_b;
    double affd[2];
    aff_from_to(T->ref_exposure, T->new_exposure, T->ref_a, T->ref_b, aff_a, aff_b, affd);
    const float a = (float)affd[0];
    const int n = T->bw_n;
    for (int i = 0; i < n; i += 4) {
        float J[9][4], w[4];
        for (int l = 0; l < 4; ++l) {
            float dx = T->bw_dx[i + l] * fxl;
            float dy = T->bw_dy[i + l] * fyl;
            float u = T->bw_u[i + l], v = T->bw_v[i + l], id = T->bw_idepth[i + l];
            J[0][l] = id * dx;
            J[1][l] = id * dy;
            J[2][l] = 0.0f - id * (u * dx + v * dy);
            J[3][l] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
            J[4][l] = (u * v) * dy + dx * (1.0f + u * u);
            J[5][l] = u * dy - v * dx;
            J[6][l] = a * (b0 - T->bw_ref[i + l]);
            J[7][l] = -1.0f;
            J[8][l] = T->bw_res[i + l];
            w[l] = T->bw_w[i + l];
        }
        acc.updateWeighted(J, w);
    }
    acc.finish();
    const double s = (double)(1.0f / n);
    double H[8][8], b[8];
    for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) H[r][c] = (double)acc.H[r][c] * s;
        b[r] = (double)acc.H[r][8] * s;
    }
    const float sc[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_A, SCALE_B};
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) H[r][c] *= sc[c];  // column blocks :472-475
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) H[r][c] *= sc[r];  // row blocks :476-479
    for (int r = 0; r < 8; ++r) b[r] *= sc[r];
    for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) H_out[r * 8 + c] = H[r][c];
        b_out[r] = b[r];
    }
}

// trace record per LM trial: [lvl, iteration, lambda, accept, inc[8], E_new, n_new, cutoffRepeat] = 15 doubles
static const int kTraceStride = 15;

// ---- a7: trackNewestCoarse (CoarseTracker.cpp:662-838) -----------------------------------------
static bool track(Tracker* T, SE3& lastToNew_out, double aff_io[2], int coarsestLvl, const double minResForAbort[5],
                  double* trace, int trace_cap, int* trace_n) {
    for (int i = 0; i < 5; ++i) T->lastResiduals[i] = NAN;
    for (int i = 0; i < 3; ++i) T->lastFlowIndicators[i] = 1000;
    int maxIterations[] = {10, 20, 50, 50, 50};
    float lambdaExtrapolationLimit = 0.001;
    SE3 refToNew_current = lastToNew_out;
    double aff_a = aff_io[0], aff_b = aff_io[1];
    bool haveRepeated = false;
    int ntr = 0;

    for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
        double H[64], b[8];
        float levelCutoffRepeat = 1;
        double resOld[6];
        calc_res(T, lvl, refToNew_current, aff_a, aff_b, T->coarseCutoffTH * levelCutoffRepeat, resOld);
        while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
            levelCutoffRepeat *= 2;
            calc_res(T, lvl, refToNew_current, aff_a, aff_b, T->coarseCutoffTH * levelCutoffRepeat, resOld);
        }
        calc_gs(T, lvl, aff_a, aff_b, H, b);
        float lambda = 0.01;

        for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
            double Hl[64];
            std::memcpy(Hl, H, sizeof(Hl));
            for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda);
            double negb[8], inc[8];
            for (int i = 0; i < 8; ++i) negb[i] = -b[i];
            ldlt_solve(8, Hl, negb, inc);

            if (T->affineOptModeA < 0 && T->affineOptModeB < 0) {  // fix a, b :726-730
                double H6[36], x6[6];
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H6[r * 6 + c] = Hl[r * 8 + c];
                ldlt_solve(6, H6, negb, x6);
                for (int i = 0; i < 6; ++i) inc[i] = x6[i];
                inc[6] = inc[7] = 0;
            }
            if (!(T->affineOptModeA < 0) && T->affineOptModeB < 0) {  // fix b :731-735
                double H7[49], x7[7];
                for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) H7[r * 7 + c] = Hl[r * 8 + c];
                ldlt_solve(7, H7, negb, x7);
                for (int i = 0; i < 7; ++i) inc[i] = x7[i];
                inc[7] = 0;
            }
            if (T->affineOptModeA < 0 && !(T->affineOptModeB < 0)) {  // fix a :736-748
                double Hs[64], bs[8];
                std::memcpy(Hs, Hl, sizeof(Hs));
                for (int i = 0; i < 8; ++i) bs[i] = b[i];
                for (int r = 0; r < 8; ++r) Hs[r * 8 + 6] = Hs[r * 8 + 7];
                for (int c = 0; c < 8; ++c) Hs[6 * 8 + c] = Hs[7 * 8 + c];
                bs[6] = bs[7];
                double H7[49], nb7[7], x7[7];
                for (int r = 0; r < 7; ++r) { for (int c = 0; c < 7; ++c) H7[r * 7 + c] = Hs[r * 8 + c]; nb7[r] = -bs[r]; }
                ldlt_solve(7, H7, nb7, x7);
                for (int i = 0; i < 8; ++i) inc[i] = 0;
                for (int i = 0; i < 6; ++i) inc[i] = x7[i];
                inc[6] = 0; inc[7] = x7[6];
            }

            float extrapFac = 1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = std::sqrt(std::sqrt(lambdaExtrapolationLimit / lambda));
            for (int i = 0; i < 8; ++i) inc[i] *= extrapFac;

            double incScaled[8];
            for (int i = 0; i < 8; ++i) incScaled[i] = inc[i];
            for (int i = 0; i < 3; ++i) incScaled[i] *= SCALE_XI_ROT;
            for (int i = 3; i < 6; ++i) incScaled[i] *= SCALE_XI_TRANS;
            incScaled[6] *= SCALE_A;
            incScaled[7] *= SCALE_B;
            double sum = 0;
            for (int i = 0; i < 8; ++i) sum += incScaled[i];
            if (!std::isfinite(sum)) for (int i = 0; i < 8; ++i) incScaled[i] = 0;

            SE3 refToNew_new = se3_mul(se3_exp(incScaled), refToNew_current);
            double aff_a_new = aff_a + incScaled[6];
            double aff_b_new = aff_b + incScaled[7];

            double resNew[6];
            calc_res(T, lvl, refToNew_new, aff_a_new, aff_b_new, T->coarseCutoffTH * levelCutoffRepeat, resNew);
            bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);

            if (trace && ntr < trace_cap) {
                double* tr = trace + (size_t)ntr * kTraceStride;
                tr[0] = lvl; tr[1] = iteration; tr[2] = lambda; tr[3] = accept ? 1 : 0;
                for (int i = 0; i < 8; ++i) tr[4 + i] = incScaled[i];
                tr[12] = resNew[0]; tr[13] = resNew[1]; tr[14] = levelCutoffRepeat;
            }
            ntr++;

            if (accept) {
                calc_gs(T, lvl, aff_a_new, aff_b_new, H, b);
                std::memcpy(resOld, resNew, sizeof(resOld));
                aff_a = aff_a_new; aff_b = aff_b_new;
                refToNew_current = refToNew_new;
                lambda *= 0.5;
            } else {
                lambda *= 4;
                if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
            }
            double nrm = 0;
            for (int i = 0; i < 8; ++i) nrm += inc[i] * inc[i];
            nrm = std::sqrt(nrm);
            if (!(nrm > 1e-3)) break;
        }

        T->lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
        for (int i = 0; i < 3; ++i) T->lastFlowIndicators[i] = resOld[2 + i];
        if (T->lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) { if (trace_n) *trace_n = ntr; return false; }

        if (levelCutoffRepeat > 1 && !haveRepeated) {
            lvl++;
            haveRepeated = true;
        }
    }

    lastToNew_out = refToNew_current;
    aff_io[0] = aff_a; aff_io[1] = aff_b;
    if (trace_n) *trace_n = ntr;

    if ((T->affineOptModeA != 0 && (fabsf((float)aff_io[0]) > 1.2)) || (T->affineOptModeB != 0 && (fabsf((float)aff_io[1]) > 200)))
        return false;
    double rel[2];
    aff_from_to(T->ref_exposure, T->new_exposure, T->ref_a, T->ref_b, aff_io[0], aff_io[1], rel);
    const float relAff0 = (float)rel[0], relAff1 = (float)rel[1];
    if ((T->affineOptModeA == 0 && (fabsf(logf(relAff0)) > 1.5)) || (T->affineOptModeB == 0 && (fabsf(relAff1) > 200)))
        return false;
    if (T->affineOptModeA < 0) aff_io[0] = 0;
    if (T->affineOptModeB < 0) aff_io[1] = 0;
    return true;
}

}  // namespace orc

// ------------------------------- C entry points (ctypes) ---------------------------------------
using namespace orc;
// ---- f1: structPoseEstimation (CoarseTracker.cpp:840-1007) -------------------------------------------------
// Runs right after trackNewestCoarse on every frame (FullSystem.cpp:483-489): 6-dof Gauss-Newton / LM on the 2-D
// reprojection error of <= ~1200 map points against their matched pixel positions, Tukey weights.
// Restated with the reference's quirks kept:
//   * H's diagonal is multiplied by (1+lambda) IN PLACE every iteration (:959) -- the damping accumulates over rejected steps;
//   * on accept H,b are rebuilt at the OLD pose worldToCur_current (:983) before the pose is advanced (:984): the
//     linearisation point lags one accepted step behind;
//   * resOld = energy / num with no num==0 guard (:951-952); resNew = 1e6 when num == 0 (:972-975);
//   * the function falls off its end without a return value (:1003); the caller ignores it (FullSystem.cpp:488);
//   * d_xi_x[4] = 1 - u^2 and d_xi_y[3] = -(1 - v^2) (:919,:925) where the true derivatives carry a + (calcGSSSE has it right,
//     :457,:460); kept as written, documented by tests/test_oracle_struct_pose.py::test_jacobian_vs_true_derivative.
// Eigen expression order assumed where it only affects the last bit of a double: (J^T J) * weight, (J^T res) * weight (:943-944).
// `sqrt(sqrt(limit/lambda))` (:963) is evaluated in float here (the argument is a float; result is stored to float).
struct StructPt { float wx, wy, wz; float ox, oy; };   // point2world(...) of the map point, observed pixel (cast to float)

static void struct_points(const Tracker* T, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                          const double* host_pose7, const double* obs, std::vector<StructPt>* out) {
    out->resize(n);
    const float cx = T->cx[0], cy = T->cy[0], fxi = T->Ki[0][0], fyi = T->Ki[0][4];
    for (int i = 0; i < n; ++i) {
        const double* hp = host_pose7 + 7 * host_idx[i];
        double Rd[9]; quat_to_R(hp, Rd);
        float R[9], t[3];
        for (int k = 0; k < 9; ++k) R[k] = (float)Rd[k];
        for (int k = 0; k < 3; ++k) t[k] = (float)hp[4 + k];
        // point2world (ResidualProjections.h:61-78): KliP / idepth is an element-wise division
        const float k0 = (u[i] + 0 - cx) * fxi, k1 = (v[i] + 0 - cy) * fyi, k2 = 1;
        const float p0 = k0 / idepth[i], p1 = k1 / idepth[i], p2 = k2 / idepth[i];
        StructPt& P = (*out)[i];
        P.wx = ((R[0] * p0 + R[1] * p1) + R[2] * p2) + t[0];
        P.wy = ((R[3] * p0 + R[4] * p1) + R[5] * p2) + t[1];
        P.wz = ((R[6] * p0 + R[7] * p1) + R[8] * p2) + t[2];
        P.ox = (float)obs[2 * i]; P.oy = (float)obs[2 * i + 1];
    }
}

struct StructCalib { float fx, fy, cx, cy, fxi, fyi, wM3G, hM3G; };

// world2frame (ResidualProjections.h:80-94)
static inline bool world2frame(const StructPt& P, const StructCalib& C, const float* R, const float* t, float pf[3], float& Ku, float& Kv) {
    pf[0] = ((R[0] * P.wx + R[1] * P.wy) + R[2] * P.wz) + t[0];
    pf[1] = ((R[3] * P.wx + R[4] * P.wy) + R[5] * P.wz) + t[1];
    pf[2] = ((R[6] * P.wx + R[7] * P.wy) + R[8] * P.wz) + t[2];
    const float u0 = pf[0] / pf[2], u1 = pf[1] / pf[2];
    Ku = u0 * C.fx + C.cx;
    Kv = u1 * C.fy + C.cy;
    return Ku > 1.1f && Kv > 1.1f && Ku < C.wM3G && Kv < C.hM3G;
}

static void pose_to_Rt_f(const SE3& T, float R[9], float t[3]) {
    double Rd[9]; quat_to_R(T.q, Rd);
    for (int k = 0; k < 9; ++k) R[k] = (float)Rd[k];
    for (int k = 0; k < 3; ++k) t[k] = (float)T.t[k];
}

// calculateRes (:840-872): float energy summed in point order
static float struct_calc_res(const std::vector<StructPt>& pts, const StructCalib& C, const SE3& worldToCur, int& num) {
    float R[9], t[3]; pose_to_Rt_f(worldToCur, R, t);
    float energy = 0.0f;
    num = 0;
    for (const StructPt& P : pts) {
        float pf[3], Ku, Kv;
        if (world2frame(P, C, R, t, pf, Ku, Kv)) {
            const float r0 = Ku - P.ox, r1 = Kv - P.oy;
            energy = energy + r0 * r0 + r1 * r1;
            num++;
        }
    }
    return energy;
}

// calculateWeight (:874-889)
static inline float struct_weight(float x) {
    const float b = 4.6851f;
    const float b_square = b * b;
    const float x_square = x * x;
    if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
    return 0.0f;
}

// calcHandb (:891-947): H_out/b_out are ADDED to
static void struct_calc_Hb(const std::vector<StructPt>& pts, const StructCalib& C, const SE3& worldToCur, double H[36], double b[6]) {
    float R[9], t[3]; pose_to_Rt_f(worldToCur, R, t);
    for (const StructPt& P : pts) {
        float pf[3], Ku, Kv;
        if (!world2frame(P, C, R, t, pf, Ku, Kv)) continue;
        float jx[6], jy[6];
        jx[0] = (float)(1.0 / (double)pf[2]);
        jx[1] = 0.0f;
        jx[2] = -pf[0] / (pf[2] * pf[2]);
        jx[3] = jx[2] * pf[1];
        jx[4] = 1 + pf[0] * jx[2];
        jx[5] = -pf[1] / pf[2];
        jy[0] = 0.0f;
        jy[1] = (float)(1.0 / (double)pf[2]);
        jy[2] = -pf[1] / (pf[2] * pf[2]);
        jy[3] = -(1 + pf[1] * jy[2]);
        jy[4] = -jx[3];
        jy[5] = pf[0] / pf[2];
        const float up = (Ku - C.cx) * C.fxi, vp = (Kv - C.cy) * C.fyi;          // pixel2unit (ResidualProjections.h:96-103)
        const float uo = (P.ox - C.cx) * C.fxi, vo = (P.oy - C.cy) * C.fyi;
        const float r0 = up - uo, r1 = vp - vo;
        const double weight = (double)struct_weight(sqrtf(r0 * r0 + r1 * r1));
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j)
                H[6 * i + j] += ((double)jx[i] * (double)jx[j] + (double)jy[i] * (double)jy[j]) * weight;
            b[i] += ((double)jx[i] * (double)r0 + (double)jy[i] * (double)r1) * weight;
        }
    }
}

static const int kStructTraceStride = 16;   // [it, lambda, resOld, resNew, accept, inc(6), num, extrapFac, |inc|, 0, 0]

static int struct_pose_estimation(const Tracker* T, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                                  const double* host_pose7, const double* obs, SE3& curToWorld, double* trace, double* final_res) {
    std::vector<StructPt> pts;
    struct_points(T, n, u, v, idepth, host_idx, host_pose7, obs, &pts);
    StructCalib C{T->fx[0], T->fy[0], T->cx[0], T->cy[0], T->Ki[0][0], T->Ki[0][4], (float)(T->w[0] - 3), (float)(T->h[0] - 3)};
    SE3 worldToCur_current = se3_inverse(curToWorld);
    float lambda = 0.01f;
    const float lambdaExtrapolationLimit = 0.001f;
    double H[36] = {0}, b[6] = {0};
    int num;
    float resNew = 0.0f;
    float resOld = struct_calc_res(pts, C, worldToCur_current, num);
    resOld = resOld / num;
    struct_calc_Hb(pts, C, worldToCur_current, H, b);
    int its = 0;
    for (int iteration = 0; iteration < 10; ++iteration) {
        const float lambda_used = lambda;
        for (int i = 0; i < 6; ++i) H[7 * i] *= (double)(1 + lambda);
        double nb[6], inc[6];
        for (int i = 0; i < 6; ++i) nb[i] = -b[i];
        ldlt_solve(6, H, nb, inc);
        float extrapFac = 1;
        if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
        for (int i = 0; i < 6; ++i) inc[i] *= (double)extrapFac;
        const SE3 worldToCur_new = se3_mul(se3_exp(inc), worldToCur_current);
        resNew = struct_calc_res(pts, C, worldToCur_new, num);
        if (num == 0) resNew = 1000000.0f;
        else resNew = resNew / num;
        const bool accept = (resNew < resOld);
        double incn = 0;
        for (int i = 0; i < 6; ++i) incn += inc[i] * inc[i];
        incn = std::sqrt(incn);
        if (trace) {
            double* tr = trace + (size_t)kStructTraceStride * its;
            tr[0] = iteration; tr[1] = lambda_used; tr[2] = resOld; tr[3] = resNew; tr[4] = accept ? 1 : 0;
            for (int i = 0; i < 6; ++i) tr[5 + i] = inc[i];
            tr[11] = num; tr[12] = extrapFac; tr[13] = incn; tr[14] = tr[15] = 0;
        }
        ++its;
        if (accept) {
            for (int i = 0; i < 36; ++i) H[i] = 0;
            for (int i = 0; i < 6; ++i) b[i] = 0;
            resOld = resNew;
            resNew = 0;
            struct_calc_Hb(pts, C, worldToCur_current, H, b);     // sic: the pose BEFORE this step (:983)
            worldToCur_current = worldToCur_new;
            curToWorld = se3_inverse(worldToCur_new);
            lambda *= 0.5f;
        } else {
            lambda *= 4;
            if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
        }
        if (!(incn > 1e-5)) break;
    }
    if (final_res) *final_res = resOld;
    return its;
}

// ---- a3 / 8f-4: makeCoarseDepthL0 and makeCoarseDepthForFirstFrame (CoarseTracker.cpp:108-256, 258-425) --------------------
// Both functions are: splat (u, v, new_idepth, weight) tuples into the level-0 idepth / weightSums maps in point order, sum the
// maps down the pyramid, dilate (diagonal neighbours on levels 0-1, 4-neighbours above), normalise and collect the reference
// template pc_u / pc_v / pc_idepth / pc_color in raster order.  They differ only in WHICH tuples are splat (:266-293 vs :114-125),
// which stays with the caller.  Deviation: the dilation loops read one element before / after the maps at the first / last
// processed pixel (i-1-wl = -1 at i = wl on levels 0-1; i+1+wl = w*h at i = wh-1), undefined behaviour in the reference -- here
// (and in the HIP kernels) an out-of-range neighbour counts as "no value".
static void make_coarse_depth(Tracker* T, int n, const int* pu, const int* pv, const float* new_idepth, const float* weight) {
    const int L = T->levels;
    std::vector<std::vector<float>> idepth(L), wsum(L), wbak(L);
    for (int l = 0; l < L; ++l) { idepth[l].assign((size_t)T->w[l] * T->h[l], 0.f); wsum[l].assign((size_t)T->w[l] * T->h[l], 0.f); }
    for (int i = 0; i < n; ++i) {
        const size_t k = (size_t)pu[i] + (size_t)T->w[0] * pv[i];
        idepth[0][k] += new_idepth[i] * weight[i];
        wsum[0][k] += weight[i];
    }
    for (int lvl = 1; lvl < L; lvl++) {
        const int wl = T->w[lvl], hl = T->h[lvl], wlm1 = T->w[lvl - 1];
        for (int y = 0; y < hl; y++)
            for (int x = 0; x < wl; x++) {
                const int b = 2 * x + 2 * y * wlm1;
                idepth[lvl][x + y * wl] = idepth[lvl - 1][b] + idepth[lvl - 1][b + 1] + idepth[lvl - 1][b + wlm1] + idepth[lvl - 1][b + wlm1 + 1];
                wsum[lvl][x + y * wl] = wsum[lvl - 1][b] + wsum[lvl - 1][b + 1] + wsum[lvl - 1][b + wlm1] + wsum[lvl - 1][b + wlm1 + 1];
            }
    }
    for (int lvl = 0; lvl < L; lvl++) {
        const int wl = T->w[lvl], total = T->w[lvl] * T->h[lvl], wh = total - wl;
        wbak[lvl] = wsum[lvl];
        const std::vector<float>& bak = wbak[lvl];
        std::vector<float>& id = idepth[lvl];
        const int off[2][4] = {{1 + wl, -1 - wl, wl - 1, -wl + 1}, {1, -1, wl, -wl}};
        const int* o = off[lvl < 2 ? 0 : 1];
        for (int i = wl; i < wh; i++) {
            if (bak[i] <= 0) {
                float sum = 0, num = 0, numn = 0;
                for (int q = 0; q < 4; ++q) {
                    const int j = i + o[q];
                    if (j >= 0 && j < total && bak[j] > 0) { sum += id[j]; num += bak[j]; numn++; }
                }
                if (numn > 0) { id[i] = sum / numn; wsum[lvl][i] = num / numn; }
            }
        }
    }
    for (int lvl = 0; lvl < L; lvl++) {
        const int wl = T->w[lvl], hl = T->h[lvl];
        const float* dIRef = T->dIp[lvl].data();
        T->pc_u[lvl].clear(); T->pc_v[lvl].clear(); T->pc_idepth[lvl].clear(); T->pc_color[lvl].clear();
        for (int y = 2; y < hl - 2; y++)
            for (int x = 2; x < wl - 2; x++) {
                const int i = x + y * wl;
                if (wsum[lvl][i] > 0) {
                    idepth[lvl][i] /= wsum[lvl][i];
                    const float c = dIRef[3 * i];
                    if (!std::isfinite(c) || !(idepth[lvl][i] > 0)) { idepth[lvl][i] = -1; continue; }
                    T->pc_u[lvl].push_back((float)x); T->pc_v[lvl].push_back((float)y);
                    T->pc_idepth[lvl].push_back(idepth[lvl][i]); T->pc_color[lvl].push_back(c);
                } else idepth[lvl][i] = -1;
                wsum[lvl][i] = 1;
            }
        T->pc_n[lvl] = (int)T->pc_u[lvl].size();
    }
}

extern "C" {

int orc_trace_stride() { return kTraceStride; }

void* orc_tracker_create(int w0, int h0, int levels) {
    Tracker* T = new Tracker();
    T->levels = levels;
    for (int l = 0; l < levels; ++l) { T->w[l] = w0 >> l; T->h[l] = h0 >> l; T->pc_n[l] = 0; }
    T->ref_exposure = T->new_exposure = 1; T->ref_a = T->ref_b = 0;
    T->huberTH = 6; T->coarseCutoffTH = 20; T->affineOptModeA = 0; T->affineOptModeB = 0;
    T->bw_n = 0;
    return T;
}
void orc_tracker_destroy(void* h) { delete (Tracker*)h; }
void orc_tracker_set_settings(void* h, float huberTH, float coarseCutoffTH, float affA, float affB) {
    Tracker* T = (Tracker*)h; T->huberTH = huberTH; T->coarseCutoffTH = coarseCutoffTH; T->affineOptModeA = affA; T->affineOptModeB = affB;
}
void orc_tracker_make_K(void* h, float fx, float fy, float cx, float cy) { make_K((Tracker*)h, fx, fy, cx, cy); }
void orc_tracker_get_K(void* h, int lvl, float out4[4], float Ki9[9]) {
    Tracker* T = (Tracker*)h;
    out4[0] = T->fx[lvl]; out4[1] = T->fy[lvl]; out4[2] = T->cx[lvl]; out4[3] = T->cy[lvl];
    std::memcpy(Ki9, T->Ki[lvl], sizeof(float) * 9);
}
void orc_tracker_set_ref(void* h, int lvl, int n, const float* u, const float* v, const float* idepth, const float* color) {
    Tracker* T = (Tracker*)h;
    T->pc_n[lvl] = n;
    T->pc_u[lvl].assign(u, u + n); T->pc_v[lvl].assign(v, v + n);
    T->pc_idepth[lvl].assign(idepth, idepth + n); T->pc_color[lvl].assign(color, color + n);
}
// reference template from splat tuples; the reference frame's pyramid (lastRef->dIp) is the one set with set_new_image / set_new_pyr
void orc_tracker_make_coarse_depth(void* h, int n, const int* u, const int* v, const float* new_idepth, const float* weight) {
    make_coarse_depth((Tracker*)h, n, u, v, new_idepth, weight);
}
int orc_tracker_get_ref(void* h, int lvl, float* u, float* v, float* idepth, float* color) {
    Tracker* T = (Tracker*)h;
    const int n = T->pc_n[lvl];
    if (u) for (int i = 0; i < n; ++i) { u[i] = T->pc_u[lvl][i]; v[i] = T->pc_v[lvl][i]; idepth[i] = T->pc_idepth[lvl][i]; color[i] = T->pc_color[lvl][i]; }
    return n;
}
void orc_tracker_set_ref_frame(void* h, float exposure, double a, double b) {
    Tracker* T = (Tracker*)h; T->ref_exposure = exposure; T->ref_a = a; T->ref_b = b;
}
// new frame from a raw level-0 image (runs makeImages)
void orc_tracker_set_new_image(void* h, const float* color, float exposure) {
    Tracker* T = (Tracker*)h;
    make_images(color, T->w[0], T->h[0], T->levels, T->dIp);
    T->new_exposure = exposure;
}
// new frame from an existing AoS pyramid level
void orc_tracker_set_new_pyr(void* h, int lvl, const float* dIp_aos3, float exposure) {
    Tracker* T = (Tracker*)h;
    T->dIp[lvl].assign(dIp_aos3, dIp_aos3 + (size_t)T->w[lvl] * T->h[lvl] * 3);
    T->new_exposure = exposure;
}
void orc_tracker_get_pyr(void* h, int lvl, float* out) {
    Tracker* T = (Tracker*)h;
    std::memcpy(out, T->dIp[lvl].data(), sizeof(float) * T->dIp[lvl].size());
}
void orc_make_images(const float* color, int w0, int h0, int levels, float* out_concat) {
    std::vector<float> tmp[kMaxLvl];
    make_images(color, w0, h0, levels, tmp);
    size_t off = 0;
    for (int l = 0; l < levels; ++l) { std::memcpy(out_concat + off, tmp[l].data(), sizeof(float) * tmp[l].size()); off += tmp[l].size(); }
}
// pose7 = Sophus SE3d::data() layout [qx qy qz qw tx ty tz]
void orc_calc_res(void* h, int lvl, const double pose7[7], double a, double b, float cutoffTH, double out6[6]) {
    SE3 T; std::memcpy(T.q, pose7, 4 * sizeof(double)); std::memcpy(T.t, pose7 + 4, 3 * sizeof(double));
    calc_res((Tracker*)h, lvl, T, a, b, cutoffTH, out6);
}
int orc_get_warped(void* h, float* out8) {  // 8 planes x bw_n, plane-major: idepth,u,v,dx,dy,res,w,ref
    Tracker* T = (Tracker*)h; const int n = T->bw_n;
    const std::vector<float>* p[8] = {&T->bw_idepth, &T->bw_u, &T->bw_v, &T->bw_dx, &T->bw_dy, &T->bw_res, &T->bw_w, &T->bw_ref};
    if (out8) for (int k = 0; k < 8; ++k) std::memcpy(out8 + (size_t)k * n, p[k]->data(), sizeof(float) * n);
    return n;
}
void orc_calc_gs(void* h, int lvl, double a, double b, double H64[64], double b8[8]) { calc_gs((Tracker*)h, lvl, a, b, H64, b8); }
int orc_track(void* h, double pose7_io[7], double aff_io[2], int coarsestLvl, const double minRes[5], double lastRes[5],
              double flow[3], double* trace, int trace_cap, int* trace_n) {
    Tracker* T = (Tracker*)h;
    SE3 P; std::memcpy(P.q, pose7_io, 4 * sizeof(double)); std::memcpy(P.t, pose7_io + 4, 3 * sizeof(double));
    bool ok = track(T, P, aff_io, coarsestLvl, minRes, trace, trace_cap, trace_n);
    std::memcpy(pose7_io, P.q, 4 * sizeof(double)); std::memcpy(pose7_io + 4, P.t, 3 * sizeof(double));
    for (int i = 0; i < 5; ++i) lastRes[i] = T->lastResiduals[i];
    for (int i = 0; i < 3; ++i) flow[i] = T->lastFlowIndicators[i];
    return ok ? 1 : 0;
}

int orc_struct_trace_stride() { return kStructTraceStride; }
// returns the number of iterations run; pose7_io = curToWorld (Sophus data() layout), updated only by accepted steps
int orc_struct_pose(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                    const double* host_pose7, const double* obs, double pose7_io[7], double* trace, double* final_res) {
    SE3 P; std::memcpy(P.q, pose7_io, 32); std::memcpy(P.t, pose7_io + 4, 24);
    const int its = struct_pose_estimation((const Tracker*)h, n, u, v, idepth, host_idx, host_pose7, obs, P, trace, final_res);
    std::memcpy(pose7_io, P.q, 32); std::memcpy(pose7_io + 4, P.t, 24);
    return its;
}
// calcHandb + calculateRes at one pose (parity hooks)
void orc_struct_res_Hb(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                       const double* host_pose7, const double* obs, const double worldToCur7[7], double H36[36], double b6[6],
                       double* energy, int* num) {
    const Tracker* T = (const Tracker*)h;
    std::vector<StructPt> pts;
    struct_points(T, n, u, v, idepth, host_idx, host_pose7, obs, &pts);
    StructCalib C{T->fx[0], T->fy[0], T->cx[0], T->cy[0], T->Ki[0][0], T->Ki[0][4], (float)(T->w[0] - 3), (float)(T->h[0] - 3)};
    SE3 P; std::memcpy(P.q, worldToCur7, 32); std::memcpy(P.t, worldToCur7 + 4, 24);
    for (int i = 0; i < 36; ++i) H36[i] = 0;
    for (int i = 0; i < 6; ++i) b6[i] = 0;
    struct_calc_Hb(pts, C, P, H36, b6);
    int nn; *energy = (double)struct_calc_res(pts, C, P, nn); *num = nn;
}

// ---- maths helpers exposed for the known-answer tests ----
void orc_se3_exp(const double a[6], double pose7[7]) { SE3 T = se3_exp(a); std::memcpy(pose7, T.q, 32); std::memcpy(pose7 + 4, T.t, 24); }
void orc_se3_log(const double pose7[7], double a[6]) { SE3 T; std::memcpy(T.q, pose7, 32); std::memcpy(T.t, pose7 + 4, 24); se3_log(T, a); }
void orc_se3_mul(const double A[7], const double B[7], double out[7]) {
    SE3 a, b; std::memcpy(a.q, A, 32); std::memcpy(a.t, A + 4, 24); std::memcpy(b.q, B, 32); std::memcpy(b.t, B + 4, 24);
    SE3 r = se3_mul(a, b); std::memcpy(out, r.q, 32); std::memcpy(out + 4, r.t, 24);
}
void orc_se3_inverse(const double A[7], double out[7]) {
    SE3 a; std::memcpy(a.q, A, 32); std::memcpy(a.t, A + 4, 24); SE3 r = se3_inverse(a); std::memcpy(out, r.q, 32); std::memcpy(out + 4, r.t, 24);
}
void orc_se3_matrix(const double A[7], double R9[9]) { quat_to_R(A, R9); }
void orc_se3_adj(const double A[7], double out36[36]) { SE3 a; std::memcpy(a.q, A, 32); std::memcpy(a.t, A + 4, 24); se3_adj(a, out36); }
void orc_ldlt_solve(int n, const double* A, const double* b, double* x) { ldlt_solve(n, A, b, x); }
void orc_inv3f(const float* m, float* out) { inv3f(m, out); }
void orc_interp33(const float* mat, float x, float y, int width, float out[3]) { interp33(mat, x, y, width, out); }

// ---- hooks with the signatures of oracle/ref_glue.cpp: the restatements above, driven like the reference's own classes are there
// (tests/test_ref_pin.py compares the two bit for bit) ----
void orc_kat_aff_from_to(float exposureF, float exposureT, double aF, double bF, double aT, double bT, double* ab) {
    aff_from_to(exposureF, exposureT, aF, bF, aT, bT, ab);
}
void orc_kat_interp33(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
    for (int i = 0; i < n; ++i) interp33(img3, x[i], y[i], width, out3 + 3 * i);
}
void orc_kat_acc9(int n4, const float* J, const float* w, float* H81, double* num) {
    Accumulator9* acc = new Accumulator9();
    acc->initialize();
    const size_t N = (size_t)4 * n4;
    for (int g = 0; g < n4; ++g) {
        float Jg[9][4], wg[4];
        for (int k = 0; k < 9; ++k) for (int l = 0; l < 4; ++l) Jg[k][l] = J[k * N + 4 * g + l];
        for (int l = 0; l < 4; ++l) wg[l] = w[4 * g + l];
        acc->updateWeighted(Jg, wg);
    }
    acc->finish();
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) H81[a * 9 + b] = acc->H[a][b];
    *num = (double)acc->num;
    delete acc;
}
}
