// oracle/orc_trace.cpp -- TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY PINNED against the reference's own translation units (oracle/_ref/libref.so, oracle/README.md; tests/test_ref_pin*.py).
//
// Plain C++ restatement of ImmaturePoint::traceOn (src/FullSystem/ImmaturePoint.cpp:47-353), SURVEY.md section 8f row 4: the
// epipolar-line search every immature point of every key-frame runs on each new frame (FullSystem::traceNewCoarse,
// FullSystem.cpp:519-553).  float32 arithmetic, operand order and double-literal promotions as in the reference; -ffp-contract=off.
// Settings (src/util/settings.cpp): maxPixSearch 0.027 (:130), minTraceTestRadius 2 (:132), trace_stepsize 1 (:134),
// trace_GNIterations 3 (:135), trace_GNThreshold 0.1 (:136), trace_extraSlackOnTH 1.2 (:137), trace_slackInterval 1.5 (:138),
// trace_minImprovementFactor 2 (:139), huberTH 6 (:101); pattern 8 (settings.h:174-175, settings.cpp:250).
// The reference has no tests for this path (oracle/README.md); pinned by tests/test_oracle_trace.py.
#include "orc_math.hpp"
#include <cstdint>

namespace orc {

enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };   // ImmaturePoint.h:20-30

struct TracePt {          // the ImmaturePoint fields traceOn reads / writes (ImmaturePoint.h:33-75)
    float u, v;
    float idepth_min, idepth_max;
    float quality;
    int lastTraceStatus;
    float energyTH;
    float gradH[4];       // Mat22f, row-major (symmetric)
    float color[8], weights[8];
    float lastTraceUV[2];
    float lastTracePixelInterval;
};

static inline float interp31(const float* mat, float x, float y, int width) {   // getInterpolatedElement31, globalFuncs.h:102-116
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    return ((dxdy * bp[3 + 3 * width] + (dy - dxdy) * bp[3 * width]) + (dx - dxdy) * bp[3]) + (1 - dx - dy + dxdy) * bp[0];
}
static inline void interp33t(const float* mat, float x, float y, int width, float* o) {   // getInterpolatedElement33
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    for (int c = 0; c < 3; ++c)
        o[c] = ((dxdy * bp[3 + 3 * width + c] + (dy - dxdy) * bp[3 * width + c]) + (dx - dxdy) * bp[3 + c]) + (1 - dx - dy + dxdy) * bp[c];
}

static int trace_on(TracePt& P, const float* dI, int w, int h, const float* KRKi, const float* Kt, const float* aff) {
    const float setting_maxPixSearch = 0.027f, setting_trace_stepsize = 1.0f, setting_trace_GNThreshold = 0.1f, setting_trace_extraSlackOnTH = 1.2f,
                setting_trace_slackInterval = 1.5f, setting_trace_minImprovementFactor = 2, setting_huberTH = 6;
    const int setting_minTraceTestRadius = 2, setting_trace_GNIterations = 3;
    static const int pat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};
    if (P.lastTraceStatus == IPS_OOB) return P.lastTraceStatus;
    const float maxPixSearch = (w + h) * setting_maxPixSearch;
    const float pr[3] = {(KRKi[0] * P.u + KRKi[1] * P.v) + KRKi[2] * 1.0f, (KRKi[3] * P.u + KRKi[4] * P.v) + KRKi[5] * 1.0f,
                         (KRKi[6] * P.u + KRKi[7] * P.v) + KRKi[8] * 1.0f};
    const float ptpMin[3] = {pr[0] + Kt[0] * P.idepth_min, pr[1] + Kt[1] * P.idepth_min, pr[2] + Kt[2] * P.idepth_min};
    const float uMin = ptpMin[0] / ptpMin[2], vMin = ptpMin[1] / ptpMin[2];
    auto oob = [&]() { P.lastTraceUV[0] = P.lastTraceUV[1] = -1; P.lastTracePixelInterval = 0; return P.lastTraceStatus = IPS_OOB; };
    if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) return oob();
    float dist, uMax, vMax;
    if (std::isfinite(P.idepth_max)) {
        const float ptpMax[3] = {pr[0] + Kt[0] * P.idepth_max, pr[1] + Kt[1] * P.idepth_max, pr[2] + Kt[2] * P.idepth_max};
        uMax = ptpMax[0] / ptpMax[2];
        vMax = ptpMax[1] / ptpMax[2];
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) return oob();
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf(dist);
        if (dist < setting_trace_slackInterval) {
            P.lastTraceUV[0] = (uMax + uMin) * 0.5f; P.lastTraceUV[1] = (vMax + vMin) * 0.5f;
            P.lastTracePixelInterval = dist;
            return P.lastTraceStatus = IPS_SKIPPED;
        }
    } else {
        dist = maxPixSearch;
        const float ptpMax[3] = {pr[0] + Kt[0] * 0.01f, pr[1] + Kt[1] * 0.01f, pr[2] + Kt[2] * 0.01f};
        uMax = ptpMax[0] / ptpMax[2];
        vMax = ptpMax[1] / ptpMax[2];
        const float dx = uMax - uMin, dy = vMax - vMin;
        const float d = 1.0f / sqrtf(dx * dx + dy * dy);
        uMax = uMin + dist * dx * d;
        vMax = vMin + dist * dy * d;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) return oob();
    }
    if (!(P.idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) return oob();

    float dx = setting_trace_stepsize * (uMax - uMin);
    float dy = setting_trace_stepsize * (vMax - vMin);
    const float* G = P.gradH;
    const float a = (dx * G[0] + dy * G[2]) * dx + (dx * G[1] + dy * G[3]) * dy;             // (dx,dy)^T gradH (dx,dy)
    const float b = (dy * G[0] + (-dx) * G[2]) * dy + (dy * G[1] + (-dx) * G[3]) * (-dx);    // (dy,-dx)^T gradH (dy,-dx)
    float errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * setting_trace_minImprovementFactor > dist && std::isfinite(P.idepth_max)) {
        P.lastTraceUV[0] = (uMax + uMin) * 0.5f; P.lastTraceUV[1] = (vMax + vMin) * 0.5f;
        P.lastTracePixelInterval = dist;
        return P.lastTraceStatus = IPS_BADCONDITION;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    dx /= dist;
    dy /= dist;
    if (dist > maxPixSearch) {
        uMax = uMin + maxPixSearch * dx;
        vMax = vMin + maxPixSearch * dy;
        dist = maxPixSearch;
    }
    int numSteps = (int)(1.9999f + dist / setting_trace_stepsize);
    const float randShift = uMin * 1000 - floorf(uMin * 1000);
    float ptx = uMin - randShift * dx;
    float pty = vMin - randShift * dy;
    float rp[8][2];
    for (int idx = 0; idx < 8; ++idx) {
        rp[idx][0] = KRKi[0] * pat[idx][0] + KRKi[1] * pat[idx][1];
        rp[idx][1] = KRKi[3] * pat[idx][0] + KRKi[4] * pat[idx][1];
    }
    if (!std::isfinite(dx) || !std::isfinite(dy)) { P.lastTracePixelInterval = 0; P.lastTraceUV[0] = P.lastTraceUV[1] = -1; return P.lastTraceStatus = IPS_OOB; }

    float errors[100];
    float bestU = 0, bestV = 0, bestEnergy = 1e10f;
    int bestIdx = -1;
    if (numSteps >= 100) numSteps = 99;
    for (int i = 0; i < numSteps; i++) {
        float energy = 0;
        for (int idx = 0; idx < 8; idx++) {
            const float hitColor = interp31(dI, (float)(ptx + rp[idx][0]), (float)(pty + rp[idx][1]), w);
            if (!std::isfinite(hitColor)) { energy = (float)(energy + 1e5); continue; }
            const float residual = hitColor - (float)(aff[0] * P.color[idx] + aff[1]);
            const float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
            energy += hw * residual * residual * (2 - hw);
        }
        errors[i] = energy;
        if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
        ptx += dx;
        pty += dy;
    }
    float secondBest = 1e10f;
    for (int i = 0; i < numSteps; i++)
        if ((i < bestIdx - setting_minTraceTestRadius || i > bestIdx + setting_minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
    const float newQuality = secondBest / bestEnergy;
    if (newQuality < P.quality || numSteps > 10) P.quality = newQuality;

    float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
    if (setting_trace_GNIterations > 0) bestEnergy = 1e5f;
    for (int it = 0; it < setting_trace_GNIterations; it++) {
        float H = 1, bb = 0, energy = 0;
        for (int idx = 0; idx < 8; idx++) {
            float hc[3];
            interp33t(dI, (float)(bestU + rp[idx][0]), (float)(bestV + rp[idx][1]), w, hc);
            if (!std::isfinite(hc[0])) { energy = (float)(energy + 1e5); continue; }
            const float residual = hc[0] - (aff[0] * P.color[idx] + aff[1]);
            const float dResdDist = dx * hc[1] + dy * hc[2];
            const float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);
            H += hw * dResdDist * dResdDist;
            bb += hw * residual * dResdDist;
            energy += P.weights[idx] * P.weights[idx] * hw * residual * residual * (2 - hw);
        }
        if (energy > bestEnergy) {
            stepBack *= 0.5f;
            bestU = uBak + stepBack * dx;
            bestV = vBak + stepBack * dy;
        } else {
            float step = -gnstepsize * bb / H;
            if (step < -0.5) step = -0.5f;
            else if (step > 0.5) step = 0.5f;
            if (!std::isfinite(step)) step = 0;
            uBak = bestU;
            vBak = bestV;
            stepBack = step;
            bestU += step * dx;
            bestV += step * dy;
            bestEnergy = energy;
        }
        if (fabsf(stepBack) < setting_trace_GNThreshold) break;
    }
    if (!(bestEnergy < P.energyTH * setting_trace_extraSlackOnTH)) {
        P.lastTracePixelInterval = 0;
        P.lastTraceUV[0] = P.lastTraceUV[1] = -1;
        if (P.lastTraceStatus == IPS_OUTLIER) return P.lastTraceStatus = IPS_OOB;
        return P.lastTraceStatus = IPS_OUTLIER;
    }
    if (dx * dx > dy * dy) {
        P.idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        P.idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        P.idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        P.idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (P.idepth_min > P.idepth_max) std::swap(P.idepth_min, P.idepth_max);
    if (!std::isfinite(P.idepth_min) || !std::isfinite(P.idepth_max) || (P.idepth_max < 0)) {
        P.lastTracePixelInterval = 0;
        P.lastTraceUV[0] = P.lastTraceUV[1] = -1;
        return P.lastTraceStatus = IPS_OUTLIER;
    }
    P.lastTracePixelInterval = 2 * errorInPixel;
    P.lastTraceUV[0] = bestU; P.lastTraceUV[1] = bestV;
    return P.lastTraceStatus = IPS_GOOD;
}

}  // namespace orc

using namespace orc;

extern "C" {
// hooks with the signatures of oracle/ref_glue.cpp (tests/test_ref_pin.py)
void orc_kat_interp31_trace(const float* img3, int width, int n, const float* x, const float* y, float* out) {
    for (int i = 0; i < n; ++i) out[i] = interp31(img3, x[i], y[i], width);
}
void orc_kat_interp33_trace(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
    for (int i = 0; i < n; ++i) interp33t(img3, x[i], y[i], width, out3 + 3 * i);
}


// points: SoA in/out.  state[n][5] = {idepth_min, idepth_max, quality, lastTraceUV.x, lastTraceUV.y} and interval[n], status[n] are
// updated in place; host_idx selects the per-host KRKi (9) / Kt (3) / aff (2) the caller computed as FullSystem::traceNewCoarse does.
void orc_trace_on(int n, const float* u, const float* v, const float* energyTH, const float* gradH4, const float* color8, const float* weights8,
                  const int* host_idx, const float* KRKi9, const float* Kt3, const float* aff2, const float* dI_aos3, int w, int h,
                  float* idepth_min, float* idepth_max, float* quality, int* status, float* lastTraceUV2, float* lastTracePixelInterval) {
    for (int i = 0; i < n; ++i) {
        TracePt P;
        P.u = u[i]; P.v = v[i]; P.idepth_min = idepth_min[i]; P.idepth_max = idepth_max[i]; P.quality = quality[i];
        P.lastTraceStatus = status[i]; P.energyTH = energyTH[i];
        for (int k = 0; k < 4; ++k) P.gradH[k] = gradH4[4 * i + k];
        for (int k = 0; k < 8; ++k) { P.color[k] = color8[8 * i + k]; P.weights[k] = weights8[8 * i + k]; }
        P.lastTraceUV[0] = lastTraceUV2[2 * i]; P.lastTraceUV[1] = lastTraceUV2[2 * i + 1];
        P.lastTracePixelInterval = lastTracePixelInterval[i];
        const int hh = host_idx[i];
        trace_on(P, dI_aos3, w, h, KRKi9 + 9 * hh, Kt3 + 3 * hh, aff2 + 2 * hh);
        idepth_min[i] = P.idepth_min; idepth_max[i] = P.idepth_max; quality[i] = P.quality; status[i] = P.lastTraceStatus;
        lastTraceUV2[2 * i] = P.lastTraceUV[0]; lastTraceUV2[2 * i + 1] = P.lastTraceUV[1];
        lastTracePixelInterval[i] = P.lastTracePixelInterval;
    }
}

}  // extern "C"
