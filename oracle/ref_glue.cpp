// oracle/ref_glue.cpp -- TEST INFRASTRUCTURE.  C entry points around code of the REFERENCE ITSELF: this file includes, unmodified and from
// where they lie under /root/reference, the three reference headers whose hot-path code has no dependency beyond Eigen's storage types --
//     src/util/NumType.h                                AffLight::fromToVecExposure                      
//     src/util/globalFuncs.h                            getInterpolatedElement33 / 31 / 33BiLin          (row a9)
//     src/OptimizationBackend/MatrixAccumulators.h      Accumulator9, AccumulatorApprox, Accumulator11                   (rows a6, b3)
// -- and, linked beside it, the reference's src/util/settings.cpp (every setting_* constant of the hot path; needs an empty boost/bind.hpp)
// -- against the stand-in for Eigen's interface in oracle/ref_shim (see the header of ref_shim/Eigen/Core for what that stand-in does and
// does not do).  Built by `make -C oracle ref` into oracle/_ref/libref.so when /root/reference is present; tests/test_ref_pin.py runs the
// oracle's restatements of the same classes against it bit for bit, and tools/gen_ref_pin_golden.py stores its outputs as the
// fixture tests/golden/ref_pin.npz for machines without /root/reference.  Nothing of the reference is copied into this repository.
#include <xmmintrin.h>
#include <emmintrin.h>
#include <cstring>

#include "util/NumType.h"
#include "util/globalFuncs.h"
#include "OptimizationBackend/MatrixAccumulators.h"

using namespace sdv_loam;

extern "C" {

// AffLight::fromToVecExposure (NumType.h:149-158)
void ref_aff_from_to(float exposureF, float exposureT, double aF, double bF, double aT, double bT, double* ab) {
    const Vec2 r = AffLight::fromToVecExposure(exposureF, exposureT, AffLight(aF, bF), AffLight(aT, bT));
    ab[0] = r[0]; ab[1] = r[1];
}

// getInterpolatedElement33 / 31 / 33BiLin (globalFuncs.h:51-65, :104-118, :140-161) on an AoS {I,dx,dy} image, n query points
void ref_interp33(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
    const Eigen::Vector3f* m = reinterpret_cast<const Eigen::Vector3f*>(img3);
    for (int i = 0; i < n; ++i) {
        const Eigen::Vector3f r = getInterpolatedElement33(m, x[i], y[i], width);
        out3[3 * i] = r[0]; out3[3 * i + 1] = r[1]; out3[3 * i + 2] = r[2];
    }
}
void ref_interp31(const float* img3, int width, int n, const float* x, const float* y, float* out) {
    const Eigen::Vector3f* m = reinterpret_cast<const Eigen::Vector3f*>(img3);
    for (int i = 0; i < n; ++i) out[i] = getInterpolatedElement31(m, x[i], y[i], width);
}
void ref_interp33_bilin(const float* img3, int width, int n, const float* x, const float* y, float* out3) {
    const Eigen::Vector3f* m = reinterpret_cast<const Eigen::Vector3f*>(img3);
    for (int i = 0; i < n; ++i) {
        const Eigen::Vector3f r = getInterpolatedElement33BiLin(m, x[i], y[i], width);
        out3[3 * i] = r[0]; out3[3 * i + 1] = r[1]; out3[3 * i + 2] = r[2];
    }
}

// Accumulator9 as calcGSSSE drives it (CoarseTracker.cpp:434-466): n4 groups of 4 points, J[k*4n4 + 4g + lane] for the 9 rows k (8 Jacobian
// entries + residual), w[4g + lane]; H81 = acc.H row-major after finish(), num = acc.num
void ref_acc9(int n4, const float* J, const float* w, float* H81, double* num) {
    Accumulator9* acc = new Accumulator9();      // (heap: EIGEN_ALIGN16 members want 16-byte alignment; operator new gives it)
    acc->initialize();
    const size_t N = (size_t)4 * n4;
    for (int g = 0; g < n4; ++g) {
        __m128 r[9];
        for (int k = 0; k < 9; ++k) r[k] = _mm_loadu_ps(J + k * N + 4 * g);
        acc->updateSSE_eighted(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], _mm_loadu_ps(w + 4 * g));
    }
    acc->finish();
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) H81[a * 9 + b] = acc->H(a, b);
    *num = (double)acc->num;
    delete acc;
}

// AccumulatorApprox as AccumulatedTopHessianSSE::addPoint drives it (AccumulatedTopHessian.cpp:95-135): per residual one update(),
// one updateTopRight(), one updateBotRight().  in: [n][35] = x4(4) x6(6) y4(4) y6(6) a b c | TR00 TR10 TR01 TR11 TR02 TR12 | a00 a01 a02 a11 a12 a22
// (x4.. are used by update and updateTopRight alike, as in the reference's calls); H169 = acc.H row-major after finish()
void ref_acc_approx(int n, const float* in, float* H169, double* num) {
    AccumulatorApprox* acc = new AccumulatorApprox();
    acc->initialize();
    for (int i = 0; i < n; ++i) {
        const float* p = in + (size_t)35 * i;
        acc->update(p, p + 4, p + 10, p + 14, p[20], p[21], p[22]);
        acc->updateTopRight(p, p + 4, p + 10, p + 14, p[23], p[24], p[25], p[26], p[27], p[28]);
        acc->updateBotRight(p[29], p[30], p[31], p[32], p[33], p[34]);
    }
    acc->finish();
    for (int a = 0; a < 13; ++a) for (int b = 0; b < 13; ++b) H169[a * 13 + b] = acc->H(a, b);
    *num = (double)acc->num;
    delete acc;
}

// Accumulator11 (the float energy sums of linearizeAll / calcLEnergy): n single updates
void ref_acc11(int n, const float* vals, float* A) {
    Accumulator11* acc = new Accumulator11();
    acc->initialize();
    for (int i = 0; i < n; ++i) acc->updateSingle(vals[i]);
    acc->finish();
    *A = acc->A;
    delete acc;
}

// The values of the reference's settings (src/util/settings.cpp, compiled unmodified into this library): the constants the oracle and the
// product carry as literals are compared with these by name (tests/test_ref_pin.py).  Returns 1 and *out = value, or 0 for an unknown name.
int ref_setting(const char* name, double* out) {
    struct Entry { const char* name; double value; };
    const Entry table[] = {
    {"pyrLevelsUsed", (double)sdv_loam::pyrLevelsUsed},
    {"setting_idepthFixPrior", (double)sdv_loam::setting_idepthFixPrior},
    {"setting_idepthFixPriorMargFac", (double)sdv_loam::setting_idepthFixPriorMargFac},
    {"setting_initialRotPrior", (double)sdv_loam::setting_initialRotPrior},
    {"setting_initialTransPrior", (double)sdv_loam::setting_initialTransPrior},
    {"setting_initialAffBPrior", (double)sdv_loam::setting_initialAffBPrior},
    {"setting_initialAffAPrior", (double)sdv_loam::setting_initialAffAPrior},
    {"setting_initialCalibHessian", (double)sdv_loam::setting_initialCalibHessian},
    {"setting_solverMode", (double)sdv_loam::setting_solverMode},
    {"setting_solverModeDelta", (double)sdv_loam::setting_solverModeDelta},
    {"setting_minIdepthH_act", (double)sdv_loam::setting_minIdepthH_act},
    {"setting_minIdepthH_marg", (double)sdv_loam::setting_minIdepthH_marg},
    {"setting_maxPixSearch", (double)sdv_loam::setting_maxPixSearch},
    {"setting_desiredImmatureDensity", (double)sdv_loam::setting_desiredImmatureDensity},
    {"setting_desiredPointDensity", (double)sdv_loam::setting_desiredPointDensity},
    {"setting_minPointsRemaining", (double)sdv_loam::setting_minPointsRemaining},
    {"setting_maxLogAffFacInWindow", (double)sdv_loam::setting_maxLogAffFacInWindow},
    {"setting_minFrames", (double)sdv_loam::setting_minFrames},
    {"setting_maxFrames", (double)sdv_loam::setting_maxFrames},
    {"setting_minFrameAge", (double)sdv_loam::setting_minFrameAge},
    {"setting_maxOptIterations", (double)sdv_loam::setting_maxOptIterations},
    {"setting_minOptIterations", (double)sdv_loam::setting_minOptIterations},
    {"setting_thOptIterations", (double)sdv_loam::setting_thOptIterations},
    {"setting_outlierTH", (double)sdv_loam::setting_outlierTH},
    {"setting_outlierTHSumComponent", (double)sdv_loam::setting_outlierTHSumComponent},
    {"setting_margWeightFac", (double)sdv_loam::setting_margWeightFac},
    {"setting_GNItsOnPointActivation", (double)sdv_loam::setting_GNItsOnPointActivation},
    {"setting_minTraceQuality", (double)sdv_loam::setting_minTraceQuality},
    {"setting_minTraceTestRadius", (double)sdv_loam::setting_minTraceTestRadius},
    {"setting_reTrackThreshold", (double)sdv_loam::setting_reTrackThreshold},
    {"setting_affineOptModeA", (double)sdv_loam::setting_affineOptModeA},
    {"setting_affineOptModeB", (double)sdv_loam::setting_affineOptModeB},
    {"setting_forceAceptStep", (double)sdv_loam::setting_forceAceptStep},
    {"setting_huberTH", (double)sdv_loam::setting_huberTH},
    {"setting_frameEnergyTHConstWeight", (double)sdv_loam::setting_frameEnergyTHConstWeight},
    {"setting_frameEnergyTHN", (double)sdv_loam::setting_frameEnergyTHN},
    {"setting_frameEnergyTHFacMedian", (double)sdv_loam::setting_frameEnergyTHFacMedian},
    {"setting_overallEnergyTHWeight", (double)sdv_loam::setting_overallEnergyTHWeight},
    {"setting_coarseCutoffTH", (double)sdv_loam::setting_coarseCutoffTH},
    {"setting_trace_stepsize", (double)sdv_loam::setting_trace_stepsize},
    {"setting_trace_GNIterations", (double)sdv_loam::setting_trace_GNIterations},
    {"setting_trace_GNThreshold", (double)sdv_loam::setting_trace_GNThreshold},
    {"setting_trace_extraSlackOnTH", (double)sdv_loam::setting_trace_extraSlackOnTH},
    {"setting_trace_slackInterval", (double)sdv_loam::setting_trace_slackInterval},
    {"setting_trace_minImprovementFactor", (double)sdv_loam::setting_trace_minImprovementFactor},
    {"multiThreading", (double)sdv_loam::multiThreading}
    };
    for (const Entry& e : table) if (!std::strcmp(e.name, name)) { *out = e.value; return 1; }
    return 0;
}
// the residual pattern the reference uses (settings.h:174-176: patternNum offsets of patternP = staticPattern[8]); out[2*i], out[2*i+1]
int ref_pattern(int* out) {
    for (int i = 0; i < patternNum; ++i) { out[2 * i] = patternP[i][0]; out[2 * i + 1] = patternP[i][1]; }
    return patternNum;
}
int ref_setting_count() { return 46; }
const char* ref_setting_name(int i) {
    static const char* names[] = {"pyrLevelsUsed", "setting_idepthFixPrior", "setting_idepthFixPriorMargFac", "setting_initialRotPrior", "setting_initialTransPrior", "setting_initialAffBPrior", "setting_initialAffAPrior", "setting_initialCalibHessian", "setting_solverMode", "setting_solverModeDelta", "setting_minIdepthH_act", "setting_minIdepthH_marg", "setting_maxPixSearch", "setting_desiredImmatureDensity", "setting_desiredPointDensity", "setting_minPointsRemaining", "setting_maxLogAffFacInWindow", "setting_minFrames", "setting_maxFrames", "setting_minFrameAge", "setting_maxOptIterations", "setting_minOptIterations", "setting_thOptIterations", "setting_outlierTH", "setting_outlierTHSumComponent", "setting_margWeightFac", "setting_GNItsOnPointActivation", "setting_minTraceQuality", "setting_minTraceTestRadius", "setting_reTrackThreshold", "setting_affineOptModeA", "setting_affineOptModeB", "setting_forceAceptStep", "setting_huberTH", "setting_frameEnergyTHConstWeight", "setting_frameEnergyTHN", "setting_frameEnergyTHFacMedian", "setting_overallEnergyTHWeight", "setting_coarseCutoffTH", "setting_trace_stepsize", "setting_trace_GNIterations", "setting_trace_GNThreshold", "setting_trace_extraSlackOnTH", "setting_trace_slackInterval", "setting_trace_minImprovementFactor", "multiThreading"};
    return (i >= 0 && i < 46) ? names[i] : "";
}

}  // extern "C"
