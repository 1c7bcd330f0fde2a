// oracle/dropin/EnergyFunctionalGPU.cpp -- TEST INFRASTRUCTURE and the reference-side binding of INTEGRATION.md section 2, as a file that COMPILES.
//
// This translation unit DEFINES the member function
//     void sdv_loam::EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian* HCalib)     (EnergyFunctional.cpp:650-759)
// against the reference's own, unmodified headers (src/OptimizationBackend/EnergyFunctional.h:51) -- same class, same signature, same
// mangled symbol -- with libsdvgn's C ABI (include/sdvgn.h) behind it.  oracle/Makefile (target `dropin`) links it INSTEAD OF the reference's
// definition: the reference's object files are used as compiled, only the one symbol is weakened in EnergyFunctional.o so that this strong
// definition wins.  Every caller inside the reference -- FullSystem::solveSystem (FullSystemOptimize.cpp:504-513), hence
// FullSystem::optimize (:344-502) -- then reaches the GPU without a changed line: that is the drop-in north_star asks for.  A maintainer
// does the same thing in the source tree: replace the body of solveSystemF by this one (and keep the handle as a member instead of the
// side table below).
//
// What crosses the boundary per call (the reference keeps PointFrameResidual::linearize and the accept / reject loop on the host):
//   in : calibration, frame states / linearisation points, points, residual flags and -- EFResidual::takeDataF's product -- the Jacobians
//        the EnergyFunctional side owns (efResidual->J), res_toZeroF, HM / bM, the null-space vectors FullSystem::solveSystem just computed
//   out: lastX, HCalib->step, FrameHessian::step, PointHessian::step, EFPoint::{HdiF, bdSumF, Hdd_accAF, bd_accAF, Hcd_accAF}, resInA,
//        lastHS / lastbS -- everything the reference's solveSystemF + resubstituteF_MT leave behind for their callers.
// There is no CPU fallback: a failing sdvgn_* call aborts like the reference's live asserts do.
#include "OptimizationBackend/EnergyFunctional.h"
#include "OptimizationBackend/EnergyFunctionalStructs.h"
#include "FullSystem/HessianBlocks.h"
#include "FullSystem/Residuals.h"
#include "util/globalCalib.h"

extern "C" {
#include "sdvgn.h"
}

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct GpuEF {
    sdvgn_ef* h = nullptr;
    int max_points = 0, w = 0, hgt = 0;
    unsigned long long calls = 0;
};
std::mutex g_mu;
std::map<const sdv_loam::EnergyFunctional*, GpuEF> g_handles;   // (a member `sdvgn_ef* gpu` in a real integration)

void die(const char* what, int rc) {
    fprintf(stderr, "EnergyFunctionalGPU: %s failed: %s (%d)\n", what, sdvgn_error_string(rc), rc);
    abort();
}
#define GPU_CK(call) do { const int _rc = (call); if (_rc < 0) die(#call, _rc); } while (0)

GpuEF& handle_for(const sdv_loam::EnergyFunctional* ef, int nP) {
    std::lock_guard<std::mutex> lk(g_mu);
    GpuEF& g = g_handles[ef];
    const int w = sdv_loam::wG[0], hgt = sdv_loam::hG[0];
    if (g.h && (nP > g.max_points || w != g.w || hgt != g.hgt)) { sdvgn_ef_destroy(g.h); g.h = nullptr; }
    if (!g.h) {
        g.max_points = nP > 4096 ? nP + nP / 4 : 4096;
        g.w = w; g.hgt = hgt;
        GPU_CK(sdvgn_ef_create(&g.h, /*device*/ 0, w, hgt, g.max_points, /*stream*/ nullptr));
    }
    return g;
}

}  // namespace

// test hooks (C linkage): how many solves went through the GPU; release a window's handle before its EnergyFunctional is deleted
extern "C" unsigned long long sdvgn_dropin_ef_calls(const void* ef) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find((const sdv_loam::EnergyFunctional*)ef);
    return it == g_handles.end() ? 0 : it->second.calls;
}
extern "C" void sdvgn_dropin_ef_release(const void* ef) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find((const sdv_loam::EnergyFunctional*)ef);
    if (it == g_handles.end()) return;
    if (it->second.h) sdvgn_ef_destroy(it->second.h);
    g_handles.erase(it);
}

namespace sdv_loam {

void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian* HCalib) {
    if (setting_solverMode & SOLVER_USE_GN) lambda = 0;            // EnergyFunctional.cpp:652-653
    if (setting_solverMode & SOLVER_FIX_LAMBDA) lambda = 1e-5;
    assert(EFDeltaValid);
    assert(EFAdjointsValid);
    assert(EFIndicesValid);
    const int nF = nFrames, n = CPARS + 6 * nF;

    // ---- flatten the window: frames in EF order, points grouped by host (allPoints order, makeIDX :761-782), residual rows ----
    std::vector<double> evalPT(7 * nF), state(10 * nF), state_zero(10 * nF);
    std::vector<int> frameID(nF);
    std::vector<float> ab_exposure(nF), frameTH(nF);
    for (int i = 0; i < nF; ++i) {
        const FrameHessian* fh = frames[i]->data;
        const double* q = fh->worldToCam_evalPT.so3().data();          // Sophus data(): [qx qy qz qw], then the translation
        for (int k = 0; k < 4; ++k) evalPT[7 * i + k] = q[k];
        for (int k = 0; k < 3; ++k) evalPT[7 * i + 4 + k] = fh->worldToCam_evalPT.translation()[k];
        for (int k = 0; k < 10; ++k) { state[10 * i + k] = fh->state[k]; state_zero[10 * i + k] = fh->state_zero[k]; }
        frameID[i] = frames[i]->frameID; ab_exposure[i] = fh->ab_exposure; frameTH[i] = fh->frameEnergyTH;
    }
    const int nP = (int)allPoints.size();
    std::vector<int> host(nP);
    std::vector<float> u(nP), v(nP), idepth(nP), idepth_zero(nP), color(8 * (size_t)nP), weights(8 * (size_t)nP);
    std::vector<unsigned char> hasPrior(nP), fromSensor(nP);
    std::vector<int> r_point, r_target, r_state;
    std::vector<unsigned char> r_hasMatcher, r_lin, r_act;
    std::vector<double> r_matcher;
    std::vector<float> J24, r2z;
    for (int pi = 0; pi < nP; ++pi) {
        const EFPoint* p = allPoints[pi];
        const PointHessian* ph = p->data;
        host[pi] = p->host->idx;
        u[pi] = ph->u; v[pi] = ph->v; idepth[pi] = ph->idepth; idepth_zero[pi] = ph->idepth_zero;
        for (int k = 0; k < 8; ++k) { color[8 * (size_t)pi + k] = ph->color[k]; weights[8 * (size_t)pi + k] = ph->weights[k]; }
        hasPrior[pi] = ph->hasDepthPrior ? 1 : 0; fromSensor[pi] = ph->isFromSensor ? 1 : 0;
        for (const EFResidual* r : p->residualsAll) {
            r_point.push_back(pi); r_target.push_back(r->targetIDX); r_state.push_back((int)r->data->state_state);
            r_hasMatcher.push_back(r->data->hasMatcher ? 1 : 0);
            r_matcher.push_back(r->data->matcher[0]); r_matcher.push_back(r->data->matcher[1]);
            r_lin.push_back(r->isLinearized ? 1 : 0); r_act.push_back(r->isActive() ? 1 : 0);
            const RawResidualJacobian* J = r->J;                       // what takeDataF swapped in (EnergyFunctionalStructs.cpp:15-25)
            J24.push_back(J->resF[0]); J24.push_back(J->resF[1]);
            for (int k = 0; k < 6; ++k) J24.push_back(J->Jpdxi[0][k]);
            for (int k = 0; k < 6; ++k) J24.push_back(J->Jpdxi[1][k]);
            for (int k = 0; k < 4; ++k) J24.push_back(J->Jpdc[0][k]);
            for (int k = 0; k < 4; ++k) J24.push_back(J->Jpdc[1][k]);
            J24.push_back(J->Jpdd[0]); J24.push_back(J->Jpdd[1]);
            r2z.push_back(r->res_toZeroF[0]); r2z.push_back(r->res_toZeroF[1]);
        }
    }
    const int nR = (int)r_point.size();
    GpuEF& g = handle_for(this, nP);
    ++g.calls;

    // ---- hand the window over (the entry points mirror the members they stand for, include/sdvgn.h) ----
    double vs[4], vmz[4];
    for (int k = 0; k < 4; ++k) { vs[k] = HCalib->value_scaled[k]; vmz[k] = HCalib->value_minus_value_zero[k]; }
    GPU_CK(sdvgn_ef_set_calib(g.h, vs, vmz));
    GPU_CK(sdvgn_ef_set_frames(g.h, nF, evalPT.data(), state.data(), state_zero.data(), frameID.data(), ab_exposure.data(), frameTH.data()));
    GPU_CK(sdvgn_ef_set_points(g.h, nP, host.data(), u.data(), v.data(), idepth.data(), idepth_zero.data(), color.data(), weights.data(),
                               hasPrior.data(), fromSensor.data()));
    GPU_CK(sdvgn_ef_set_residuals(g.h, nR, r_point.data(), r_target.data(), r_state.data(), r_hasMatcher.data(), r_matcher.data(), r_lin.data(),
                                  r_act.data()));
    GPU_CK(sdvgn_ef_set_residual_jacobians(g.h, nR, J24.data(), r2z.data()));
    std::vector<double> HMr((size_t)n * n), bMr(n);
    for (int r = 0; r < n; ++r) { bMr[r] = bM[r]; for (int c = 0; c < n; ++c) HMr[(size_t)r * n + c] = HM(r, c); }
    GPU_CK(sdvgn_ef_set_marg_prior(g.h, HMr.data(), bMr.data()));
    {   // the vectors FullSystem::solveSystem put there right before this call (getNullspaces, FullSystemOptimize.cpp:506-510)
        const int k = (int)(lastNullspaces_pose.size() + lastNullspaces_scale.size());
        std::vector<double> ns((size_t)k * n);
        int j = 0;
        for (const VecX& vv : lastNullspaces_pose) { for (int i = 0; i < n; ++i) ns[(size_t)j * n + i] = vv[i]; ++j; }
        for (const VecX& vv : lastNullspaces_scale) { for (int i = 0; i < n; ++i) ns[(size_t)j * n + i] = vv[i]; ++j; }
        GPU_CK(sdvgn_ef_set_nullspaces(g.h, k, ns.data()));
    }
    GPU_CK(sdvgn_ef_set_adjoints(g.h));
    GPU_CK(sdvgn_ef_set_precalc(g.h));

    // ---- solveSystemF on the device: accumulate A / L / SC, stitch, HM / bM, damped preconditioned LDL^T, orthogonalize, resubstitute ----
    VecX x(n);
    GPU_CK(sdvgn_ef_solve_system(g.h, iteration, lambda, x.data()));

    // ---- what solveSystemF / resubstituteF_MT leave behind (EnergyFunctional.cpp:221-282, 744-758) ----
    lastX = x;
    {
        std::vector<double> HF((size_t)n * n), bF(n);
        GPU_CK(sdvgn_ef_get_system(g.h, nullptr, nullptr, nullptr, nullptr, HF.data(), bF.data()));
        lastHS = MatXX(n, n); lastbS = VecX(n);
        for (int r = 0; r < n; ++r) { lastbS[r] = bF[r]; for (int c = 0; c < n; ++c) lastHS(r, c) = HF[(size_t)r * n + c]; }
    }
    HCalib->step = -x.head<CPARS>();
    for (EFFrame* h : frames) {
        h->data->step.head<6>() = -x.segment<6>(CPARS + 6 * h->idx);
        h->data->step.tail<4>().setZero();
    }
    std::vector<float> pts(9 * (size_t)nP);
    std::vector<double> topacc((size_t)nF * nF * 121);
    GPU_CK(sdvgn_ef_get_points(g.h, pts.data()));
    GPU_CK(sdvgn_ef_get_top_acc(g.h, topacc.data(), &resInA));
    for (int pi = 0; pi < nP; ++pi) {
        EFPoint* p = allPoints[pi];
        const float* o = &pts[9 * (size_t)pi];
        p->Hdd_accAF = o[0]; p->bd_accAF = o[1];
        for (int k = 0; k < 4; ++k) p->Hcd_accAF[k] = o[2 + k];
        p->HdiF = o[6]; p->bdSumF = o[7];
        p->data->step = o[8];
        assert(std::isfinite(p->data->step));
    }
}

}  // namespace sdv_loam
