// oracle/dropin/CoarseTrackerGPU.cpp -- TEST INFRASTRUCTURE and the reference-side binding of INTEGRATION.md section 1, as a file that COMPILES.
//
// DEFINES the member function
//     bool sdv_loam::CoarseTracker::trackNewestCoarse(FrameHessian*, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl,
//                                                     Vec5 minResForAbort, IOWrap::Output3DWrapper*)          (CoarseTracker.cpp:662-838)
// against the reference's unmodified header (src/FullSystem/CoarseTracker.h:24-29) with libsdvgn's C ABI behind it.  oracle/Makefile (target
// `dropin`) links it INSTEAD OF the reference's definition (that one symbol is weakened in CoarseTracker.o); FullSystem::trackNewCoarse's
// call (FullSystem.cpp:419) and every other caller then reach the GPU without a changed line.  The template (pc_u / pc_v / pc_idepth /
// pc_color, built by the reference's own makeCoarseDepthL0 on the host, row a3 of SURVEY.md section 8 "keep on host"), the intrinsics
// makeK left in the object and the new frame's pyramid dIp[] are handed over as they are; the member leaves behind exactly what the
// reference's leaves behind: the refined pose and affine parameters, lastResiduals, lastFlowIndicators, newFrame.
#include "dropin_shared.hpp"
#include "util/globalCalib.h"
#include "util/settings.h"

#include <map>
#include <mutex>

namespace {

struct GpuTracker {
    sdvgn_tracker* h = nullptr;
    int w = 0, hgt = 0, levels = 0;
    const sdv_loam::FrameHessian* ref = nullptr;      // the template that is on the device
    int ref_id = -1, ref_n0 = -1;
    const sdv_loam::FrameHessian* cur = nullptr;      // the new frame whose pyramid is on the device ...
    int cur_id = -1;                                  // ... and its FrameShell::id (the address of a deleted frame may be handed out again)
    unsigned long long calls = 0;
};
std::mutex g_mu;
std::map<const sdv_loam::CoarseTracker*, GpuTracker> g_handles;   // (a member `sdvgn_tracker* gpu` in a real integration)

[[noreturn]] void ct_die(const char* what, int rc) { sdvgn_dropin::die("CoarseTrackerGPU", what, rc); }
#define GPU_CK(call) do { const int _rc = (call); if (_rc < 0) ct_die(#call, _rc); } while (0)

// the handle of a CoarseTracker, created on first use
GpuTracker& gpu_of(sdv_loam::CoarseTracker* ct) {
    GpuTracker* gp;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        gp = &g_handles[ct];
    }
    GpuTracker& g = *gp;
    // (the geometry of the global calibration, util/globalCalib.h: what CoarseTracker::makeK copies into w[] / h[] -- which a tracker that has not been
    // given its first template yet has not run)
    if (g.h && (g.w != sdv_loam::wG[0] || g.hgt != sdv_loam::hG[0] || g.levels != sdv_loam::pyrLevelsUsed)) { sdvgn_tracker_destroy(g.h); g.h = nullptr; }
    if (!g.h) {
        g.w = sdv_loam::wG[0]; g.hgt = sdv_loam::hG[0]; g.levels = sdv_loam::pyrLevelsUsed;
        GPU_CK(sdvgn_tracker_create(&g.h, /*device*/ 0, g.w, g.hgt, g.levels, /*max_points*/ g.w * g.hgt, /*max_batch*/ 32, nullptr));
        g.ref = nullptr; g.cur = nullptr;
    }
    return g;
}
void set_new_frame(GpuTracker& g, sdv_loam::FrameHessian* fh) {
    if (g.cur == fh && g.cur_id == fh->shell->id) return;             // once per frame, not once per pose hypothesis
    for (int l = 0; l < sdv_loam::pyrLevelsUsed; ++l)
        GPU_CK(sdvgn_tracker_set_new_pyr(g.h, l, reinterpret_cast<const float*>(fh->dIp[l]), fh->ab_exposure));
    g.cur = fh; g.cur_id = fh->shell->id;
}

}  // namespace

namespace sdvgn_dropin {
sdvgn_tracker* tracker_handle(CoarseTracker* ct) { return gpu_of(ct).h; }
void tracker_set_new_frame(CoarseTracker* ct, FrameHessian* fh) { set_new_frame(gpu_of(ct), fh); }
sdvgn_tracker* tracker_holding(const FrameHessian* fh) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_handles) if (kv.second.h && kv.second.cur == fh && kv.second.cur_id == fh->shell->id) return kv.second.h;
    return nullptr;
}
void tracker_mark_ref_on_device(CoarseTracker* ct) {
    GpuTracker& g = gpu_of(ct);
    g.ref = ct->lastRef; g.ref_id = ct->refFrameID; g.ref_n0 = -2;      // (-2: "whatever pc_n[0] says": see trackNewestCoarse)
}
}  // namespace sdvgn_dropin

extern "C" unsigned long long sdvgn_dropin_tracker_calls(const void* ct) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find((const sdv_loam::CoarseTracker*)ct);
    return it == g_handles.end() ? 0 : it->second.calls;
}
extern "C" void sdvgn_dropin_tracker_release(const void* ct) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find((const sdv_loam::CoarseTracker*)ct);
    if (it == g_handles.end()) return;
    if (it->second.h) sdvgn_tracker_destroy(it->second.h);
    g_handles.erase(it);
}
// a template rebuilt in place (same lastRef, same point count) -- the glue of the tests sets pc_* directly -- must be uploaded again
extern "C" void sdvgn_dropin_tracker_invalidate(const void* ct) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_handles.find((const sdv_loam::CoarseTracker*)ct);
    if (it != g_handles.end()) { it->second.ref = nullptr; it->second.cur = nullptr; }
}

namespace sdv_loam {

bool CoarseTracker::trackNewestCoarse(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl,
                                      Vec5 minResForAbort, IOWrap::Output3DWrapper* wrap) {
    debugPlot = setting_render_displayCoarseTrackingFull;          // CoarseTracker.cpp:668-676
    debugPrint = false;
    assert(coarsestLvl < 5 && coarsestLvl < pyrLevelsUsed);
    lastResiduals.setConstant(NAN);
    lastFlowIndicators.setConstant(1000);
    newFrame = newFrameHessian;

    GpuTracker& g = gpu_of(this);
    ++g.calls;
    GPU_CK(sdvgn_tracker_set_settings(g.h, setting_huberTH, setting_coarseCutoffTH, setting_affineOptModeA, setting_affineOptModeB));
    GPU_CK(sdvgn_tracker_make_K(g.h, fx[0], fy[0], cx[0], cy[0]));     // what makeK(HCalib) derived the pyramid of intrinsics from (:77-106)
    if (g.ref != lastRef || g.ref_id != refFrameID || (g.ref_n0 != pc_n[0] && g.ref_n0 != -2)) {   // once per key-frame: the template of setCoarseTrackingRef (:649-660)
        for (int l = 0; l < pyrLevelsUsed; ++l) GPU_CK(sdvgn_tracker_set_ref(g.h, l, pc_n[l], pc_u[l], pc_v[l], pc_idepth[l], pc_color[l]));
        g.ref = lastRef; g.ref_id = refFrameID; g.ref_n0 = pc_n[0];
    }
    GPU_CK(sdvgn_tracker_set_ref_frame(g.h, lastRef->ab_exposure, lastRef_aff_g2l.a, lastRef_aff_g2l.b));
    set_new_frame(g, newFrameHessian);
    double aff[2] = {aff_g2l_out.a, aff_g2l_out.b};
    double minRes[5], lastRes[5], flow[3];
    for (int i = 0; i < 5; ++i) minRes[i] = minResForAbort[i];
    // pose7 = Sophus SE3d::data() layout [qx qy qz qw | tx ty tz]; the quaternion goes in and comes back as stored (no re-normalisation)
    double pose7[7];
    for (int k = 0; k < 4; ++k) pose7[k] = lastToNew_out.so3().data()[k];
    for (int k = 0; k < 3; ++k) pose7[4 + k] = lastToNew_out.translation()[k];
    const int ok = sdvgn_tracker_track(g.h, pose7, aff, coarsestLvl, minRes, lastRes, flow);
    if (ok < 0) ct_die("sdvgn_tracker_track", ok);
    for (int k = 0; k < 4; ++k) lastToNew_out.so3().data()[k] = pose7[k];
    for (int k = 0; k < 3; ++k) lastToNew_out.translation()[k] = pose7[4 + k];
    aff_g2l_out.a = aff[0]; aff_g2l_out.b = aff[1];
    for (int i = 0; i < 5; ++i) lastResiduals[i] = lastRes[i];          // read by FullSystem.cpp:436-458
    for (int i = 0; i < 3; ++i) lastFlowIndicators[i] = flow[i];        // read by FullSystem.cpp:494
    return ok == 1;
}

}  // namespace sdv_loam
