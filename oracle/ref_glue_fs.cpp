// oracle/ref_glue_fs.cpp -- TEST INFRASTRUCTURE.  C entry points that run the REFERENCE'S OWN FullSystem::trackNewCoarse (FullSystem.cpp:283-517) --
// the function whose line 419 is the coarse tracker's call site: the motion-model tries, the retry / early-out logic around
// CoarseTracker::trackNewestCoarse, then Reprojector::reprojectMap and CoarseTracker::structPoseEstimation -- on a small world built from flat
// arrays: key-frames (pose, level-0 {I,dx,dy} image, active points), the tracking template of the newest key-frame, and the new frame's
// image.  Compiled into oracle/_ref/libref.so (all CPU) and into oracle/_ref/libref_dropin.so, where the trackNewestCoarse the function
// calls is the GPU-backed definition of oracle/dropin/CoarseTrackerGPU.cpp: tests/test_dropin_gpu.py runs both and compares what the
// reference's own host code leaves behind.  The glue only allocates objects and fills input fields; no arithmetic of the path is restated.
#include "ref_common.hpp"
#include "FullSystem/Reprojector.h"

#include <cstdlib>

using namespace refglue;

namespace {
struct RefFS {
    Globals g;
    FullSystem* fs = nullptr;
    std::vector<FrameHessian*> kfs;
    std::vector<PointHessian*> pts;
    FrameHessian* cur = nullptr;
    std::string last_log;
    void on() const { install(g); }
};
FrameHessian* fs_bare(FrameShell* sh) {
    FrameHessian* fh = new FrameHessian();
    fh->shell = sh; fh->dI = 0; fh->ab_exposure = 1;
    for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    return fh;
}
void fs_free_frame(FrameHessian* fh) {    // ~FrameHessian deletes dIp[0 .. pyrLevelsUsed) itself
    for (int l = pyrLevelsUsed; l < PYR_LEVELS; ++l) { if (fh->dIp[l]) delete[] fh->dIp[l]; if (fh->absSquaredGrad[l]) delete[] fh->absSquaredGrad[l]; }
    for (int l = 0; l < pyrLevelsUsed; ++l) if (!fh->dIp[l]) { fh->dIp[l] = new Eigen::Vector3f[1]; fh->absSquaredGrad[l] = new float[1]; }
    fh->pointHessians.clear();
    delete fh;
}
}  // namespace

extern "C" {

void* ref_fs_create(int w0, int h0, int levels, float fx, float fy, float cx, float cy) {
    RefFS* F = new RefFS();
    F->g = Globals{w0, h0, levels, fx, fy, cx, cy};
    F->on();
    F->fs = new FullSystem();                 // FullSystem.cpp:38-41 -> :119-232: builds coarseTracker, coarseTracker_forNewKF, ef ...
    F->fs->selectionMapFromLidar = 0;         // (left uninitialised by the constructor and delete[]d by the destructor, see ref_glue_ef.cpp)
    VecC vs; vs << fx, fy, cx, cy;
    F->fs->Hcalib.setValueScaled(vs);
    return F;
}

void ref_fs_destroy(void* h) {
    RefFS* F = (RefFS*)h; F->on();
    for (PointHessian* p : F->pts) delete p;
    for (FrameHessian* f : F->kfs) fs_free_frame(f);
    if (F->cur) fs_free_frame(F->cur);
    std::string sink = capture_stdout([&] { delete F->fs; });   // ~FullSystem deletes the shells of allFrameHistory and both trackers
    delete F;
}

// a key-frame, oldest first: its shell goes to allFrameHistory (FullSystem.cpp:831-835), the frame to frameHessians
void ref_fs_add_keyframe(void* h, const double* camToWorld7, const float* dI_aos3, float exposure, double a, double b) {
    RefFS* F = (RefFS*)h; F->on();
    FullSystem* fs = F->fs;
    FrameShell* sh = new FrameShell();
    sh->id = sh->incoming_id = (int)fs->allFrameHistory.size();
    sh->camToWorld = pose_from7(camToWorld7);
    sh->aff_g2l = AffLight(a, b);
    sh->poseValid = true;
    fs->allFrameHistory.push_back(sh);
    FrameHessian* fh = fs_bare(sh);
    fh->ab_exposure = exposure;
    fh->idx = (int)fs->frameHessians.size();
    fh->frameID = fh->idx;
    const size_t n = (size_t)wG[0] * hG[0];
    fh->dIp[0] = new Eigen::Vector3f[n]; fh->absSquaredGrad[0] = new float[n];
    std::memcpy((void*)fh->dIp[0], dI_aos3, sizeof(float) * 3 * n);
    fh->dI = fh->dIp[0];
    fs->frameHessians.push_back(fh);
    F->kfs.push_back(fh);
}

// active points of the key-frames (what Reprojector::reprojectMap projects into the new frame, Reprojector.cpp:117-156)
void ref_fs_add_points(void* h, int n, const int* host, const float* u, const float* v, const float* idepth, const int* type) {
    RefFS* F = (RefFS*)h; F->on();
    for (int i = 0; i < n; ++i) {
        FrameHessian* fh = F->kfs[host[i]];
        ImmaturePoint ip(2, 2, fh, 1, &F->fs->Hcalib);
        ip.idepth_min = ip.idepth_max = idepth[i];
        ip.type = type[i] ? ImmaturePoint::EDGELET : ImmaturePoint::CORNER;
        PointHessian* ph = new PointHessian(&ip, &F->fs->Hcalib);
        ph->u = u[i]; ph->v = v[i];
        ph->setIdepth(idepth[i]);
        ph->host = fh;
        ph->setPointStatus(PointHessian::ACTIVE);
        ph->efPoint = 0;
        fh->pointHessians.push_back(ph);
        F->pts.push_back(ph);
    }
}

// the tracking template of the newest key-frame: CoarseTracker::makeK (CoarseTracker.cpp:77-106), then pc_* as setCoarseTrackingRef leaves
// them (:649-660; the template itself is row a3, built by makeCoarseDepthL0 from the window's points -- here handed over ready-made)
void ref_fs_set_tracker_ref(void* h, int lvl, int n, const float* pc_u, const float* pc_v, const float* pc_idepth, const float* pc_color) {
    RefFS* F = (RefFS*)h; F->on();
    CoarseTracker* ct = F->fs->coarseTracker;
    if (lvl == 0) {
        ct->makeK(&F->fs->Hcalib);
        ct->lastRef = F->kfs.back();
        ct->refFrameID = ct->lastRef->shell->id;
        ct->lastRef_aff_g2l = ct->lastRef->shell->aff_g2l;
        ct->firstCoarseRMSE = -1;
        ct->debugPlot = ct->debugPrint = false;
    }
    ct->pc_n[lvl] = n;
    std::memcpy(ct->pc_u[lvl], pc_u, sizeof(float) * n); std::memcpy(ct->pc_v[lvl], pc_v, sizeof(float) * n);
    std::memcpy(ct->pc_idepth[lvl], pc_idepth, sizeof(float) * n); std::memcpy(ct->pc_color[lvl], pc_color, sizeof(float) * n);
}

// the new frame: a fresh FrameHessian + shell (appended to allFrameHistory like FullSystem::addActiveFrame does, :826-835), its pyramid by
// the reference's FrameHessian::makeImages (HessianBlocks.cpp:107-167); rows 0 / h-1 of the gradient planes, which makeImages leaves
// uninitialised, are zeroed (the template never reaches them)
void ref_fs_set_new_frame(void* h, const float* color_lvl0, float exposure) {
    RefFS* F = (RefFS*)h; F->on();
    FullSystem* fs = F->fs;
    FrameShell* sh = new FrameShell();
    sh->id = sh->incoming_id = (int)fs->allFrameHistory.size();
    sh->poseValid = true;
    fs->allFrameHistory.push_back(sh);
    if (F->cur) fs_free_frame(F->cur);
    F->cur = fs_bare(sh);
    F->cur->ab_exposure = exposure;
    std::vector<float> c(color_lvl0, color_lvl0 + (size_t)F->g.w * F->g.h);
    F->cur->makeImages(c.data(), &fs->Hcalib);
    for (int l = 0; l < pyrLevelsUsed; ++l) {
        const int wl = wG[l], hl = hG[l];
        for (int x = 0; x < wl; ++x) for (int k = 1; k < 3; ++k) { F->cur->dIp[l][x][k] = 0; F->cur->dIp[l][(size_t)wl * (hl - 1) + x][k] = 0; }
    }
}

// Vec4 FullSystem::trackNewCoarse(FrameHessian* fh)   FullSystem.cpp:283-517.  out4 = its return value {achievedRes[0], flow x3}; the pose
// and brightness it leaves in the new frame's shell; lastCoarseRMSE.  The Reprojector shuffles its grid cells with rand(): seeded here.
void ref_fs_track_new_coarse(void* h, double out4[4], double camToWorld7[7], double camToTrackingRef7[7], double aff2[2], double lastCoarseRMSE5[5]) {
    RefFS* F = (RefFS*)h; F->on();
    FullSystem* fs = F->fs;
    srand(1);
    Vec4 r;
    F->last_log = capture_stdout([&] { r = fs->trackNewCoarse(F->cur); });
    for (int i = 0; i < 4; ++i) out4[i] = r[i];
    pose_to7(F->cur->shell->camToWorld, camToWorld7);
    pose_to7(F->cur->shell->camToTrackingRef, camToTrackingRef7);
    aff2[0] = F->cur->shell->aff_g2l.a; aff2[1] = F->cur->shell->aff_g2l.b;
    for (int i = 0; i < 5; ++i) lastCoarseRMSE5[i] = fs->lastCoarseRMSE[i];
}
int ref_fs_last_log(void* h, char* buf, int cap) {
    RefFS* F = (RefFS*)h;
    const int n = (int)F->last_log.size();
    if (buf && cap > 0) { const int m = n < cap - 1 ? n : cap - 1; std::memcpy(buf, F->last_log.data(), (size_t)m); buf[m] = 0; }
    return n;
}
void* ref_fs_coarse_tracker(void* h) { return ((RefFS*)h)->fs->coarseTracker; }
void ref_fs_set_last_coarse_rmse(void* h, double v) { ((RefFS*)h)->fs->lastCoarseRMSE.setConstant(v); }

}  // extern "C"
