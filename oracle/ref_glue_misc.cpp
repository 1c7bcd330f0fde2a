// oracle/ref_glue_misc.cpp -- TEST INFRASTRUCTURE.  C entry points around the REFERENCE'S OWN Reprojector (Reprojector.cpp: reprojectPoint,
// findMatchDirect -> getWarpMatrixAffine / getBestSearchLevel / warpAffine / align1D / align2D) and ImmaturePoint::traceOn (ImmaturePoint.cpp:50-353),
// compiled unmodified from /root/reference into oracle/_ref/libref.so (oracle/Makefile, target `ref`).  Signatures follow orc_reproject.cpp /
// orc_trace.cpp.  The glue builds the reference's objects and calls its (private) members; no arithmetic of the
// path is restated here except where a comment says which single expression of the reference is evaluated for reporting.
#include "ref_common.hpp"
#include "FullSystem/Reprojector.h"

using namespace refglue;

namespace {
struct RefRp {
    Globals g;
    CalibHessian* Hcalib = nullptr;
    std::vector<FrameHessian*> frames;      // key-frames 0..n-1
    std::vector<FrameShell*> shells;
    FrameHessian* cur = nullptr;
    FrameShell curShell;
    Reprojector* rp = nullptr;
    std::vector<FrameHessian*> all;         // frames + cur: what Reprojector keeps a reference to
    void on() const { install(g); }
};
FrameHessian* rp_bare(FrameShell* sh) {
    FrameHessian* fh = new FrameHessian();
    fh->shell = sh; fh->dI = 0; fh->ab_exposure = 1;
    for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    return fh;
}
void rp_set_level(FrameHessian* fh, int lvl, const float* aos3) {
    const size_t n = (size_t)wG[lvl] * hG[lvl];
    if (!fh->dIp[lvl]) { fh->dIp[lvl] = new Eigen::Vector3f[n]; fh->absSquaredGrad[lvl] = new float[n]; }
    std::memcpy((void*)fh->dIp[lvl], aos3, sizeof(float) * 3 * n);
    if (lvl == 0) fh->dI = fh->dIp[0];
}
void rp_free(FrameHessian* fh) {
    for (int l = pyrLevelsUsed; l < PYR_LEVELS; ++l) { if (fh->dIp[l]) delete[] fh->dIp[l]; if (fh->absSquaredGrad[l]) delete[] fh->absSquaredGrad[l]; }
    for (int l = 0; l < pyrLevelsUsed; ++l) if (!fh->dIp[l]) { fh->dIp[l] = new Eigen::Vector3f[1]; fh->absSquaredGrad[l] = new float[1]; }
    delete fh;
}
void rp_ready(RefRp* R) {
    if (R->rp) return;
    R->all = R->frames;
    R->all.push_back(R->cur);
    R->rp = new Reprojector(R->Hcalib, R->cur, R->all);      // Reprojector.cpp:81-87
}
PointHessian* rp_point(RefRp* R, float u, float v, float idepth, int host, int type) {
    ImmaturePoint ip(2, 2, R->cur, 1, R->Hcalib);
    ip.idepth_min = ip.idepth_max = idepth;
    ip.type = type ? ImmaturePoint::EDGELET : ImmaturePoint::CORNER;
    PointHessian* ph = new PointHessian(&ip, R->Hcalib);
    ph->u = u; ph->v = v;
    ph->setIdepth(idepth);
    ph->host = R->frames[host];
    ph->setPointStatus(PointHessian::ACTIVE);
    return ph;
}
}  // namespace

extern "C" {

void* ref_rp_create(int w0, int h0, int levels) {
    RefRp* R = new RefRp();
    R->g = Globals{w0, h0, levels, 1.f, 1.f, 0.f, 0.f};
    R->on();
    R->Hcalib = new CalibHessian();
    R->cur = rp_bare(&R->curShell);
    return R;
}
void ref_rp_destroy(void* h) {
    RefRp* R = (RefRp*)h; R->on();
    delete R->rp;
    for (FrameHessian* f : R->frames) rp_free(f);
    rp_free(R->cur);
    for (FrameShell* s : R->shells) delete s;
    delete R->Hcalib;
    delete R;
}
void ref_rp_set_calib(void* h, float fx, float fy, float cx, float cy) {
    RefRp* R = (RefRp*)h;
    R->g.fx = fx; R->g.fy = fy; R->g.cx = cx; R->g.cy = cy;
    R->on();
    VecC vs; vs << fx, fy, cx, cy;
    R->Hcalib->setValueScaled(vs);
}
void ref_rp_set_frame(void* h, int idx, const double* camToWorld7, const float* dI_aos3, float exposure, double a, double b) {
    RefRp* R = (RefRp*)h; R->on();
    assert(!R->rp);
    while ((int)R->frames.size() <= idx) { FrameShell* sh = new FrameShell(); sh->id = (int)R->frames.size(); R->shells.push_back(sh); R->frames.push_back(rp_bare(sh)); }
    FrameHessian* f = R->frames[idx];
    f->shell->camToWorld = pose_from7(camToWorld7);
    f->shell->aff_g2l = AffLight(a, b);
    f->ab_exposure = exposure;
    rp_set_level(f, 0, dI_aos3);
}
void ref_rp_set_cur_pose(void* h, const double* camToWorld7, float exposure, double a, double b) {
    RefRp* R = (RefRp*)h; R->on();
    R->cur->shell->camToWorld = pose_from7(camToWorld7);
    R->cur->shell->aff_g2l = AffLight(a, b);
    R->cur->shell->id = 1000;
    R->cur->ab_exposure = exposure;
}
void ref_rp_set_cur_level(void* h, int lvl, const float* dIp_aos3) { RefRp* R = (RefRp*)h; R->on(); rp_set_level(R->cur, lvl, dIp_aos3); }

// Reprojector::reprojectPoint (:602-616) per point: the projected pixel, the grid cell it is filed under (-1: outside the 8 px border), and
// the quantity pointQualityComparator (:186-194) orders a cell by -- `host->dI[(int)(v * wG[0] + u)].tail<2>().norm()`, evaluated here
void ref_rp_project(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, double* px2, int* cell, float* quality) {
    RefRp* R = (RefRp*)h; R->on();
    rp_ready(R);
    for (int i = 0; i < n; ++i) {
        PointHessian* ph = rp_point(R, u[i], v[i], idepth[i], host_idx[i], 0);
        Eigen::Vector3d ptWorld = R->rp->pixelFrame2PointWorld(ph);
        Eigen::Vector3d pixelCur = R->rp->pointWorld2PixelFrame(R->cur, ptWorld);
        Eigen::Vector2d px(pixelCur(0, 0), pixelCur(1, 0));
        px2[2 * i] = px[0]; px2[2 * i + 1] = px[1];
        if (R->rp->isInFrame(px.cast<int>(), 8))
            cell[i] = static_cast<int>(px[1] / R->rp->grid_.cell_size) * R->rp->grid_.grid_n_cols + static_cast<int>(px[0] / R->rp->grid_.cell_size);
        else cell[i] = -1;
        Vec2f g = ((ph->host->dI)[(int)(ph->v * wG[0] + ph->u)]).tail<2>();
        quality[i] = g.norm();
        delete ph;
    }
}
// Reprojector::findMatchDirect (:235-292) per candidate (windows of more than 2 frames: the reference patch comes from the point's host).
// level[i]: the search level, recomputed with the reference's own getWarpMatrixAffine / getBestSearchLevel (findMatchDirect keeps it local).
void ref_rp_find_match(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const int* ref_idx,
                       const int* type, double* px2_io, int* success, int* level) {
    RefRp* R = (RefRp*)h; R->on();
    rp_ready(R);
    assert(R->all.size() > 2);
    for (int i = 0; i < n; ++i) {
        assert(ref_idx[i] == host_idx[i]);
        PointHessian* ph = rp_point(R, u[i], v[i], idepth[i], host_idx[i], type[i]);
        Eigen::Vector2d px(px2_io[2 * i], px2_io[2 * i + 1]);
        success[i] = R->rp->findMatchDirect(ph, R->cur, px) ? 1 : 0;
        px2_io[2 * i] = px[0]; px2_io[2 * i + 1] = px[1];
        if (level) {
            FrameHessian* ref = ph->host;
            Eigen::Vector3d ptWorld = R->rp->pixelFrame2PointWorld(ph);
            Eigen::Vector3d ptRef = R->rp->pointWorld2PointFrame(ref, ptWorld);
            Eigen::Vector3d pixelRef = R->rp->pointWorld2PixelFrame(ref, ptWorld);
            Eigen::Vector2d pxr(pixelRef(0, 0), pixelRef(1, 0));
            if (!R->rp->isInFrame(pxr.cast<int>(), 4 + 2)) level[i] = -1;
            else {
                Eigen::Matrix2d A;
                R->rp->getWarpMatrixAffine(pxr, ptRef, R->cur->shell->camToWorld.inverse() * ref->shell->camToWorld, A);
                level[i] = R->rp->getBestSearchLevel(A, pyrLevelsUsed - 1);
            }
        }
        delete ph;
    }
}

// ImmaturePoint::traceOn (ImmaturePoint.cpp:50-353) for n points on one target image; state arrays updated in place (see orc_trace_on)
void ref_trace_on(int n, const float* u, const float* v, const float* energyTH, const float* gradH4, const float* color8, const float* weights8,
                  const int* host_idx, const float* KRKi9, const float* Kt3, const float* aff2, const float* dI_aos3, int w, int h,
                  float* idepth_min, float* idepth_max, float* quality, int* status, float* lastTraceUV2, float* lastTracePixelInterval) {
    Globals g{w, h, 1, 1.f, 1.f, 0.f, 0.f};
    install(g);
    CalibHessian* Hcalib = new CalibHessian();
    FrameShell sh;
    FrameHessian* frame = rp_bare(&sh);
    rp_set_level(frame, 0, dI_aos3);
    for (int i = 0; i < n; ++i) {
        ImmaturePoint ip(8, 8, frame, 1, Hcalib);          // (the constructor reads 8 pattern pixels of its host; every field it derives is overwritten)
        ip.u = u[i]; ip.v = v[i]; ip.host = frame;
        ip.energyTH = energyTH[i];
        ip.gradH(0, 0) = gradH4[4 * i]; ip.gradH(0, 1) = gradH4[4 * i + 1]; ip.gradH(1, 0) = gradH4[4 * i + 2]; ip.gradH(1, 1) = gradH4[4 * i + 3];
        for (int k = 0; k < 8; ++k) { ip.color[k] = color8[8 * i + k]; ip.weights[k] = weights8[8 * i + k]; }
        ip.idepth_min = idepth_min[i]; ip.idepth_max = idepth_max[i]; ip.quality = quality[i];
        ip.lastTraceStatus = (ImmaturePointStatus)status[i];
        ip.lastTraceUV = Vec2f(lastTraceUV2[2 * i], lastTraceUV2[2 * i + 1]);
        ip.lastTracePixelInterval = lastTracePixelInterval[i];
        const int hh = host_idx[i];
        Mat33f KRKi; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) KRKi(r, c) = KRKi9[9 * hh + 3 * r + c];
        const Vec3f Kt(Kt3[3 * hh], Kt3[3 * hh + 1], Kt3[3 * hh + 2]);
        const Vec2f aff(aff2[2 * hh], aff2[2 * hh + 1]);
        ip.traceOn(frame, KRKi, Kt, aff, Hcalib, false);
        idepth_min[i] = ip.idepth_min; idepth_max[i] = ip.idepth_max; quality[i] = ip.quality; status[i] = (int)ip.lastTraceStatus;
        lastTraceUV2[2 * i] = ip.lastTraceUV[0]; lastTraceUV2[2 * i + 1] = ip.lastTraceUV[1];
        lastTracePixelInterval[i] = ip.lastTracePixelInterval;
    }
    rp_free(frame);
    delete Hcalib;
}

}  // extern "C"
