// oracle/ref_glue_tracker.cpp -- TEST INFRASTRUCTURE.  C entry points that drive the REFERENCE'S OWN coarse tracker -- CoarseTracker (makeK,
// calcRes, calcGSSSE, trackNewestCoarse, makeCoarseDepthL0 / ForFirstFrame, structPoseEstimation / calcHandb / calculateRes),
// FrameHessian::makeImages, and the vendored Sophus SE3 -- compiled unmodified from /root/reference into oracle/_ref/libref.so
// (oracle/Makefile, target `ref`).  Every ref_* function has the signature of the orc_* function of the same name in orc_tracker.cpp, so
// that oracle/__init__.py can run one problem through both and tests/test_ref_pin_tracker.py can compare them.
// The glue allocates the reference's objects, fills their input fields and calls the reference's member functions (private ones through
// -fno-access-control); no arithmetic of the path is restated here.
#include "ref_common.hpp"

using namespace refglue;

namespace {

struct RefTracker {
    Globals g;
    float huberTH = 6, coarseCutoffTH = 20, affA = 0, affB = 0;
    CalibHessian* Hcalib = nullptr;
    CoarseTracker* ct = nullptr;
    FrameHessian* ref = nullptr;      // lastRef
    FrameHessian* cur = nullptr;      // newFrame
    FrameShell refShell, curShell;
    void on() const {
        install(g);
        setting_huberTH = huberTH; setting_coarseCutoffTH = coarseCutoffTH;
        setting_affineOptModeA = affA; setting_affineOptModeB = affB;
    }
};

FrameHessian* bare_frame(FrameShell* sh) {
    FrameHessian* fh = new FrameHessian();
    fh->shell = sh; fh->dI = 0; fh->ab_exposure = 1;
    for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    return fh;
}
void free_images(FrameHessian* fh) {
    for (int l = 0; l < PYR_LEVELS; ++l) { if (fh->dIp[l]) delete[] fh->dIp[l]; if (fh->absSquaredGrad[l]) delete[] fh->absSquaredGrad[l]; fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    fh->dI = 0;
}
void alloc_images(FrameHessian* fh, int levels) {
    for (int l = 0; l < levels; ++l) {
        if (!fh->dIp[l]) { fh->dIp[l] = new Eigen::Vector3f[(size_t)wG[l] * hG[l]]; fh->absSquaredGrad[l] = new float[(size_t)wG[l] * hG[l]];
                           std::memset((void*)fh->dIp[l], 0, sizeof(Eigen::Vector3f) * (size_t)wG[l] * hG[l]); }
    }
    fh->dI = fh->dIp[0];
}
void destroy_frame(FrameHessian* fh) {
    // ~FrameHessian deletes dIp[0..pyrLevelsUsed) itself
    for (int l = pyrLevelsUsed; l < PYR_LEVELS; ++l) { if (fh->dIp[l]) delete[] fh->dIp[l]; if (fh->absSquaredGrad[l]) delete[] fh->absSquaredGrad[l]; }
    for (int l = 0; l < pyrLevelsUsed; ++l) if (!fh->dIp[l]) { fh->dIp[l] = new Eigen::Vector3f[1]; fh->absSquaredGrad[l] = new float[1]; }
    delete fh;
}

}  // namespace

extern "C" {

void* ref_tracker_create(int w0, int h0, int levels) {
    RefTracker* T = new RefTracker();
    T->g = Globals{w0, h0, levels, 1.f, 1.f, 0.f, 0.f};
    T->on();
    T->Hcalib = new CalibHessian();
    T->ct = new CoarseTracker(w0, h0);           // CoarseTracker.cpp:34-69
    T->ct->debugPlot = T->ct->debugPrint = false;   // the constructor switches the residual-image display on (:64); trackNewestCoarse resets it (:668-669)
    T->ref = bare_frame(&T->refShell);
    T->cur = bare_frame(&T->curShell);
    T->ct->lastRef = T->ref;
    T->ct->newFrame = T->cur;
    return T;
}
void ref_tracker_destroy(void* h) {
    RefTracker* T = (RefTracker*)h;
    T->on();
    delete T->ct;
    destroy_frame(T->ref); destroy_frame(T->cur);
    delete T->Hcalib;
    delete T;
}
void* ref_tracker_object(void* h) { return ((RefTracker*)h)->ct; }   // key of the drop-in's side table (oracle/dropin/CoarseTrackerGPU.cpp)
void ref_tracker_set_settings(void* h, float huberTH, float coarseCutoffTH, float affA, float affB) {
    RefTracker* T = (RefTracker*)h; T->huberTH = huberTH; T->coarseCutoffTH = coarseCutoffTH; T->affA = affA; T->affB = affB;
}
// CoarseTracker::makeK (CoarseTracker.cpp:77-106) from a CalibHessian holding the given level-0 intrinsics
void ref_tracker_make_K(void* h, float fx, float fy, float cx, float cy) {
    RefTracker* T = (RefTracker*)h;
    T->g.fx = fx; T->g.fy = fy; T->g.cx = cx; T->g.cy = cy;
    T->on();
    VecC vs; vs << fx, fy, cx, cy;
    T->Hcalib->setValueScaled(vs);
    T->ct->makeK(T->Hcalib);
}
void ref_tracker_get_K(void* h, int lvl, float out4[4], float Ki9[9]) {
    RefTracker* T = (RefTracker*)h;
    out4[0] = T->ct->fx[lvl]; out4[1] = T->ct->fy[lvl]; out4[2] = T->ct->cx[lvl]; out4[3] = T->ct->cy[lvl];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ki9[3 * r + c] = T->ct->Ki[lvl](r, c);
}
void ref_tracker_set_ref(void* h, int lvl, int n, const float* u, const float* v, const float* idepth, const float* color) {
    RefTracker* T = (RefTracker*)h;
    CoarseTracker* ct = T->ct;
    assert(n <= (T->g.w >> lvl) * (T->g.h >> lvl));
    ct->pc_n[lvl] = n;
    std::memcpy(ct->pc_u[lvl], u, sizeof(float) * n); std::memcpy(ct->pc_v[lvl], v, sizeof(float) * n);
    std::memcpy(ct->pc_idepth[lvl], idepth, sizeof(float) * n); std::memcpy(ct->pc_color[lvl], color, sizeof(float) * n);
}
int ref_tracker_get_ref(void* h, int lvl, float* u, float* v, float* idepth, float* color) {
    RefTracker* T = (RefTracker*)h; CoarseTracker* ct = T->ct;
    const int n = ct->pc_n[lvl];
    if (u) { std::memcpy(u, ct->pc_u[lvl], 4 * n); std::memcpy(v, ct->pc_v[lvl], 4 * n); std::memcpy(idepth, ct->pc_idepth[lvl], 4 * n); std::memcpy(color, ct->pc_color[lvl], 4 * n); }
    return n;
}
void ref_tracker_set_ref_frame(void* h, float exposure, double a, double b) {
    RefTracker* T = (RefTracker*)h;
    T->ref->ab_exposure = exposure;
    T->ct->lastRef_aff_g2l = AffLight(a, b);
}
// FrameHessian::makeImages (HessianBlocks.cpp:107-167) on the new frame.  Rows 0 and h_l-1 of the gradient planes, which makeImages leaves
// uninitialised, are set to NaN afterwards (the oracle does the same so that any use of them shows).
void ref_tracker_set_new_image(void* h, const float* color, float exposure) {
    RefTracker* T = (RefTracker*)h; T->on();
    free_images(T->cur);
    std::vector<float> c(color, color + (size_t)T->g.w * T->g.h);
    T->cur->makeImages(c.data(), T->Hcalib);
    for (int l = 0; l < pyrLevelsUsed; ++l) {
        const int wl = wG[l], hl = hG[l];
        for (int x = 0; x < wl; ++x) for (int k = 1; k < 3; ++k) { T->cur->dIp[l][x][k] = NAN; T->cur->dIp[l][(size_t)wl * (hl - 1) + x][k] = NAN; }
    }
    T->cur->ab_exposure = exposure;
}
void ref_tracker_set_new_pyr(void* h, int lvl, const float* dIp_aos3, float exposure) {
    RefTracker* T = (RefTracker*)h; T->on();
    alloc_images(T->cur, T->g.levels);
    std::memcpy((void*)T->cur->dIp[lvl], dIp_aos3, sizeof(float) * 3 * (size_t)wG[lvl] * hG[lvl]);
    T->cur->ab_exposure = exposure;
}
void ref_tracker_get_pyr(void* h, int lvl, float* out) {
    RefTracker* T = (RefTracker*)h; T->on();
    std::memcpy(out, (void*)T->cur->dIp[lvl], sizeof(float) * 3 * (size_t)wG[lvl] * hG[lvl]);
}
void ref_make_images(const float* color, int w0, int h0, int levels, float* out_concat) {
    void* h = ref_tracker_create(w0, h0, levels);
    ref_tracker_set_new_image(h, color, 1.f);
    size_t off = 0;
    for (int l = 0; l < levels; ++l) { ref_tracker_get_pyr(h, l, out_concat + off); off += (size_t)3 * (w0 >> l) * (h0 >> l); }
    ref_tracker_destroy(h);
}
void ref_calc_res(void* h, int lvl, const double pose7[7], double a, double b, float cutoffTH, double out6[6]) {
    RefTracker* T = (RefTracker*)h; T->on();
    T->ct->debugPlot = false;
    const Vec6 r = T->ct->calcRes(lvl, pose_from7(pose7), AffLight(a, b), cutoffTH);    // CoarseTracker.cpp:486-634
    for (int i = 0; i < 6; ++i) out6[i] = r[i];
}
int ref_get_warped(void* h, float* out8) {
    RefTracker* T = (RefTracker*)h; CoarseTracker* ct = T->ct;
    const int n = ct->buf_warped_n;
    const float* p[8] = {ct->buf_warped_idepth, ct->buf_warped_u, ct->buf_warped_v, ct->buf_warped_dx, ct->buf_warped_dy, ct->buf_warped_residual,
                         ct->buf_warped_weight, ct->buf_warped_refColor};
    if (out8) for (int k = 0; k < 8; ++k) std::memcpy(out8 + (size_t)k * n, p[k], sizeof(float) * n);
    return n;
}
void ref_calc_gs(void* h, int lvl, double a, double b, double H64[64], double b8[8]) {
    RefTracker* T = (RefTracker*)h; T->on();
    Mat88 H; Vec8 bb;
    T->ct->calcGSSSE(lvl, H, bb, SE3(), AffLight(a, b));                               // CoarseTracker.cpp:427-484
    for (int r = 0; r < 8; ++r) { b8[r] = bb[r]; for (int c = 0; c < 8; ++c) H64[8 * r + c] = H(r, c); }
}
int ref_trace_stride() { return 0; }
// CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:662-838); no per-iteration trace (the reference exposes none)
int ref_track(void* h, double pose7_io[7], double aff_io[2], int coarsestLvl, const double minRes[5], double lastRes[5], double flow[3],
              double* trace, int trace_cap, int* trace_n) {
    RefTracker* T = (RefTracker*)h; T->on();
    SE3 P = pose_from7(pose7_io);
    AffLight aff(aff_io[0], aff_io[1]);
    Vec5 mr; for (int i = 0; i < 5; ++i) mr[i] = minRes[i];
    bool ok = false;
    std::string sink = capture_stdout([&] { ok = T->ct->trackNewestCoarse(T->cur, P, aff, coarsestLvl, mr, 0); });
    pose_to7(P, pose7_io);
    aff_io[0] = aff.a; aff_io[1] = aff.b;
    for (int i = 0; i < 5; ++i) lastRes[i] = T->ct->lastResiduals[i];
    for (int i = 0; i < 3; ++i) flow[i] = T->ct->lastFlowIndicators[i];
    if (trace_n) *trace_n = 0;
    return ok ? 1 : 0;
}

// makeCoarseDepthL0 (CoarseTracker.cpp:258-425) / makeCoarseDepthForFirstFrame (:108-256) driven with real PointHessian objects on ONE frame
// (the branch both functions take for points of `frameHessians.back()` that come from the sensor, :268-277 / :114-125): per point the pixel
// (float u, v), its idepth and the HdiF of its EFPoint, from which the reference forms the splat weight sqrtf(1e-3 / (HdiF + 1e-12)).
// lastRef's pyramid = the pyramid currently set on the new frame (like the oracle's hook).
void ref_tracker_make_coarse_depth_pts(void* h, int n, const float* u, const float* v, const float* idepth, const float* HdiF, int first_frame) {
    RefTracker* T = (RefTracker*)h; T->on();
    FrameHessian* fh = T->cur;
    std::vector<EFPoint*> efps;
    for (int i = 0; i < n; ++i) {
        ImmaturePoint ip(2, 2, fh, 1, T->Hcalib);
        ip.idepth_min = ip.idepth_max = idepth[i];
        ip.type = ImmaturePoint::CORNER;
        PointHessian* ph = new PointHessian(&ip, T->Hcalib);
        ph->u = u[i]; ph->v = v[i];
        ph->setIdepth(idepth[i]);
        ph->isFromSensor = true;
        ph->lastResiduals[0] = std::pair<PointFrameResidual*, ResState>((PointFrameResidual*)0, ResState::OOB);
        ph->lastResiduals[1] = ph->lastResiduals[0];
        ph->host = fh;
        EFPoint* ep = (EFPoint*)::operator new(sizeof(EFPoint));   // only HdiF is read (:273 / :120); EFPoint's constructor wants an EFFrame
        std::memset((void*)ep, 0, sizeof(EFPoint));
        ep->HdiF = HdiF[i];
        ph->efPoint = ep;
        efps.push_back(ep);
        fh->pointHessians.push_back(ph);
    }
    FrameHessian* savedRef = T->ct->lastRef;
    T->ct->lastRef = fh;
    std::vector<FrameHessian*> fhs; fhs.push_back(fh);
    if (first_frame) T->ct->makeCoarseDepthForFirstFrame(fh);
    else T->ct->makeCoarseDepthL0(fhs);
    T->ct->lastRef = savedRef;
    for (PointHessian* ph : fh->pointHessians) { ph->efPoint = 0; delete ph; }
    fh->pointHessians.clear();
    for (EFPoint* ep : efps) ::operator delete((void*)ep);
}

// ---- structPoseEstimation (CoarseTracker.cpp:840-1007) ------------------------------------------------------------------------------------
static void struct_objects(RefTracker* T, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const double* host_pose7,
                           const double* obs, std::vector<FrameHessian*>& hosts, std::vector<FrameShell*>& shells,
                           std::vector<std::pair<PointHessian*, Eigen::Vector2d> >& pts) {
    int nh = 0;
    for (int i = 0; i < n; ++i) if (host_idx[i] + 1 > nh) nh = host_idx[i] + 1;
    for (int k = 0; k < nh; ++k) {
        FrameShell* sh = new FrameShell();
        sh->camToWorld = pose_from7(host_pose7 + 7 * k);
        shells.push_back(sh);
        hosts.push_back(bare_frame(sh));
    }
    for (int i = 0; i < n; ++i) {
        ImmaturePoint ip(2, 2, T->cur, 1, T->Hcalib);
        ip.idepth_min = ip.idepth_max = idepth[i];
        ip.type = ImmaturePoint::CORNER;
        PointHessian* ph = new PointHessian(&ip, T->Hcalib);
        ph->u = u[i]; ph->v = v[i];
        ph->setIdepth(idepth[i]);
        ph->host = hosts[host_idx[i]];
        pts.push_back(std::pair<PointHessian*, Eigen::Vector2d>(ph, Eigen::Vector2d(obs[2 * i], obs[2 * i + 1])));
    }
}
static void struct_free(std::vector<FrameHessian*>& hosts, std::vector<FrameShell*>& shells, std::vector<std::pair<PointHessian*, Eigen::Vector2d> >& pts) {
    for (auto& p : pts) delete p.first;
    for (FrameHessian* f : hosts) destroy_frame(f);
    for (FrameShell* s : shells) delete s;
}
int ref_struct_trace_stride() { return 0; }
// returns 0 (the reference keeps no iteration count); pose7_io = curToWorld, updated by accepted steps
int ref_struct_pose(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const double* host_pose7,
                    const double* obs, double pose7_io[7], double* trace, double* final_res) {
    RefTracker* T = (RefTracker*)h; T->on();
    alloc_images(T->cur, T->g.levels);
    std::vector<FrameHessian*> hosts; std::vector<FrameShell*> shells; std::vector<std::pair<PointHessian*, Eigen::Vector2d> > pts;
    struct_objects(T, n, u, v, idepth, host_idx, host_pose7, obs, hosts, shells, pts);
    SE3 P = pose_from7(pose7_io);
    T->ct->debugPrint = false;
    T->ct->structPoseEstimation(P, pts);
    pose_to7(P, pose7_io);
    if (final_res) *final_res = 0;
    struct_free(hosts, shells, pts);
    return 0;
}
void ref_struct_res_Hb(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const double* host_pose7,
                       const double* obs, const double worldToCur7[7], double H36[36], double b6[6], double* energy, int* num) {
    RefTracker* T = (RefTracker*)h; T->on();
    alloc_images(T->cur, T->g.levels);
    std::vector<FrameHessian*> hosts; std::vector<FrameShell*> shells; std::vector<std::pair<PointHessian*, Eigen::Vector2d> > pts;
    struct_objects(T, n, u, v, idepth, host_idx, host_pose7, obs, hosts, shells, pts);
    const SE3 P = pose_from7(worldToCur7);
    Mat66 H; H.setZero(); Vec6 b; b.setZero();
    T->ct->calcHandb(H, b, P, pts);
    int nn = 0;
    *energy = (double)T->ct->calculateRes(P, pts, nn);
    *num = nn;
    for (int r = 0; r < 6; ++r) { b6[r] = b[r]; for (int c = 0; c < 6; ++c) H36[6 * r + c] = H(r, c); }
    struct_free(hosts, shells, pts);
}

// ---- the vendored Sophus (thirdparty/Sophus/sophus/se3.hpp, so3.hpp) --------------------------------------------------------------------
void ref_se3_exp(const double a[6], double pose7[7]) { Vec6 t; for (int i = 0; i < 6; ++i) t[i] = a[i]; pose_to7(SE3::exp(t), pose7); }
void ref_se3_log(const double pose7[7], double a[6]) { const Vec6 t = pose_from7(pose7).log(); for (int i = 0; i < 6; ++i) a[i] = t[i]; }
void ref_se3_mul(const double A[7], const double B[7], double out[7]) { pose_to7(pose_from7(A) * pose_from7(B), out); }
void ref_se3_inverse(const double A[7], double out[7]) { pose_to7(pose_from7(A).inverse(), out); }
void ref_se3_matrix(const double A[7], double R9[9]) { const Mat33 R = pose_from7(A).rotationMatrix(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R9[3 * r + c] = R(r, c); }
void ref_se3_adj(const double A[7], double out36[36]) { const Mat66 M = pose_from7(A).Adj(); for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out36[6 * r + c] = M(r, c); }

}  // extern "C"
