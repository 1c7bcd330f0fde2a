// oracle/dropin/FullSystemFrameGPU.cpp -- TEST INFRASTRUCTURE and the reference-side bindings of INTEGRATION.md sections 2b-2d (form B+), as a file that
// COMPILES: the per-FRAME rows of SURVEY.md section 8f bound at the reference's own call sites.
//
// This translation unit DEFINES five member functions of the reference against its unmodified headers, each on top of libsdvgn's C ABI; oracle/Makefile
// (target `dropin_frame`) weakens exactly these symbols in the reference's objects, so that every caller inside the reference -- FullSystem::trackNewCoarse
// (FullSystem.cpp:483-489), FullSystem::makeNonKeyFrame / makeKeyFrame (:1017, :1047: traceNewCoarse; :1106: activatePointsMT; :1144: setCoarseTrackingRef) --
// reaches the GPU without a changed line:
//   void FullSystem::traceNewCoarse(FrameHessian*)                       FullSystem.cpp:519-553     -> ONE sdvgn_tracker_trace_points launch for all immature points
//   void FullSystem::activatePointsMT_Reductor(...)                      FullSystem.cpp:555-566     -> ONE sdvgn_ef_optimize_immature launch for all points to activate
//   void CoarseTracker::makeCoarseDepthL0(std::vector<FrameHessian*>)    CoarseTracker.cpp:258-425  -> sdvgn_tracker_make_coarse_depth (the template stays on the device)
//   bool CoarseTracker::structPoseEstimation(SE3&, overlap_pts)          CoarseTracker.cpp:946-1007 -> sdvgn_tracker_struct_pose
//   void Reprojector::reprojectMap(FrameHessian*, overlap_pts)           Reprojector.cpp:117-156    -> sdvgn_reproj_match for every candidate, then the reference's grid walk
// Everything the reference does AROUND these -- which immature points to trace / activate / delete, the distance map, PointHessian / PointFrameResidual
// construction, insertPoint / insertResidual, the grid-cell order (random_shuffle), the motion model -- stays the reference's code.
// There is no CPU fallback: a failing sdvgn_* call aborts like the reference's live asserts do.
#include "dropin_shared.hpp"
#include "FullSystem/ImmaturePoint.h"
#include "FullSystem/Reprojector.h"
#include "FullSystem/Residuals.h"
#include "util/globalCalib.h"
#include "util/settings.h"

#include <algorithm>
#include <map>
#include <mutex>

using namespace sdvgn_dropin;

namespace {

[[noreturn]] void fr_die(const char* what, int rc) { sdvgn_dropin::die("FullSystemFrameGPU", what, rc); }
#define GPU_CK(call) do { const int _rc = (call); if (_rc < 0) fr_die(#call, _rc); } while (0)

void pose7(const SE3& T, double* o) {     // Sophus data(): [qx qy qz qw], then the translation
    const double* q = T.so3().data();
    for (int k = 0; k < 4; ++k) o[k] = q[k];
    for (int k = 0; k < 3; ++k) o[4 + k] = T.translation()[k];
}

std::mutex g_mu;
// what sdvgn_tracker_trace_set_points last registered on a tracker handle: the immature points (by object) with their host indices
struct TraceSet { std::vector<const ImmaturePoint*> pts; std::vector<int> host; };
std::map<const sdvgn_tracker*, TraceSet> g_trace_sets;
struct FrameStats { unsigned long long trace_calls = 0, trace_points = 0, trace_registrations = 0, activate_calls = 0, activate_points = 0, template_calls = 0,
                    struct_pose_calls = 0, reproject_calls = 0, reproject_candidates = 0; };
FrameStats g_stats;

// the Reprojector is a per-frame stack object in the reference (FullSystem.cpp:484); its GPU side lives as long as the calibration object it is built from
struct GpuReproj {
    sdvgn_reproj* h = nullptr;
    int w = 0, hgt = 0;
    const FrameHessian* slot_frame[16];
    int slot_uid[16];
    GpuReproj() { for (int k = 0; k < 16; ++k) { slot_frame[k] = nullptr; slot_uid[k] = -1; } }
};
std::map<const CalibHessian*, GpuReproj> g_reproj;

}  // namespace

// test / bench hooks (C linkage): [0] traceNewCoarse calls, [1] points traced, [2] static registrations, [3] activation launches, [4] points through them,
// [5] templates built on the device, [6] structPoseEstimation calls, [7] reprojectMap calls, [8] candidates matched
extern "C" void sdvgn_dropin_frame_stats(double out9[9]) {
    std::lock_guard<std::mutex> lk(g_mu);
    out9[0] = (double)g_stats.trace_calls; out9[1] = (double)g_stats.trace_points; out9[2] = (double)g_stats.trace_registrations;
    out9[3] = (double)g_stats.activate_calls; out9[4] = (double)g_stats.activate_points; out9[5] = (double)g_stats.template_calls;
    out9[6] = (double)g_stats.struct_pose_calls; out9[7] = (double)g_stats.reproject_calls; out9[8] = (double)g_stats.reproject_candidates;
}
extern "C" void sdvgn_dropin_frame_release(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_reproj) if (kv.second.h) sdvgn_reproj_destroy(kv.second.h);
    g_reproj.clear();
    g_trace_sets.clear();
    g_stats = FrameStats();
}

namespace sdv_loam {

// ---------------------------------------------------------------------------------------------------------------------------------------------
// FullSystem.cpp:519-553.  The per-host quantities are the reference's own expressions (:525-538); the search itself -- ImmaturePoint::traceOn for
// every immature point of every key-frame -- is one launch on the tracker handle that holds fh's pyramid (the one trackNewestCoarse has just used).
void FullSystem::traceNewCoarse(FrameHessian* fh) {
    boost::unique_lock<boost::mutex> lock(mapMutex);

    Mat33f K = Mat33f::Identity();
    K(0, 0) = Hcalib.fxl();
    K(1, 1) = Hcalib.fyl();
    K(0, 2) = Hcalib.cxl();
    K(1, 2) = Hcalib.cyl();

    const int nH = (int)frameHessians.size();
    std::vector<float> KRKi9(9 * (size_t)nH), Kt3(3 * (size_t)nH), aff2(2 * (size_t)nH);
    size_t n = 0;
    for (int h = 0; h < nH; ++h) {
        FrameHessian* host = frameHessians[h];
        SE3 hostToNew = fh->PRE_worldToCam * host->PRE_camToWorld;
        Mat33f KRKi = K * hostToNew.rotationMatrix().cast<float>() * K.inverse();
        Vec3f Kt = K * hostToNew.translation().cast<float>();
        Vec2f aff = AffLight::fromToVecExposure(host->ab_exposure, fh->ab_exposure, host->aff_g2l(), fh->aff_g2l()).cast<float>();
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) KRKi9[9 * (size_t)h + 3 * r + c] = KRKi(r, c); Kt3[3 * (size_t)h + r] = Kt[r]; }
        aff2[2 * (size_t)h] = aff[0]; aff2[2 * (size_t)h + 1] = aff[1];
        n += host->immaturePoints.size();
    }
    if (n == 0) return;

    CoarseTracker* ct = coarseTracker;
    sdvgn_tracker* t = tracker_handle(ct);
    tracker_set_new_frame(ct, fh);                    // (already there when trackNewCoarse ran on this frame)
    GPU_CK(sdvgn_tracker_make_K(t, Hcalib.fxl(), Hcalib.fyl(), Hcalib.cxl(), Hcalib.cyl()));

    // the static part (u, v, energyTH, gradH, color, weights, host index: ImmaturePoint.h:33-75) is registered when the set of immature points changed:
    // at key-frames (makeNewTraces, activatePointsMT), not from frame to frame
    std::vector<ImmaturePoint*> pts; pts.reserve(n);
    std::vector<int> host_idx; host_idx.reserve(n);
    for (int h = 0; h < nH; ++h) for (ImmaturePoint* ph : frameHessians[h]->immaturePoints) { pts.push_back(ph); host_idx.push_back(h); }
    bool same;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        TraceSet& S = g_trace_sets[t];
        same = S.pts.size() == n && S.host == host_idx && std::equal(pts.begin(), pts.end(), S.pts.begin());
        if (!same) { S.pts.assign(pts.begin(), pts.end()); S.host = host_idx; ++g_stats.trace_registrations; }
        ++g_stats.trace_calls; g_stats.trace_points += n;
    }
    if (!same) {
        std::vector<float> u(n), v(n), eth(n), gH(4 * n), col(8 * n), wt(8 * n);
        for (size_t i = 0; i < n; ++i) {
            const ImmaturePoint* ph = pts[i];
            u[i] = ph->u; v[i] = ph->v; eth[i] = ph->energyTH;
            gH[4 * i] = ph->gradH(0, 0); gH[4 * i + 1] = ph->gradH(0, 1); gH[4 * i + 2] = ph->gradH(1, 0); gH[4 * i + 3] = ph->gradH(1, 1);
            for (int k = 0; k < 8; ++k) { col[8 * i + k] = ph->color[k]; wt[8 * i + k] = ph->weights[k]; }
        }
        GPU_CK(sdvgn_tracker_trace_set_points(t, (int)n, u.data(), v.data(), eth.data(), gH.data(), col.data(), wt.data(), host_idx.data()));
    }
    std::vector<float> imin(n), imax(n), q(n), uv(2 * n), iv(n);
    std::vector<int> st(n);
    for (size_t i = 0; i < n; ++i) {
        const ImmaturePoint* ph = pts[i];
        imin[i] = ph->idepth_min; imax[i] = ph->idepth_max; q[i] = ph->quality; st[i] = (int)ph->lastTraceStatus;
        uv[2 * i] = ph->lastTraceUV[0]; uv[2 * i + 1] = ph->lastTraceUV[1]; iv[i] = ph->lastTracePixelInterval;
    }
    GPU_CK(sdvgn_tracker_trace_points(t, nH, KRKi9.data(), Kt3.data(), aff2.data(), imin.data(), imax.data(), q.data(), st.data(), uv.data(), iv.data()));
    for (size_t i = 0; i < n; ++i) {
        ImmaturePoint* ph = pts[i];
        ph->idepth_min = imin[i]; ph->idepth_max = imax[i]; ph->quality = q[i]; ph->lastTraceStatus = (ImmaturePointStatus)st[i];
        ph->lastTraceUV = Vec2f(uv[2 * i], uv[2 * i + 1]); ph->lastTracePixelInterval = iv[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// FullSystem.cpp:555-566: optimizeImmaturePoint (FullSystemOptPoint.cpp:18-185) for (*toOptimize)[min .. max) -- its Gauss-Newton part for all of them in ONE
// launch on the resident window (which must hold the key-frame that has just been inserted, its image and the frames' current states: the window is
// synchronised first), its object construction (:133-181) here, statement for statement.
void FullSystem::activatePointsMT_Reductor(std::vector<PointHessian*>* optimized, std::vector<ImmaturePoint*>* toOptimize, int min, int max, Vec10* stats, int tid) {
    const int n = max - min;
    if (n <= 0) return;
    const std::vector<EFPoint*> points = all_points(ef);
    GpuWindow& g = window_for(this, (int)points.size() + n);
    std::vector<int> pid;
    sync_window(g, ef, points, Hcalib, pid);
    const int nF = (int)frameHessians.size();
    std::vector<int> host(n);
    std::vector<float> u(n), v(n), imin(n), imax(n), eth(n), col(8 * (size_t)n), wt(8 * (size_t)n);
    std::vector<unsigned char> sensor(n);
    for (int k = 0; k < n; ++k) {
        const ImmaturePoint* ph = (*toOptimize)[min + k];
        host[k] = ph->host->idx;
        u[k] = ph->u; v[k] = ph->v; imin[k] = ph->idepth_min; imax[k] = ph->idepth_max; eth[k] = ph->energyTH;
        for (int i = 0; i < 8; ++i) { col[8 * (size_t)k + i] = ph->color[i]; wt[8 * (size_t)k + i] = ph->weights[i]; }
        sensor[k] = ph->isFromSensor ? 1 : 0;
    }
    std::vector<int> result(n), rs((size_t)n * nF);
    std::vector<float> idepth(n);
    GPU_CK(sdvgn_ef_optimize_immature(g.h, n, host.data(), u.data(), v.data(), imin.data(), imax.data(), eth.data(), col.data(), wt.data(), sensor.data(),
                                      /*minObs*/ 1, result.data(), idepth.data(), rs.data()));
    ++g.immature_calls; g.immature_points += (unsigned long long)n;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stats.activate_calls; g_stats.activate_points += (unsigned long long)n;
    }
    for (int k = 0; k < n; ++k) {
        ImmaturePoint* point = (*toOptimize)[min + k];
        if (result[k] == 0) { (*optimized)[min + k] = 0; continue; }                                                  // :57-63, :83-90
        if (result[k] < 0) { (*optimized)[min + k] = (PointHessian*)((long)(-1)); continue; }                        // :113-131
        PointHessian* p = new PointHessian(point, &Hcalib);                                                            // :133-181
        if (!std::isfinite(p->energyTH)) { delete p; (*optimized)[min + k] = (PointHessian*)((long)(-1)); continue; }
        p->isFromSensor = point->isFromSensor;
        p->lastResiduals[0].first = 0;
        p->lastResiduals[0].second = ResState::OOB;
        p->lastResiduals[1].first = 0;
        p->lastResiduals[1].second = ResState::OOB;
        p->setIdepthZero(idepth[k]);                 // (trueDepth for a point from the sensor, currentIdepth otherwise: what the entry point returns)
        p->setIdepth(idepth[k]);
        p->setPointStatus(PointHessian::ACTIVE);
        for (int t = 0; t < nF; ++t) {
            if (frameHessians[t] == point->host || rs[(size_t)k * nF + t] != (int)ResState::IN) continue;
            PointFrameResidual* r = new PointFrameResidual(p, p->host, frameHessians[t]);
            r->state_NewEnergy = r->state_energy = 0;
            r->state_NewState = ResState::OUTLIER;
            r->setState(ResState::IN);
            p->residuals.push_back(r);
            if (r->target == frameHessians.back()) {
                p->lastResiduals[0].first = r;
                p->lastResiduals[0].second = ResState::IN;
            } else if (r->target == (frameHessians.size() < 2 ? 0 : frameHessians[frameHessians.size() - 2])) {
                p->lastResiduals[1].first = r;
                p->lastResiduals[1].second = ResState::IN;
            }
        }
        statistics_numActivatedPoints++;
        (*optimized)[min + k] = p;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// CoarseTracker.cpp:258-425 (called by setCoarseTrackingRef, :649-660, which has set lastRef): the tuples the reference splats into level 0, in its order;
// the splat, the pyramid of weighted sums, the two dilations, the normalisation and the compaction into pc_u / pc_v / pc_idepth / pc_color run on the device,
// and the template STAYS there for trackNewestCoarse.  The private host arrays are filled from the device copy (readers: debugPlot; the tests' template getter).
void CoarseTracker::makeCoarseDepthL0(std::vector<FrameHessian*> frameHessians) {
    std::vector<int> pu, pv;
    std::vector<float> pid, pw;
    for (FrameHessian* fh : frameHessians) {
        for (PointHessian* ph : fh->pointHessians) {
            if (fh == frameHessians.back() && ph->isFromSensor == true) {
                pu.push_back((int)ph->u); pv.push_back((int)ph->v);
                pid.push_back(ph->idepth);
                pw.push_back(sqrtf(1e-3 / (ph->efPoint->HdiF + 1e-12)));
            } else if (ph->lastResiduals[0].first != 0 && ph->lastResiduals[0].second == ResState::IN) {
                if (fh == frameHessians.back() || ph->isFromSensor == false) continue;
                PointFrameResidual* r = ph->lastResiduals[0].first;
                assert(r->efResidual->isActive() && r->target == lastRef);
                pu.push_back((int)(r->centerProjectedTo[0] + 0.5f)); pv.push_back((int)(r->centerProjectedTo[1] + 0.5f));
                pid.push_back(r->centerProjectedTo[2]);
                pw.push_back(sqrtf(1e-3 / (ph->efPoint->HdiF + 1e-12)));
            }
        }
    }
    sdvgn_tracker* t = tracker_handle(this);
    GPU_CK(sdvgn_tracker_make_K(t, fx[0], fy[0], cx[0], cy[0]));
    // lastRef's pyramid: still on the device when the key-frame was tracked as a frame (the other CoarseTracker's handle), uploaded otherwise
    const float* pyr[PYR_LEVELS];
    const float* const* pyr_arg = nullptr;
    if (sdvgn_tracker* holder = tracker_holding(lastRef)) {
        for (int l = 0; l < pyrLevelsUsed; ++l) pyr[l] = sdvgn_tracker_pyr_dev(holder, l);
        pyr_arg = pyr;
    } else tracker_set_new_frame(this, lastRef);
    GPU_CK(sdvgn_tracker_make_coarse_depth(t, (int)pu.size(), pu.data(), pv.data(), pid.data(), pw.data(), pyr_arg));
    for (int l = 0; l < pyrLevelsUsed; ++l) {
        const int nl = sdvgn_tracker_get_ref(t, l, pc_u[l], pc_v[l], pc_idepth[l], pc_color[l]);
        if (nl < 0) fr_die("sdvgn_tracker_get_ref", nl);
        pc_n[l] = nl;
    }
    tracker_mark_ref_on_device(this);
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_stats.template_calls;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// CoarseTracker.cpp:946-1007 (with calculateRes :840-872, calculateWeight :874-889, calcHandb :891-947): the Levenberg-Marquardt refinement of the new
// frame's pose on the matches of Reprojector::reprojectMap, all ten iterations in one launch.  (The reference's function is declared bool and has no
// return statement; its only caller ignores the value, FullSystem.cpp:488.)
bool CoarseTracker::structPoseEstimation(SE3& curToWorld, std::vector<std::pair<PointHessian*, Eigen::Vector2d>>& overlap_pts) {
    const int n = (int)overlap_pts.size();
    sdvgn_tracker* t = tracker_handle(this);
    GPU_CK(sdvgn_tracker_make_K(t, fx[0], fy[0], cx[0], cy[0]));
    std::vector<float> u(n), v(n), id(n);
    std::vector<int> hidx(n);
    std::vector<double> obs(2 * (size_t)n), hpose;
    std::vector<const FrameHessian*> hosts;
    for (int i = 0; i < n; ++i) {
        const PointHessian* ph = overlap_pts[i].first;
        u[i] = ph->u; v[i] = ph->v; id[i] = ph->idepth;
        obs[2 * (size_t)i] = overlap_pts[i].second[0]; obs[2 * (size_t)i + 1] = overlap_pts[i].second[1];
        int k = 0;
        while (k < (int)hosts.size() && hosts[k] != ph->host) ++k;
        if (k == (int)hosts.size()) { hosts.push_back(ph->host); hpose.resize(7 * hosts.size()); pose7(ph->host->shell->camToWorld, &hpose[7 * (size_t)k]); }
        hidx[i] = k;
    }
    double cw[7];
    pose7(curToWorld, cw);
    const int its = sdvgn_tracker_struct_pose(t, n, u.data(), v.data(), id.data(), hidx.data(), (int)hosts.size(), hpose.data(), obs.data(), cw, nullptr, nullptr);
    if (its < 0) fr_die("sdvgn_tracker_struct_pose", its);
    for (int k = 0; k < 4; ++k) curToWorld.so3().data()[k] = cw[k];
    for (int k = 0; k < 3; ++k) curToWorld.translation()[k] = cw[4 + k];
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_stats.struct_pose_calls;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Reprojector.cpp:117-156: candidates in the reference's push order (key-frames by distance to the new frame, :123-131; points in pointHessians order),
// reprojectPoint + findMatchDirect for ALL of them in one launch, then the reference's walk over the grid: cells in cell_order (random_shuffle of the
// constructor, :103), per cell the candidates sorted by pointQualityComparator (std::list::sort: stable), the first ACTIVE one that matched wins (:196-234).
void Reprojector::reprojectMap(FrameHessian* frame, std::vector<std::pair<PointHessian*, Eigen::Vector2d>>& overlap_pts) {
    resetGrid();
    GpuReproj* gp;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        gp = &g_reproj[Hcalib_];
    }
    GpuReproj& g = *gp;
    sdvgn_tracker* holder = tracker_holding(frame);          // the tracker that has just tracked this frame: its pyramid is on the device
    if (g.h && (g.w != wG[0] || g.hgt != hG[0])) { sdvgn_reproj_destroy(g.h); g = GpuReproj(); }
    if (!g.h) {
        g.w = wG[0]; g.hgt = hG[0];
        GPU_CK(sdvgn_reproj_create(&g.h, /*device*/ 0, wG[0], hG[0], pyrLevelsUsed, 16, 1 << 16, nullptr));
    }
    GPU_CK(sdvgn_reproj_set_calib(g.h, Hcalib_->fxl(), Hcalib_->fyl(), Hcalib_->cxl(), Hcalib_->cyl()));
    // key-frames keep their slot (image uploaded once per key-frame); poses and brightness parameters are re-sent every frame
    int slot_of[16];
    const int nK = (int)frameHessians_.size();
    if (nK > 16) fr_die("more than 16 key-frames", -1);
    for (int s = 0; s < 16; ++s) {
        bool alive = false;
        for (FrameHessian* kf : frameHessians_) if (g.slot_frame[s] == kf && g.slot_uid[s] == kf->shell->id) alive = true;
        if (!alive) { g.slot_frame[s] = nullptr; g.slot_uid[s] = -1; }
    }
    for (int k = 0; k < nK; ++k) {
        FrameHessian* kf = frameHessians_[k];
        int s = 0;
        while (s < 16 && !(g.slot_frame[s] == kf && g.slot_uid[s] == kf->shell->id)) ++s;
        double cw[7];
        pose7(kf->shell->camToWorld, cw);
        if (s == 16) {
            s = 0;
            while (s < 16 && g.slot_frame[s]) ++s;
            g.slot_frame[s] = kf; g.slot_uid[s] = kf->shell->id;
            GPU_CK(sdvgn_reproj_set_frame(g.h, s, cw, (const float*)kf->dI, nullptr, kf->ab_exposure, kf->shell->aff_g2l.a, kf->shell->aff_g2l.b));
        } else GPU_CK(sdvgn_reproj_set_frame(g.h, s, cw, nullptr, nullptr, kf->ab_exposure, kf->shell->aff_g2l.a, kf->shell->aff_g2l.b));
        slot_of[k] = s;
    }
    double cw[7];
    pose7(frame->shell->camToWorld, cw);
    GPU_CK(sdvgn_reproj_set_cur(g.h, cw, frame->ab_exposure, frame->shell->aff_g2l.a, frame->shell->aff_g2l.b));
    for (int l = 0; l < pyrLevelsUsed; ++l)
        GPU_CK(sdvgn_reproj_set_cur_level(g.h, l, holder ? nullptr : (const float*)frame->dIp[l], holder ? sdvgn_tracker_pyr_dev(holder, l) : nullptr));

    // :123-131: key-frames newest first, then sorted by distance (std::list::sort: stable)
    std::vector<std::pair<int, double>> close_kfs;
    for (int i = nK - 1; i >= 0; i--)
        close_kfs.push_back(std::make_pair(i, (newframe_->shell->camToWorld.translation() - frameHessians_[i]->shell->camToWorld.translation()).norm()));
    std::stable_sort(close_kfs.begin(), close_kfs.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.second < b.second; });
    std::vector<PointHessian*> pts;
    std::vector<float> u, v, id;
    std::vector<int> host, ref, type;
    for (const auto& kd : close_kfs) {
        FrameHessian* ref_frame = frameHessians_[kd.first];
        if (ref_frame == frame) continue;
        for (PointHessian* ph : ref_frame->pointHessians) {
            pts.push_back(ph); u.push_back(ph->u); v.push_back(ph->v); id.push_back(ph->idepth);
            host.push_back(slot_of[kd.first]);
            int rk = kd.first;                                   // findMatchDirect's reference frame (:240-251): the host, or frame 0 of a window of <= 2
            if (frameHessians_.size() <= 2) rk = 0;
            else { rk = 0; while (rk < nK && frameHessians_[rk] != ph->host) ++rk; if (rk == nK) fr_die("a point's host is not in the window", -1); }
            ref.push_back(slot_of[rk]);
            type.push_back(ph->type == PointHessian::EDGELET ? 1 : 0);
        }
    }
    const int n = (int)pts.size();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_stats.reproject_calls; g_stats.reproject_candidates += (unsigned long long)n;
    }
    if (n == 0) return;
    std::vector<double> px0(2 * (size_t)n), px(2 * (size_t)n);
    std::vector<int> cell(n), ok(n);
    std::vector<float> q(n);
    GPU_CK(sdvgn_reproj_match(g.h, n, u.data(), v.data(), id.data(), host.data(), ref.data(), type.data(), px0.data(), cell.data(), q.data(), ok.data(), px.data(), nullptr));
    std::vector<std::vector<int>> cells(grid_.cells.size());
    for (int i = 0; i < n; ++i) if (cell[i] >= 0) cells[cell[i]].push_back(i);
    for (size_t c = 0; c < grid_.cells.size(); ++c) {
        std::vector<int>& L = cells[grid_.cell_order[c]];
        std::stable_sort(L.begin(), L.end(), [&](int a, int b) { return q[a] < q[b]; });
        for (int i : L) {
            if (pts[i]->status != PointHessian::ACTIVE || !ok[i]) continue;
            overlap_pts.push_back(std::pair<PointHessian*, Eigen::Vector2d>(pts[i], Eigen::Vector2d(px[2 * (size_t)i], px[2 * (size_t)i + 1])));
            ++n_matches_;
            break;
        }
        if (n_matches_ > (int)(0.8 * setting_desiredImmatureDensity)) break;
    }
}

}  // namespace sdv_loam
