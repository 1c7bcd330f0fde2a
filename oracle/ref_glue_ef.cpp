// oracle/ref_glue_ef.cpp -- TEST INFRASTRUCTURE.  C entry points that drive the REFERENCE'S OWN sliding-window back end -- FullSystem,
// EnergyFunctional, AccumulatedTopHessianSSE, AccumulatedSCHessianSSE, PointFrameResidual, FrameFramePrecalc ... compiled unmodified from
// /root/reference into oracle/_ref/libref.so (oracle/Makefile, target `ref`) -- on the same flattened window description the CPU oracle
// takes.  Every ref_ef_* function has the signature of the orc_ef_* function of the same name in orc_backend.cpp, so that
// oracle/backend.py can run one window through both and tests/test_ref_pin_backend.py can compare them.
//
// What the glue itself does is object plumbing only: it allocates the reference's FrameHessian / PointHessian / PointFrameResidual objects,
// fills their input fields from the arrays, registers them with the reference's EnergyFunctional through its own insertFrame / insertPoint /
// insertResidual, and calls the reference's member functions (private ones through -fno-access-control: the reference offers no public
// way to build a window without its ROS front end).  No arithmetic of the path is restated here.
#include "ref_common.hpp"

#include <execinfo.h>
#include <map>
#include <set>
#include <signal.h>
#include <time.h>
#include <unistd.h>

using namespace refglue;

namespace {

struct RefEF {
    Globals g;
    FullSystem* fs = nullptr;
    std::vector<FrameHessian*> fhs;
    std::vector<FrameShell*> shells;
    std::vector<PointHessian*> phs;
    std::vector<PointFrameResidual*> prs;      // input order; nullptr once the reference has deleted the residual
    std::vector<int> r_point;
    std::vector<uint8_t> removed_by_finish;     // input order: dropped by the last linearizeAll(true)
    bool custom_nullspaces = false;
    // copies made by ref_ef_solve_system (the reference keeps HA_top / H_sc as locals of solveSystemF)
    MatXX HA, Hsc, HFinal; VecX bA, bsc, bFinal;
    int color_mismatch = 0;                     // points whose given color/weights differ from what the ImmaturePoint constructor computed
    double image_grad_maxdiff = 0;              // max |given dx,dy - makeImages' dx,dy| over interior rows
    std::string last_log;
    void on() const { install(g); }
};

void fill_active(RefEF* E) {   // FullSystem::optimize :353-372 without the resetOOB
    FullSystem* fs = E->fs;
    fs->activeResiduals.clear();
    for (FrameHessian* fh : fs->frameHessians)
        for (PointHessian* ph : fh->pointHessians)
            for (PointFrameResidual* r : ph->residuals)
                if (!r->efResidual->isLinearized) fs->activeResiduals.push_back(r);
}

void refresh_live(RefEF* E) {   // which of the input residuals still exist
    std::set<PointFrameResidual*> live;
    for (PointHessian* ph : E->phs) if (ph) for (PointFrameResidual* r : ph->residuals) live.insert(r);
    for (size_t i = 0; i < E->prs.size(); ++i) if (E->prs[i] && !live.count(E->prs[i])) E->prs[i] = nullptr;
}

template <typename M> void to_rowmajor(const M& m, double* out) {
    if (!out) return;
    for (int r = 0; r < m.rows(); ++r) for (int c = 0; c < m.cols(); ++c) out[(size_t)r * m.cols() + c] = m(r, c);
}

}  // namespace

extern "C" {

void* ref_ef_create(int w, int h) {
    RefEF* E = new RefEF();
    E->g = Globals{w, h, 1, 1.f, 1.f, 0.f, 0.f};
    return E;
}

// pyramid levels of the world (default 1: the back end works on level 0 only).  CoarseTracker::makeCoarseDepthL0 touches levels 0 and 1
// whatever pyrLevelsUsed says (CoarseTracker.cpp:324-351), so a world that runs setCoarseTrackingRef (ref_ef_keyframe_tail) needs >= 2.
// Before ref_ef_set_calib (which constructs the FullSystem and its trackers).
void ref_ef_set_levels(void* e, int levels) { RefEF* E = (RefEF*)e; if (!E->fs) E->g.levels = levels; }

void ref_ef_destroy(void* e) {
    RefEF* E = (RefEF*)e;
    E->on();
    if (E->fs) {
        std::vector<FrameHessian*> alive = E->fs->frameHessians;    // (FullSystem::marginalizeFrame has deleted the frames that left the window)
        std::string sink = capture_stdout([&] { delete E->fs; });   // ~FullSystem deletes ef (which detaches every EF* back pointer)
        for (FrameHessian* fh : alive) delete fh;                      // ~FrameHessian releases its points and their residuals
        for (FrameShell* s : E->shells) delete s;
    }
    delete E;
}

// CalibHessian::setValueScaled (HessianBlocks.h:318-330) with the given scaled intrinsics; value_zero is placed so that
// value_minus_value_zero is the given vector
void ref_ef_set_calib(void* e, const double value_scaled[4], const double value_minus_value_zero[4]) {
    RefEF* E = (RefEF*)e;
    E->g.fx = (float)value_scaled[0]; E->g.fy = (float)value_scaled[1]; E->g.cx = (float)value_scaled[2]; E->g.cy = (float)value_scaled[3];
    E->on();
    if (!E->fs) {
        E->fs = new FullSystem();            // the reference's constructor (FullSystem.cpp:38-41 -> initializationValue :119-232)
        // the constructor leaves this pointer uninitialised and the destructor delete[]s it (FullSystem.cpp:62; only makeNewTraces,
        // :1289, ever assigns it): without this line destroying a FullSystem that never traced a lidar frame frees a wild pointer
        E->fs->selectionMapFromLidar = 0;
    }
    CalibHessian& C = E->fs->Hcalib;
    VecC vs; vs << value_scaled[0], value_scaled[1], value_scaled[2], value_scaled[3];
    C.setValueScaled(vs);
    VecC d; d << value_minus_value_zero[0], value_minus_value_zero[1], value_minus_value_zero[2], value_minus_value_zero[3];
    C.value_zero = C.value - d;
    C.value_minus_value_zero = d;
}

void ref_ef_set_frames(void* e, int nF, const double* evalPT7, const double* state10, const double* state_zero10, const int* frameID,
                       const float* ab_exposure, const float* frameEnergyTH) {
    RefEF* E = (RefEF*)e;
    E->on();
    FullSystem* fs = E->fs;
    assert(fs);
    // (called again on a loaded window it APPENDS key-frames, like makeKeyFrame does, FullSystem.cpp:1071-1075: index = position in the window)
    for (int i = 0; i < nF; ++i) {
        FrameShell* sh = new FrameShell();
        sh->id = (int)E->shells.size(); sh->incoming_id = sh->id;
        FrameHessian* fh = new FrameHessian();
        fh->shell = sh;
        fh->ab_exposure = ab_exposure[i];
        fh->frameID = frameID[i];
        fh->idx = (int)fs->frameHessians.size();
        fh->dI = 0;
        for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
        fh->worldToCam_evalPT = pose_from7(evalPT7 + 7 * i);
        Vec10 sz, st;
        for (int k = 0; k < 10; ++k) { sz[k] = state_zero10[10 * i + k]; st[k] = state10[10 * i + k]; }
        fh->setStateZero(sz);     // also the null-space columns (HessianBlocks.cpp:49-85)
        fh->setState(st);
        fh->frameEnergyTH = frameEnergyTH[i];
        E->fhs.push_back(fh); E->shells.push_back(sh);
        fs->frameHessians.push_back(fh);
        fs->ef->insertFrame(fh, &fs->Hcalib);
        sh->camToWorld = fh->PRE_camToWorld;
        sh->aff_g2l = fh->aff_g2l();
    }
}

void ref_ef_set_frame_state(void* e, int idx, const double* state10) {
    RefEF* E = (RefEF*)e; E->on();
    Vec10 st; for (int k = 0; k < 10; ++k) st[k] = state10[k];
    E->fhs[idx]->setState(st);
}

// level-0 {I,dx,dy} image of a frame: the I plane goes through the reference's FrameHessian::makeImages (HessianBlocks.cpp:107-167), whose
// gradients are compared with the given ones on the rows it defines (1 .. h-2); rows 0 and h-1, which makeImages leaves uninitialised, take
// the given values so that both sides read the same memory
void ref_ef_set_frame_image(void* e, int idx, const float* dI) {
    RefEF* E = (RefEF*)e; E->on();
    FrameHessian* fh = E->fhs[idx];
    const int w = E->g.w, h = E->g.h;
    std::vector<float> color((size_t)w * h);
    for (size_t i = 0; i < color.size(); ++i) color[i] = dI[3 * i];
    if (fh->dIp[0]) { for (int l = 0; l < pyrLevelsUsed; ++l) { delete[] fh->dIp[l]; delete[] fh->absSquaredGrad[l]; } }
    fh->makeImages(color.data(), &E->fs->Hcalib);
    for (int i = w; i < w * (h - 1); ++i)
        for (int k = 1; k < 3; ++k) {
            const double d = std::fabs((double)fh->dI[i][k] - (double)dI[3 * (size_t)i + k]);
            if (d > E->image_grad_maxdiff) E->image_grad_maxdiff = d;
        }
    for (int i = 0; i < w; ++i)
        for (int k = 1; k < 3; ++k) { fh->dI[i][k] = dI[3 * (size_t)i + k]; fh->dI[(size_t)w * (h - 1) + i][k] = dI[3 * ((size_t)w * (h - 1) + i) + k]; }
}

void ref_ef_set_points(void* e, int nP, const int* host, const float* u, const float* v, const float* idepth, const float* idepth_zero,
                       const float* color8, const float* weights8, const uint8_t* hasDepthPrior, const uint8_t* isFromSensor) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    // (called again it APPENDS points -- activatePointsMT's insertPoint, FullSystem.cpp:690-699; host = index in the window as it is now)
    for (int i = 0; i < nP; ++i) {
        FrameHessian* fh = fs->frameHessians[host[i]];
        // ImmaturePoint constructor (ImmaturePoint.cpp:8-40): colour and weights of the 8 pattern pixels from the host image
        ImmaturePoint ip((int)u[i], (int)v[i], fh, 1, &fs->Hcalib);
        ip.idepth_min = ip.idepth_max = idepth[i];
        ip.type = ImmaturePoint::CORNER;
        ip.isFromSensor = isFromSensor[i] != 0;
        bool same = true;
        for (int k = 0; k < 8; ++k) same = same && ip.color[k] == color8[8 * i + k] && ip.weights[k] == weights8[8 * i + k];
        if (!same) E->color_mismatch++;
        for (int k = 0; k < 8; ++k) { ip.color[k] = color8[8 * i + k]; ip.weights[k] = weights8[8 * i + k]; }
        ip.u = u[i]; ip.v = v[i];
        PointHessian* ph = new PointHessian(&ip, &fs->Hcalib);      // HessianBlocks.cpp:15-47
        ph->setIdepth(idepth[i]);
        ph->setIdepthZero(idepth_zero[i]);
        ph->hasDepthPrior = hasDepthPrior[i] != 0;
        ph->isFromSensor = isFromSensor[i] != 0;
        ph->idepth_fromSensor = idepth[i];
        ph->setPointStatus(PointHessian::ACTIVE);
        ph->step = 0; ph->step_backup = 0; ph->idepth_backup = idepth[i];
        ph->lastResiduals[0] = std::pair<PointFrameResidual*, ResState>((PointFrameResidual*)0, ResState::OOB);
        ph->lastResiduals[1] = std::pair<PointFrameResidual*, ResState>((PointFrameResidual*)0, ResState::OOB);
        ph->idx = (int)fh->pointHessians.size();
        fh->pointHessians.push_back(ph);
        fs->ef->insertPoint(ph);                                     // EFPoint::takeData: priorF, deltaF
        ph->efPoint->bdSumF = 0; ph->efPoint->HdiF = 0;
        ph->efPoint->Hdd_accLF = 0; ph->efPoint->bd_accLF = 0; ph->efPoint->Hcd_accLF.setZero();
        ph->efPoint->Hdd_accAF = 0; ph->efPoint->bd_accAF = 0; ph->efPoint->Hcd_accAF.setZero();
        E->phs.push_back(ph);
    }
    fs->ef->makeIDX();
}

void ref_ef_set_point_idepth(void* e, const float* idepth, const float* idepth_zero) {
    RefEF* E = (RefEF*)e; E->on();
    for (size_t i = 0; i < E->phs.size(); ++i) { E->phs[i]->setIdepth(idepth[i]); E->phs[i]->setIdepthZero(idepth_zero[i]); }
}

void ref_ef_set_residuals(void* e, int nR, const int* point, const int* target, const int* state_state, const uint8_t* hasMatcher,
                          const double* matcher2, const uint8_t* isLinearized, const uint8_t* isActive) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    // (called again it APPENDS residuals; point = index over all points ever set, target = index in the window as it is now)
    // the matcher table of each point first: PointFrameResidual's constructor looks its target up there (Residuals.cpp:46-58)
    for (int i = 0; i < nR; ++i) {
        if (!hasMatcher[i]) continue;
        PointHessian* ph = E->phs[point[i]];
        FrameHessian* tg = fs->frameHessians[target[i]];
        ph->matcher.targetFrames.push_back(tg);
        ph->matcher.pxs.push_back(Eigen::Vector2d(matcher2[2 * i], matcher2[2 * i + 1]));
        ph->matcher.frameIDs.push_back(tg->shell->id);
    }
    for (int i = 0; i < nR; ++i) {
        PointHessian* ph = E->phs[point[i]];
        PointFrameResidual* r = new PointFrameResidual(ph, ph->host, fs->frameHessians[target[i]]);
        r->setState((ResState)state_state[i]);
        ph->residuals.push_back(r);
        fs->ef->insertResidual(r);
        ph->lastResiduals[1] = ph->lastResiduals[0];                 // FullSystem.cpp:1093-1097
        ph->lastResiduals[0] = std::pair<PointFrameResidual*, ResState>(r, ResState::IN);
        r->efResidual->isLinearized = isLinearized[i] != 0;
        r->efResidual->isActiveAndIsGoodNEW = isActive[i] != 0;
        r->efResidual->res_toZeroF.setZero();
        r->efResidual->JpJdF.setZero();
        std::memset((void*)r->J, 0, sizeof(RawResidualJacobian));
        std::memset((void*)r->efResidual->J, 0, sizeof(RawResidualJacobian));
        E->prs.push_back(r);
        E->r_point.push_back(point[i]);
    }
    fs->ef->makeIDX();
    E->removed_by_finish.assign(E->prs.size(), 0);
}

void ref_ef_set_marg_prior(void* e, const double* HM, const double* bM) {
    RefEF* E = (RefEF*)e; E->on();
    EnergyFunctional* ef = E->fs->ef;
    const int n = CPARS + 6 * ef->nFrames;
    assert(ef->HM.rows() == n);
    for (int r = 0; r < n; ++r) { ef->bM[r] = bM[r]; for (int c = 0; c < n; ++c) ef->HM(r, c) = HM[(size_t)r * n + c]; }
}

// FullSystem::getNullspaces (FullSystemOptimize.cpp:548-588) over the null-space columns FrameHessian::setStateZero computed
int ref_ef_compute_nullspaces(void* e, double* out) {
    RefEF* E = (RefEF*)e; E->on();
    EnergyFunctional* ef = E->fs->ef;
    ef->lastNullspaces_forLogging = E->fs->getNullspaces(ef->lastNullspaces_pose, ef->lastNullspaces_scale, ef->lastNullspaces_affA, ef->lastNullspaces_affB);
    E->custom_nullspaces = false;
    const int n = CPARS + 6 * ef->nFrames;
    if (out) {
        for (int j = 0; j < 6; ++j) for (int i = 0; i < n; ++i) out[(size_t)j * n + i] = ef->lastNullspaces_pose[j][i];
        for (int i = 0; i < n; ++i) out[(size_t)6 * n + i] = ef->lastNullspaces_scale[0][i];
    }
    return 7;
}
// arbitrary vectors instead (first k-1: pose set, last: scale set); solve_system then calls EnergyFunctional::solveSystemF directly
void ref_ef_set_nullspaces(void* e, int k, const double* ns) {
    RefEF* E = (RefEF*)e; E->on();
    EnergyFunctional* ef = E->fs->ef;
    const int n = CPARS + 6 * ef->nFrames;
    ef->lastNullspaces_pose.clear(); ef->lastNullspaces_scale.clear();
    for (int j = 0; j < k; ++j) {
        VecX v(n);
        for (int i = 0; i < n; ++i) v[i] = ns[(size_t)j * n + i];
        if (j + 1 < k) ef->lastNullspaces_pose.push_back(v); else ef->lastNullspaces_scale.push_back(v);
    }
    E->custom_nullspaces = true;
}

void ref_ef_set_precalc(void* e) { RefEF* E = (RefEF*)e; E->on(); E->fs->setPrecalcValues(); }          // FullSystem.cpp:1358-1368
void ref_ef_set_adjoints(void* e) { RefEF* E = (RefEF*)e; E->on(); E->fs->ef->setAdjointsF(&E->fs->Hcalib); }

double ref_ef_linearize_all(void* e) {                       // FullSystem::linearizeAll(false)
    RefEF* E = (RefEF*)e; E->on();
    fill_active(E);
    return E->fs->linearizeAll(false)[0];
}
void ref_ef_apply_res(void* e) {                             // FullSystem::applyRes_Reductor(true, 0, n, 0, 0)
    RefEF* E = (RefEF*)e; E->on();
    fill_active(E);
    E->fs->applyRes_Reductor(true, 0, (int)E->fs->activeResiduals.size(), 0, 0);
}
void ref_ef_reset_oob(void* e, const uint8_t* mask) {       // PointFrameResidual::resetOOB (Residuals.h:70-76)
    RefEF* E = (RefEF*)e; E->on();
    for (size_t i = 0; i < E->prs.size(); ++i) {
        PointFrameResidual* r = E->prs[i];
        if (!r || (mask && !mask[E->r_point[i]]) || r->efResidual->isLinearized) continue;
        r->resetOOB();
    }
}

void ref_ef_solve_system(void* e, int iteration, double lambda) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs; EnergyFunctional* ef = fs->ef;
    // the two accumulations solveSystemF is about to repeat (they are idempotent), for the getters: HA_top / H_sc are its locals
    ef->accumulateAF_MT(E->HA, E->bA, false);
    { MatXX HL; VecX bL; ef->accumulateLF_MT(HL, bL, false); }
    ef->accumulateSCF_MT(E->Hsc, E->bsc, false);
    if (E->custom_nullspaces) ef->solveSystemF(iteration, lambda, &fs->Hcalib);
    else fs->solveSystem(iteration, lambda);                  // FullSystemOptimize.cpp:504-513
    E->HFinal = ef->lastHS; E->bFinal = ef->lastbS;
}

int ref_ef_dim(void* e) { RefEF* E = (RefEF*)e; return CPARS + 6 * E->fs->ef->nFrames; }
void ref_ef_get_system(void* e, double* HA, double* bA, double* Hsc, double* bsc, double* HFinal, double* bFinal, double* x) {
    RefEF* E = (RefEF*)e;
    to_rowmajor(E->HA, HA); to_rowmajor(E->bA, bA); to_rowmajor(E->Hsc, Hsc); to_rowmajor(E->bsc, bsc);
    to_rowmajor(E->HFinal, HFinal); to_rowmajor(E->bFinal, bFinal); to_rowmajor(E->fs->ef->lastX, x);
}

static void put_J(const RawResidualJacobian* J, float* o) {
    o[0] = J->resF[0]; o[1] = J->resF[1];
    for (int k = 0; k < 6; ++k) { o[2 + k] = J->Jpdxi[0][k]; o[8 + k] = J->Jpdxi[1][k]; }
    for (int k = 0; k < 4; ++k) { o[14 + k] = J->Jpdc[0][k]; o[18 + k] = J->Jpdc[1][k]; }
    o[22] = J->Jpdd[0]; o[23] = J->Jpdd[1];
}
// which = 0: PointFrameResidual::J (what linearize just wrote), 1: EFResidual::J (what takeDataF swapped in)
void ref_ef_get_residual_J(void* e, int which, float* out24) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->prs.size(); ++i) {
        float* o = out24 + 24 * i;
        if (!E->prs[i]) { std::memset(o, 0, 96); continue; }
        put_J(which ? E->prs[i]->efResidual->J : E->prs[i]->J, o);
    }
}
void ref_ef_get_residual_state(void* e, int* state_state, int* state_new, double* energy_new, double* energy_with_outlier, uint8_t* isActive) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->prs.size(); ++i) {
        PointFrameResidual* r = E->prs[i];
        if (state_state) state_state[i] = r ? (int)r->state_state : -1;
        if (state_new) state_new[i] = r ? (int)r->state_NewState : -1;
        if (energy_new) energy_new[i] = r ? r->state_NewEnergy : 0;
        if (energy_with_outlier) energy_with_outlier[i] = r ? r->state_NewEnergyWithOutlier : 0;
        if (isActive) isActive[i] = r ? (uint8_t)r->efResidual->isActive() : 0;
    }
}
void ref_ef_get_center_projected(void* e, float* out3) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->prs.size(); ++i) for (int k = 0; k < 3; ++k) out3[3 * i + k] = E->prs[i] ? E->prs[i]->centerProjectedTo[k] : 0.f;
}
void ref_ef_get_points(void* e, float* out9) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->phs.size(); ++i) {
        const PointHessian* ph = E->phs[i];
        float* o = out9 + 9 * i;
        if (!ph) { for (int k = 0; k < 9; ++k) o[k] = NAN; continue; }
        const EFPoint* p = ph->efPoint;
        o[0] = p->Hdd_accAF; o[1] = p->bd_accAF;
        for (int k = 0; k < 4; ++k) o[2 + k] = p->Hcd_accAF[k];
        o[6] = p->HdiF; o[7] = p->bdSumF; o[8] = ph->step;
    }
}
void ref_ef_get_frame_steps(void* e, double* steps6, double* calibStep4) {
    RefEF* E = (RefEF*)e;
    for (size_t h = 0; h < E->fhs.size(); ++h) for (int i = 0; i < 6; ++i) steps6[6 * h + i] = E->fhs[h]->step[i];
    for (int i = 0; i < 4; ++i) calibStep4[i] = E->fs->Hcalib.step[i];
}
void ref_ef_get_top_acc(void* e, float* out) {
    RefEF* E = (RefEF*)e; const int nF = E->fs->ef->nFrames;
    for (int k = 0; k < nF * nF; ++k)
        for (int r = 0; r < 13; ++r) for (int c = 0; c < 13; ++c) out[(size_t)k * 169 + r * 13 + c] = E->fs->ef->accSSE_top_A->acc[0][k].H(r, c);
}
void ref_ef_get_precalc(void* e, int h, int t, float* out) {  // KRKi9, Kt3, R0 9, t0 3, aff2, b0 = 27 (row-major 3x3s)
    RefEF* E = (RefEF*)e;
    const FrameFramePrecalc& P = E->fhs[h]->targetPrecalc[t];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { out[r * 3 + c] = P.PRE_KRKiTll(r, c); out[12 + r * 3 + c] = P.PRE_RTll_0(r, c); }
    for (int k = 0; k < 3; ++k) { out[9 + k] = P.PRE_KtTll[k]; out[21 + k] = P.PRE_tTll_0[k]; }
    out[24] = P.PRE_aff_mode[0]; out[25] = P.PRE_aff_mode[1]; out[26] = P.PRE_b0_mode;
}
void ref_ef_get_adjoints(void* e, double* adHost, double* adTarget) {
    RefEF* E = (RefEF*)e; const int nF = E->fs->ef->nFrames;
    for (int k = 0; k < nF * nF; ++k) { to_rowmajor(E->fs->ef->adHost[k], adHost + 36 * (size_t)k); to_rowmajor(E->fs->ef->adTarget[k], adTarget + 36 * (size_t)k); }
}
int ref_ef_res_in_A(void* e) { return ((RefEF*)e)->fs->ef->resInA; }
void ref_ef_get_sc_acc(void* e, float* accE, float* accEB, float* accD, float* Hcc, float* bc) {
    RefEF* E = (RefEF*)e; const int nF = E->fs->ef->nFrames;
    AccumulatedSCHessianSSE* S = E->fs->ef->accSSE_bot;
    for (int k = 0; k < nF * nF; ++k) {
        for (int r = 0; r < 8; ++r) { for (int c = 0; c < 4; ++c) accE[(size_t)k * 32 + r * 4 + c] = S->accE[0][k].A1m(r, c); accEB[(size_t)k * 8 + r] = S->accEB[0][k].A1m[r]; }
    }
    for (int k = 0; k < nF * nF * nF; ++k) for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) accD[(size_t)k * 64 + r * 8 + c] = S->accD[0][k].A1m(r, c);
    for (int r = 0; r < 4; ++r) { for (int c = 0; c < 4; ++c) Hcc[r * 4 + c] = S->accHcc[0].A1m(r, c); bc[r] = S->accbc[0].A1m[r]; }
}
void ref_ef_get_frame_energy_th(void* e, float* th) { RefEF* E = (RefEF*)e; for (size_t i = 0; i < E->fhs.size(); ++i) th[i] = E->fhs[i]->frameEnergyTH; }
void ref_ef_get_evalPT(void* e, int idx, double* q4t3, double* state_zero10) {
    RefEF* E = (RefEF*)e;
    pose_to7(E->fhs[idx]->worldToCam_evalPT, q4t3);
    for (int i = 0; i < 10; ++i) state_zero10[i] = E->fhs[idx]->state_zero[i];
}
void ref_ef_get_state(void* e, double* value_scaled4, double* state10, float* idepth) {
    RefEF* E = (RefEF*)e;
    for (int i = 0; i < 4; ++i) value_scaled4[i] = E->fs->Hcalib.value_scaled[i];
    for (size_t h = 0; h < E->fhs.size(); ++h) for (int i = 0; i < 10; ++i) state10[10 * h + i] = E->fhs[h]->state[i];
    for (size_t i = 0; i < E->phs.size(); ++i) idepth[i] = E->phs[i] ? E->phs[i]->idepth : NAN;      // (NaN: the point has left the window)
}
void ref_ef_get_marg_prior(void* e, double* HM, double* bM) { RefEF* E = (RefEF*)e; to_rowmajor(E->fs->ef->HM, HM); to_rowmajor(E->fs->ef->bM, bM); }
void ref_ef_get_res_toZero(void* e, float* out2, uint8_t* isLinearized) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->prs.size(); ++i) {
        PointFrameResidual* r = E->prs[i];
        out2[2 * i] = r ? r->efResidual->res_toZeroF[0] : 0.f; out2[2 * i + 1] = r ? r->efResidual->res_toZeroF[1] : 0.f;
        isLinearized[i] = r ? (uint8_t)r->efResidual->isLinearized : 1;
    }
}
void ref_ef_get_adHTdeltaF(void* e, float* out) {
    RefEF* E = (RefEF*)e; const int nF = E->fs->ef->nFrames;
    for (int k = 0; k < nF * nF; ++k) for (int i = 0; i < 6; ++i) out[6 * (size_t)k + i] = E->fs->ef->adHTdeltaF[k][i];
}
void ref_ef_get_frame_prior(void* e, int idx, double* prior6, double* delta_prior6) {
    RefEF* E = (RefEF*)e; const EFFrame* f = E->fhs[idx]->efFrame;
    for (int i = 0; i < 6; ++i) { prior6[i] = f->prior[i]; delta_prior6[i] = f->delta_prior[i]; }
}
double ref_ef_calc_L_energy(void* e) { RefEF* E = (RefEF*)e; E->on(); return E->fs->calcLEnergy(); }
double ref_ef_calc_M_energy(void* e) { RefEF* E = (RefEF*)e; E->on(); return E->fs->calcMEnergy(); }

// EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:45-55) on the active residuals of the flagged points (FullSystem.cpp:775-783)
void ref_ef_fix_linearization(void* e, const uint8_t* mask) {
    RefEF* E = (RefEF*)e; E->on();
    for (size_t pi = 0; pi < E->phs.size(); ++pi) {
        if (!mask[pi]) continue;
        for (EFResidual* r : E->phs[pi]->efPoint->residualsAll) if (r->isActive()) r->fixLinearizationF(E->fs->ef);
    }
}

// EnergyFunctional::marginalizePointsF / dropPointsF (EnergyFunctional.cpp:514-597) for the flagged points.  The reference deletes the
// EFPoint / EFResidual objects; the PointHessian and its PointFrameResidual objects are detached the way flagPointsForRemoval's callers
// leave them (FullSystem.cpp:790-800): moved out of the host's active list.
void ref_ef_marginalize_points(void* e, const uint8_t* marg, const uint8_t* drop) {
    RefEF* E = (RefEF*)e; E->on();
    EnergyFunctional* ef = E->fs->ef;
    for (size_t pi = 0; pi < E->phs.size(); ++pi) {
        PointHessian* ph = E->phs[pi];
        if (!ph || !ph->efPoint) continue;
        if (marg[pi]) ph->efPoint->stateFlag = EFPointStatus::PS_MARGINALIZE;
        else if (drop && drop[pi]) ph->efPoint->stateFlag = EFPointStatus::PS_DROP;
        else continue;
        FrameHessian* fh = ph->host;
        for (size_t k = 0; k < fh->pointHessians.size(); ++k) if (fh->pointHessians[k] == ph) { fh->pointHessians[k] = fh->pointHessians.back(); fh->pointHessians.pop_back(); break; }
        (marg[pi] ? fh->pointHessiansMarginalized : fh->pointHessiansOut).push_back(ph);
    }
    ef->marginalizePointsF();
    ef->dropPointsF();
    for (size_t i = 0; i < E->prs.size(); ++i) {
        const int pi = E->r_point[i];
        if (marg[pi] || (drop && drop[pi])) E->prs[i] = nullptr;      // efResidual is gone; the objects die with their PointHessian
    }
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:434-512) on a COPY of HM / bM semantics: the oracle's function leaves the window
// untouched and returns the reduced prior, so the reference's member (which also removes the frame) runs on a throw-away EnergyFunctional
// carrying this window's HM, bM and the frame's prior / delta_prior.
void ref_ef_marginalize_frame(void* e, int idx, double* HM_out, double* bM_out) {
    RefEF* E = (RefEF*)e; E->on();
    EnergyFunctional* src = E->fs->ef;
    const int nF = src->nFrames;
    EnergyFunctional* tmp = new EnergyFunctional();
    tmp->red = src->red;
    std::vector<FrameHessian*> dummies;
    for (int i = 0; i < nF; ++i) {
        FrameHessian* fh = new FrameHessian();
        fh->shell = E->shells[i]; fh->dI = 0;
        for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
        fh->ab_exposure = E->fhs[i]->ab_exposure; fh->frameID = E->fhs[i]->frameID; fh->idx = i;
        fh->worldToCam_evalPT = E->fhs[i]->worldToCam_evalPT;
        fh->setStateZero(E->fhs[i]->state_zero);
        fh->setState(E->fhs[i]->state);
        dummies.push_back(fh);
        tmp->insertFrame(fh, &E->fs->Hcalib);
        fh->efFrame->prior = E->fhs[i]->efFrame->prior;
        fh->efFrame->delta_prior = E->fhs[i]->efFrame->delta_prior;
        fh->efFrame->delta = E->fhs[i]->efFrame->delta;
    }
    tmp->HM = src->HM; tmp->bM = src->bM;
    tmp->setDeltaF(&E->fs->Hcalib);
    tmp->marginalizeFrame(dummies[idx]->efFrame);
    to_rowmajor(tmp->HM, HM_out); to_rowmajor(tmp->bM, bM_out);
    dummies[idx]->efFrame = 0;
    delete tmp;
    for (int i = 0; i < nF; ++i) { for (int l = 0; l < pyrLevelsUsed; ++l) { dummies[i]->dIp[l] = 0; dummies[i]->absSquaredGrad[l] = 0; } delete dummies[i]; }
    // restore the flags the throw-away object's insertFrame / makeIDX toggled for the real one
    src->setAdjointsF(&E->fs->Hcalib);
    src->makeIDX();
    src->setDeltaF(&E->fs->Hcalib);
}

// FullSystem::optimize (FullSystemOptimize.cpp:344-502), the WHOLE function: loop and tail (setEvalPT of the newest frame, adjoints, precalc,
// linearizeAll(true) which deletes the residuals that are not active).  The reference reports its accept / reject decisions through printf
// only; its console output is captured and returned by ref_ef_last_log.  Returns the function's return value (the RMSE).
// min_its >= 0 sets setting_minOptIterations for the call (settings.cpp:56 default 1; = mnumOptIts makes the loop run exactly that many
// bodies, the timing protocol of bench.py's cpu_baseline leg); seconds_out: wall time of the optimize() call alone.
static int g_min_its = -1;
static double g_last_seconds = 0, g_last_optimize_seconds = 0;
void ref_ef_set_min_its(int n) { g_min_its = n; }
double ref_ef_last_seconds() { return g_last_seconds; }
double ref_ef_last_optimize_seconds() { return g_last_optimize_seconds; }   // of the optimize call inside the last ref_ef_keyframe_tail
double ref_ef_optimize_full(void* e, int mnumOptIts) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    std::vector<PointFrameResidual*> before = E->prs;
    float rmse = 0;
    const int saved_min = setting_minOptIterations;
    if (g_min_its >= 0) setting_minOptIterations = g_min_its;
    setting_debugout_runquiet = false;
    E->last_log = capture_stdout([&] {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        rmse = fs->optimize(mnumOptIts);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_last_seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    });
    setting_minOptIterations = saved_min;
    setting_debugout_runquiet = true;
    refresh_live(E);
    for (size_t i = 0; i < E->prs.size(); ++i) E->removed_by_finish[i] = (before[i] && !E->prs[i]) ? 1 : 0;
    return rmse;
}
int ref_ef_last_log(void* e, char* buf, int cap) {
    RefEF* E = (RefEF*)e;
    const int n = (int)E->last_log.size();
    if (buf && cap > 0) { const int m = n < cap - 1 ? n : cap - 1; std::memcpy(buf, E->last_log.data(), (size_t)m); buf[m] = 0; }
    return n;
}
void ref_ef_get_removed(void* e, uint8_t* removed) { RefEF* E = (RefEF*)e; std::memcpy(removed, E->removed_by_finish.data(), E->removed_by_finish.size()); }
void ref_ef_get_point_stats(void* e, float* maxRelBaseline, int* numGoodResiduals) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->phs.size(); ++i) { maxRelBaseline[i] = E->phs[i] ? E->phs[i]->maxRelBaseline : NAN; numGoodResiduals[i] = E->phs[i] ? E->phs[i]->numGoodResiduals : -1; }
}
// FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:18-185) for n immature points against the frames / precalc of the handle
// (ref_ef_set_precalc first).  result[i]: 0 = not well constrained, -1 = rejected, 1 = activated (idepth[i] = its idepth); res_state[n][nF]
void ref_ef_optimize_immature(void* e, int n, const int* host, const float* u, const float* v, const float* idepth_min, const float* idepth_max,
                              const float* energyTH, const float* color8, const float* weights8, const uint8_t* isFromSensor, int minObs,
                              int* result, float* idepth, int* res_state) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    const int nF = (int)E->fhs.size();
    std::vector<ImmaturePointTemporaryResidual> tr((size_t)nF);
    for (int i = 0; i < n; ++i) {
        ImmaturePoint ip((int)u[i], (int)v[i], E->fhs[host[i]], 1, &fs->Hcalib);
        ip.u = u[i]; ip.v = v[i];
        ip.idepth_min = idepth_min[i]; ip.idepth_max = idepth_max[i]; ip.energyTH = energyTH[i];
        for (int k = 0; k < 8; ++k) { ip.color[k] = color8[8 * i + k]; ip.weights[k] = weights8[8 * i + k]; }
        ip.isFromSensor = isFromSensor[i] != 0;
        ip.type = ImmaturePoint::CORNER;
        PointHessian* p = fs->optimizeImmaturePoint(&ip, minObs, tr.data());
        idepth[i] = NAN;
        if (p == 0) result[i] = 0;
        else if (p == (PointHessian*)((long)(-1))) result[i] = -1;
        else { result[i] = 1; idepth[i] = p->idepth; delete p; }
        for (int t = 0; t < nF; ++t) res_state[(size_t)i * nF + t] = -1;
        int k = 0;
        for (int t = 0; t < nF; ++t) if (t != host[i]) res_state[(size_t)i * nF + t] = (int)tr[k++].state_state;
    }
}

// The part of FullSystem::makeKeyFrame that follows the activation of new points (FullSystem.cpp:1133-1178), the reference's own statements in
// the reference's own order: optimize, removeOutliers, setCoarseTrackingRef of the tracker for the next key-frame (makeCoarseDepthL0 reads
// centerProjectedTo, lastResiduals and EFPoint::HdiF of what optimize left), flagPointsForRemoval, dropPointsF, getNullspaces,
// marginalizePointsF, marginalizeFrame for every flagged frame.  flag[k] != 0 marks window frame k like flagFramesForMarginalization would.
// Returns optimize's return value.  Afterwards the handle's frame / point lists describe the window that is left: points that were
// dropped or marginalised read as NaN / -1 in the getters.
static void refresh_after_keyframe(RefEF* E) {
    FullSystem* fs = E->fs;
    std::set<PointHessian*> live;
    for (FrameHessian* fh : fs->frameHessians) for (PointHessian* ph : fh->pointHessians) live.insert(ph);
    for (size_t i = 0; i < E->phs.size(); ++i) if (E->phs[i] && !live.count(E->phs[i])) E->phs[i] = nullptr;
    std::set<PointFrameResidual*> liver;
    for (PointHessian* ph : live) for (PointFrameResidual* r : ph->residuals) liver.insert(r);
    for (size_t i = 0; i < E->prs.size(); ++i) if (E->prs[i] && !liver.count(E->prs[i])) E->prs[i] = nullptr;
    E->fhs = fs->frameHessians;
}
double ref_ef_keyframe_tail(void* e, int mnumOptIts, const uint8_t* flag) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    std::vector<PointFrameResidual*> before = E->prs;
    float rmse = 0;
    setting_debugout_runquiet = false;
    E->last_log = capture_stdout([&] {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (size_t k = 0; k < fs->frameHessians.size(); ++k) fs->frameHessians[k]->flaggedForMarginalization = flag && flag[k];
        static const bool tr = getenv("REF_GLUE_TRACE") != nullptr;
        if (tr) signal(SIGSEGV, [](int) { void* bt[48]; const int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 2); _exit(139); });
#define KF_STAGE(name) do { if (tr) fprintf(stderr, "[ref glue] key-frame tail: %s\n", name); } while (0)
        KF_STAGE("optimize");
        const int saved_min = setting_minOptIterations;
        if (g_min_its >= 0) setting_minOptIterations = g_min_its;
        struct timespec o0, o1;
        clock_gettime(CLOCK_MONOTONIC, &o0);
        rmse = fs->optimize(mnumOptIts);                                                    // :1134
        clock_gettime(CLOCK_MONOTONIC, &o1);
        g_last_optimize_seconds = (o1.tv_sec - o0.tv_sec) + 1e-9 * (o1.tv_nsec - o0.tv_nsec);
        setting_minOptIterations = saved_min;
        KF_STAGE("removeOutliers");
        fs->removeOutliers();                                                               // :1138
        KF_STAGE("setCoarseTrackingRef");
        fs->coarseTracker_forNewKF->makeK(&fs->Hcalib);                                     // :1143-1144
        fs->coarseTracker_forNewKF->setCoarseTrackingRef(fs->frameHessians);
        KF_STAGE("flagPointsForRemoval");
        fs->flagPointsForRemoval();                                                         // :1152
        KF_STAGE("dropPointsF");
        fs->ef->dropPointsF();
        fs->getNullspaces(fs->ef->lastNullspaces_pose, fs->ef->lastNullspaces_scale, fs->ef->lastNullspaces_affA, fs->ef->lastNullspaces_affB);
        KF_STAGE("marginalizePointsF");
        fs->ef->marginalizePointsF();
        KF_STAGE("marginalizeFrame");
        for (unsigned int i = 0; i < fs->frameHessians.size(); i++)                         // :1169-1171
            if (fs->frameHessians[i]->flaggedForMarginalization) { fs->marginalizeFrame(fs->frameHessians[i]); i = 0; }
        KF_STAGE("done");
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_last_seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    });
    setting_debugout_runquiet = true;
    refresh_after_keyframe(E);
    E->removed_by_finish.assign(E->prs.size(), 0);
    for (size_t i = 0; i < E->prs.size(); ++i) E->removed_by_finish[i] = (before[i] && !E->prs[i]) ? 1 : 0;
    return rmse;
}
// the window that is left: number of frames, and for every point ever set the index of its host frame in the window now (-1: the point is gone)
int ref_ef_window_frames(void* e) { return (int)((RefEF*)e)->fs->frameHessians.size(); }
void ref_ef_point_hosts(void* e, int* host) {
    RefEF* E = (RefEF*)e;
    for (size_t i = 0; i < E->phs.size(); ++i) host[i] = E->phs[i] ? E->phs[i]->host->idx : -1;
}
// the tracking template setCoarseTrackingRef built for the next key-frame (CoarseTracker::makeCoarseDepthL0, CoarseTracker.cpp:258-425): pc_n
// of every level, and level `lvl`'s points (caller allocates w*h floats per array)
void ref_ef_tracking_ref(void* e, int lvl, int* pc_n5, float* pc_u, float* pc_v, float* pc_idepth, float* pc_color) {
    RefEF* E = (RefEF*)e;
    CoarseTracker* ct = E->fs->coarseTracker_forNewKF;
    for (int l = 0; l < 5; ++l) pc_n5[l] = l < pyrLevelsUsed ? ct->pc_n[l] : 0;
    const int n = ct->pc_n[lvl];
    std::memcpy(pc_u, ct->pc_u[lvl], sizeof(float) * n); std::memcpy(pc_v, ct->pc_v[lvl], sizeof(float) * n);
    std::memcpy(pc_idepth, ct->pc_idepth[lvl], sizeof(float) * n); std::memcpy(pc_color, ct->pc_color[lvl], sizeof(float) * n);
}
void ref_ef_get_marg_prior_dim(void* e, int* n) { *n = (int)((RefEF*)e)->fs->ef->HM.rows(); }
void* ref_ef_full_system(void* e) { return ((RefEF*)e)->fs; }

// the EnergyFunctional object of the window (key of the drop-in's side table, oracle/dropin/EnergyFunctionalGPU.cpp)
void* ref_ef_energy_functional(void* e) { RefEF* E = (RefEF*)e; return E->fs ? (void*)E->fs->ef : nullptr; }

// diagnostics of the loading step
void ref_ef_load_report(void* e, int* color_mismatch, double* image_grad_maxdiff) {
    RefEF* E = (RefEF*)e; *color_mismatch = E->color_mismatch; *image_grad_maxdiff = E->image_grad_maxdiff;
}

// ---- the per-frame rows around the window: immature points, FullSystem::traceNewCoarse, FullSystem::activatePointsMT ----------------------
// Immature points on the window's key-frames, created by the reference's own constructor (ImmaturePoint.cpp:17-45: colour, weights, gradH, energyTH from
// the host's image) the way FullSystem::makeNewTraces creates them (FullSystem.cpp:1273-1356; a point whose energyTH is not finite is deleted like there),
// with the trace state given (what earlier frames would have left; idepth_max = NaN and status 5 = a point never traced).  Returns the number kept.
int ref_ef_add_immature(void* e, int n, const int* host, const int* u, const int* v, const float* my_type, const float* idepth_min, const float* idepth_max,
                        const float* quality, const int* status, const float* interval, const uint8_t* isFromSensor) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        FrameHessian* fh = fs->frameHessians[host[i]];
        ImmaturePoint* ip = new ImmaturePoint(u[i], v[i], fh, my_type[i], &fs->Hcalib);
        if (!std::isfinite(ip->energyTH)) { delete ip; continue; }
        ip->idepth_min = idepth_min[i]; ip->idepth_max = idepth_max[i]; ip->quality = quality[i];
        ip->lastTraceStatus = (ImmaturePointStatus)status[i]; ip->lastTracePixelInterval = interval[i];
        ip->lastTraceUV = Vec2f(-1, -1);
        ip->isFromSensor = isFromSensor[i] != 0;
        ip->idepth_fromSensor = 0.5f * (idepth_min[i] + idepth_max[i]);
        ip->type = ImmaturePoint::CORNER;
        fh->immaturePoints.push_back(ip);
        ++kept;
    }
    return kept;
}
// FullSystem::traceNewCoarse (FullSystem.cpp:519-553) on a NEW frame (not part of the window): its image through the reference's makeImages, its pose and
// brightness through setEvalPT_scaled like FullSystem::addActiveFrame's callers do before tracing (:1010-1017, :1040-1047)
void ref_ef_trace_new_frame(void* e, const float* color_lvl0, const double* camToWorld7, float exposure, double a, double b) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    FrameShell* sh = new FrameShell();
    static int next_new_frame_id = 100000;          // (every new frame its own id, like FullSystem::addActiveFrame's allFrameHistory.size())
    sh->id = sh->incoming_id = next_new_frame_id++;
    sh->camToWorld = pose_from7(camToWorld7);
    sh->aff_g2l = AffLight(a, b);
    FrameHessian* fh = new FrameHessian();
    fh->shell = sh; fh->dI = 0; fh->ab_exposure = exposure;
    for (int l = 0; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    std::vector<float> c(color_lvl0, color_lvl0 + (size_t)E->g.w * E->g.h);
    fh->makeImages(c.data(), &fs->Hcalib);
    for (int l = 0; l < pyrLevelsUsed; ++l) {      // rows 0 / h-1 of the gradient planes, which makeImages leaves uninitialised: zero on both sides
        const int wl = wG[l], hl = hG[l];
        for (int x = 0; x < wl; ++x) for (int k = 1; k < 3; ++k) { fh->dIp[l][x][k] = 0; fh->dIp[l][(size_t)wl * (hl - 1) + x][k] = 0; }
    }
    fh->setEvalPT_scaled(sh->camToWorld.inverse(), sh->aff_g2l);
    {   // (the call alone is timed: ref_ef_last_seconds)
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        E->last_log = capture_stdout([&] { fs->traceNewCoarse(fh); });
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_last_seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    }
    for (int l = pyrLevelsUsed; l < PYR_LEVELS; ++l) { fh->dIp[l] = 0; fh->absSquaredGrad[l] = 0; }
    delete fh;            // (~FrameHessian frees the pyramid levels in use)
    delete sh;
}
// the immature points in frameHessians / immaturePoints order (a slot the reference has emptied reads status -1); returns their number
int ref_ef_get_immature(void* e, int* host, float* u, float* v, float* idepth_min, float* idepth_max, float* quality, int* status, float* uv2, float* interval) {
    RefEF* E = (RefEF*)e;
    FullSystem* fs = E->fs;
    int n = 0;
    for (size_t h = 0; h < fs->frameHessians.size(); ++h)
        for (ImmaturePoint* ip : fs->frameHessians[h]->immaturePoints) {
            if (host) {
                host[n] = (int)h;
                if (!ip) { status[n] = -1; u[n] = v[n] = idepth_min[n] = idepth_max[n] = quality[n] = interval[n] = NAN; uv2[2 * n] = uv2[2 * n + 1] = NAN; }
                else {
                    u[n] = ip->u; v[n] = ip->v; idepth_min[n] = ip->idepth_min; idepth_max[n] = ip->idepth_max; quality[n] = ip->quality;
                    status[n] = (int)ip->lastTraceStatus; uv2[2 * n] = ip->lastTraceUV[0]; uv2[2 * n + 1] = ip->lastTraceUV[1]; interval[n] = ip->lastTracePixelInterval;
                }
            }
            ++n;
        }
    return n;
}
// FullSystem::activatePointsMT (FullSystem.cpp:569-717), the reference's own function: the distance map, the choice of the points to activate, their
// optimisation (activatePointsMT_Reductor -> optimizeImmaturePoint), insertPoint / insertResidual of the ones that made it, deletion of the others.
// The new PointHessians join the handle's point list (indices behind the ones set so far, in frame / pointHessians order); returns their number.
int ref_ef_activate_points(void* e) {
    RefEF* E = (RefEF*)e; E->on();
    FullSystem* fs = E->fs;
    {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        E->last_log = capture_stdout([&] { fs->activatePointsMT(); });
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_last_seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    }
    fs->ef->makeIDX();                                                                      // FullSystem.cpp:1103
    std::set<PointHessian*> known(E->phs.begin(), E->phs.end());
    int added = 0;
    for (FrameHessian* fh : fs->frameHessians)
        for (PointHessian* ph : fh->pointHessians)
            if (!known.count(ph)) {
                E->phs.push_back(ph);
                for (PointFrameResidual* r : ph->residuals) { E->prs.push_back(r); E->r_point.push_back((int)E->phs.size() - 1); }
                ++added;
            }
    return added;
}
// u, v, host (window index) and the target frames (bit t set) of the LAST n points of the handle's point list
void ref_ef_get_new_points(void* e, int n, float* u, float* v, int* host, unsigned* targets) {
    RefEF* E = (RefEF*)e;
    const size_t n0 = E->phs.size() - (size_t)n;
    for (int i = 0; i < n; ++i) {
        PointHessian* ph = E->phs[n0 + i];
        u[i] = ph->u; v[i] = ph->v; host[i] = ph->host->idx;
        unsigned m = 0;
        for (PointFrameResidual* r : ph->residuals) m |= 1u << r->target->idx;
        targets[i] = m;
    }
}
int ref_ef_num_points(void* e) { return (int)((RefEF*)e)->phs.size(); }
int ref_ef_num_residuals(void* e) { return (int)((RefEF*)e)->prs.size(); }

}  // extern "C"
