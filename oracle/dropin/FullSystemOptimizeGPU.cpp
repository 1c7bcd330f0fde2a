// oracle/dropin/FullSystemOptimizeGPU.cpp -- TEST INFRASTRUCTURE and the reference-side binding of INTEGRATION.md, form B, as a file that COMPILES.
//
// This translation unit DEFINES the member function
//     float sdv_loam::FullSystem::optimize(int mnumOptIts)                                      (FullSystemOptimize.cpp:344-502)
// against the reference's own, unmodified headers, with libsdvgn's C ABI behind it: sdvgn_ef_optimize + sdvgn_ef_optimize_finish on a window
// that STAYS ON THE DEVICE from key-frame to key-frame.  oracle/Makefile (target `dropin_opt`) links it instead of the reference's definition
// (that one symbol weakened in FullSystemOptimize.o), so that FullSystem::makeKeyFrame (FullSystem.cpp:1134) -- and everything the reference
// does around it: removeOutliers, setCoarseTrackingRef, flagPointsForRemoval, dropPointsF, marginalizePointsF, marginalizeFrame -- runs
// unchanged on what this function leaves behind.  That is the drop-in as the FAST path: PointFrameResidual::linearize, the accumulators, the
// solve and the accept / reject loop all run on the GPU (form A, EnergyFunctionalGPU.cpp, keeps linearize and the loop on the host and
// re-sends every plane per solve).
//
// Per call (= per key-frame):
//   1. the EnergyFunctional graph is walked once (the reference's own prologue walks it too, :353-372) and DIFFED against what the device
//      holds: frames that left / arrived (one image upload per NEW key-frame), points that left / arrived, residuals that were dropped /
//      inserted / got a matcher -- sent as edits (sdvgn_ef_remove_frame / _insert_frame / _remove_points / _insert_points / _drop_residuals /
//      _insert_residuals / _update_residuals) and committed on the device (sdvgn_ef_make_idx = makeIDX);
//   2. calib, frame states / linearisation points, HM / bM (marginalizePointsF and marginalizeFrame run on the host) are re-sent: a few kB;
//   3. sdvgn_ef_optimize runs the loop, sdvgn_ef_optimize_finish the tail's linearizeAll(true);
//   4. what the reference's optimize leaves behind is written back into ITS objects: CalibHessian value, FrameHessian state / step / the new
//      linearisation point of the newest frame (by the reference's own setEvalPT), frameEnergyTH, PointHessian idepth / idepth_zero / step /
//      idepth_hessian / maxRelBaseline / numGoodResiduals / lastResiduals, EFPoint HdiF / bdSumF / Hdd_accAF / bd_accAF / Hcd_accAF,
//      PointFrameResidual state_state / state_energy / state_New* / centerProjectedTo (by the reference's own projectPoint),
//      EFResidual::isActiveAndIsGoodNEW, the residuals of linearizeAll(true)'s toRemove list dropped exactly like :136-155 drops them,
//      ef->lastX / resInA, statistics_lastFineTrackRMSE, the shells' poses, isLost, the return value.
//   NOT written back: RawResidualJacobian contents (efResidual->J, r->J) and projectedTo[] -- nothing on the host reads them before the next
//   linearize rewrites them (flagPointsForRemoval re-linearises the residuals it fixes, FullSystem.cpp:771-783; projectedTo feeds debugPlot only).
// The two graph walks (1. and 4.) are chains of cache misses over the reference's heap objects and independent point by point: with
// SDVGN_DROPIN_THREADS=n they run on n host threads, everything that touches shared structures -- the edit calls, the toRemove surgery, the mirror's
// id map -- staying in the calling thread in the reference's order, so the result does not depend on n.  Default 1: measured on a 7 426-point window,
// 4 threads take 1.1 ms off the walks (residual diff 1.18 -> 0.81 ms, write-back walk 1.87 -> 1.12 ms) and the call is no faster, box to box -- what
// dominates is serial and the reference's own: dropResidual + deleteOut of the toRemove list (1.4-3.2 ms), the new / gone point lists (0.4-1.2 ms).
// SDVGN_DROPIN_TIMING=1 prints where a call's host time goes.
// There is no CPU fallback: a failing sdvgn_* call aborts like the reference's live asserts do.
#include "dropin_shared.hpp"
#include "FullSystem/ResidualProjections.h"
#include "FullSystem/Residuals.h"
#include "util/globalCalib.h"
#include "util/globalFuncs.h"

#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>

using namespace sdvgn_dropin;

namespace {

std::mutex g_mu;
std::map<const FullSystem*, GpuWindow> g_windows;   // (a member `GpuWindow gpu` in a real integration)

[[noreturn]] void fs_die(const char* what, int rc) { sdvgn_dropin::die("FullSystemOptimizeGPU", what, rc); }
#define GPU_CK(call) do { const int _rc = (call); if (_rc < 0) fs_die(#call, _rc); } while (0)
double us_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }

int dropin_threads() {
    const char* s = getenv("SDVGN_DROPIN_THREADS");
    const int v = s ? atoi(s) : 1;
    return v < 1 ? 1 : (v > 64 ? 64 : v);
}
// f(chunk, begin, end) over [0, n) in contiguous chunks, chunk c on its own thread (the last one on the caller's); returns the number of chunks
template <class F>
int for_chunks(size_t n, F f) {
    int T = dropin_threads();
    if ((size_t)T > n / 512 + 1) T = (int)(n / 512 + 1);
    if (T <= 1) { f(0, (size_t)0, n); return 1; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    const size_t per = (n + T - 1) / T;
    for (int c = 0; c + 1 < T; ++c) th.emplace_back([=, &f] { f(c, std::min(n, c * per), std::min(n, (c + 1) * per)); });
    f(T - 1, std::min(n, (T - 1) * per), n);
    for (std::thread& t : th) t.join();
    return T;
}

struct Laps {      // SDVGN_DROPIN_TIMING=1: where a call's host time goes, one line per call on stderr
    bool on = getenv("SDVGN_DROPIN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::string line;
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.0f |", what, std::chrono::duration<double, std::micro>(n - t).count());
        line += b; t = n;
    }
    void print(const char* head) { if (on) fprintf(stderr, "[dropin] %s:%s us\n", head, line.c_str()); }
};

void pose7(const SE3& T, double* o) {     // Sophus data(): [qx qy qz qw], then the translation
    const double* q = T.so3().data();
    for (int k = 0; k < 4; ++k) o[k] = q[k];
    for (int k = 0; k < 3; ++k) o[4 + k] = T.translation()[k];
}

}  // namespace

namespace sdvgn_dropin {

GpuWindow& window_for(const FullSystem* fs, int nP) {
    std::lock_guard<std::mutex> lk(g_mu);
    GpuWindow& g = g_windows[fs];
    const int w = wG[0], hgt = hG[0];
    if (g.h && (nP > g.max_points || w != g.w || hgt != g.hgt)) {       // a larger table: start over (everything is re-sent once)
        sdvgn_ef_destroy(g.h);
        const unsigned long long calls = g.calls;
        g = GpuWindow(); g.calls = calls;
    }
    if (!g.h) {
        g.max_points = nP > 12000 ? 2 * nP : 24000;
        g.w = w; g.hgt = hgt;
        GPU_CK(sdvgn_ef_create(&g.h, /*device*/ 0, w, hgt, g.max_points, /*stream*/ nullptr));
    }
    return g;
}

// EnergyFunctional::allPoints (private) in the order makeIDX builds it (EnergyFunctional.cpp:768-771), from the public frame / point lists
std::vector<EFPoint*> all_points(const EnergyFunctional* ef) {
    std::vector<EFPoint*> v;
    v.reserve(ef->nPoints);
    for (EFFrame* f : ef->frames) for (EFPoint* p : f->points) v.push_back(p);
    return v;
}

// step 1 of the header comment: the graph as it is now against what the device holds
// (the reference's objects are heap nodes scattered over memory: both walks below are chains of cache misses -- ~100 ns per residual object -- unless the
// objects a few points ahead are requested early; pid_out[k] = library id of allPoints[k])
void sync_window(GpuWindow& g, EnergyFunctional* ef, const std::vector<EFPoint*>& allPoints, CalibHessian& Hcalib, std::vector<int>& pid_out) {
    const int nF = ef->nFrames;
    ++g.epoch;
    Laps L;
    // ---- points: seen -> known or new; not seen -> removed (removePoint :597-620) ----
    std::vector<int> new_host; std::vector<float> nu, nv, nid, nidz, ncol, nwt; std::vector<unsigned char> nprior, nsens;
    std::vector<const EFPoint*> new_pts;
    const size_t nAll = allPoints.size();
    pid_out.assign(nAll, -1);
    // known points (the map and the mirrors are only read / marked here: distinct points, distinct mirrors)
    for_chunks(nAll, [&](int, size_t k0, size_t k1) {
        for (size_t k_ = k0; k_ < k1; ++k_) {
            const EFPoint* p = allPoints[k_];
            if (k_ + 8 < k1) __builtin_prefetch(allPoints[k_ + 8]);
            if (k_ + 4 < k1) __builtin_prefetch(allPoints[k_ + 4]->data);
            auto it = g.id_of.find(p);
            if (it == g.id_of.end()) continue;
            PointMirror& m = g.pts[it->second];
            if (m.ph == p->data && m.u == p->data->u && m.v == p->data->v && m.host_uid == p->host->data->shell->id) { m.seen = g.epoch; pid_out[k_] = it->second; }
        }
    });
    L.lap("known points");
    std::vector<size_t> new_k;
    for (size_t k_ = 0; k_ < nAll; ++k_) {
        if (pid_out[k_] >= 0) continue;
        const EFPoint* p = allPoints[k_];
        auto it = g.id_of.find(p);
        if (it != g.id_of.end()) g.id_of.erase(it);        // the address of a deleted EFPoint, re-used: the old id is removed below, this is a new point
        const PointHessian* ph = p->data;
        new_pts.push_back(p); new_k.push_back(k_);
        new_host.push_back(p->host->idx);
        nu.push_back(ph->u); nv.push_back(ph->v); nid.push_back(ph->idepth); nidz.push_back(ph->idepth_zero);
        for (int k = 0; k < 8; ++k) { ncol.push_back(ph->color[k]); nwt.push_back(ph->weights[k]); }
        nprior.push_back(ph->hasDepthPrior ? 1 : 0); nsens.push_back(ph->isFromSensor ? 1 : 0);
    }
    {
        std::vector<int> gone;
        for (size_t id = 0; id < g.pts.size(); ++id) {
            PointMirror& m = g.pts[id];
            if (!m.p || m.seen == g.epoch) continue;
            gone.push_back((int)id);
            auto it = g.id_of.find(m.p);
            if (it != g.id_of.end() && it->second == (int)id) g.id_of.erase(it);
            m = PointMirror();
        }
        if (!gone.empty()) GPU_CK(sdvgn_ef_remove_points(g.h, (int)gone.size(), gone.data()));
        g.points_removed += gone.size();
    }
    L.lap("new + gone points");
    // ---- frames that left (marginalizeFrame, EnergyFunctional.cpp:434-512) ----
    for (int t = (int)g.frames.size() - 1; t >= 0; --t) {
        bool alive = false;
        for (const EFFrame* f : ef->frames) if (f->data == g.frames[t] && f->data->shell->id == g.frame_uid[t]) alive = true;
        if (alive) continue;
        const int n1 = CPARS + 6 * ((int)g.frames.size() - 1);
        std::vector<double> Z((size_t)n1 * n1, 0.0), z(n1, 0.0);           // (the prior is re-sent below: marginalizeFrame ran on the host)
        GPU_CK(sdvgn_ef_remove_frame(g.h, t, Z.data(), z.data()));
        g.col_uid[g.frame_col[t]] = -1;
        g.frames.erase(g.frames.begin() + t); g.frame_uid.erase(g.frame_uid.begin() + t); g.frame_col.erase(g.frame_col.begin() + t);
    }
    // ---- frames that arrived (insertFrame :352-398): appended, like the reference appends them ----
    for (int t = 0; t < nF; ++t) {
        const FrameHessian* fh = ef->frames[t]->data;
        if (t < (int)g.frames.size()) {
            if (g.frames[t] != fh) fs_die("frame order of the window differs from the device's", -1);
            continue;
        }
        double ev[7], st[10], sz[10];
        pose7(fh->worldToCam_evalPT, ev);
        for (int k = 0; k < 10; ++k) { st[k] = fh->state[k]; sz[k] = fh->state_zero[k]; }
        GPU_CK(sdvgn_ef_insert_frame(g.h, ev, st, sz, ef->frames[t]->frameID, fh->ab_exposure, fh->frameEnergyTH, (const float*)fh->dI, nullptr));
        int col = 0;
        while (col < SDVGN_MAX_FRAMES && g.col_uid[col] >= 0) ++col;
        if (col == SDVGN_MAX_FRAMES) fs_die("more than SDVGN_MAX_FRAMES key-frames", -1);
        g.col_uid[col] = fh->shell->id;
        g.frames.push_back(fh); g.frame_uid.push_back(fh->shell->id); g.frame_col.push_back(col);
        ++g.frames_uploaded;
    }
    if (!new_pts.empty()) {
        std::vector<int> ids(new_pts.size());
        GPU_CK(sdvgn_ef_insert_points(g.h, (int)new_pts.size(), new_host.data(), nu.data(), nv.data(), nid.data(), nidz.data(), ncol.data(), nwt.data(),
                                      nprior.data(), nsens.data(), ids.data()));
        for (size_t k = 0; k < new_pts.size(); ++k) {
            if ((int)g.pts.size() <= ids[k]) g.pts.resize(ids[k] + 1);
            PointMirror& m = g.pts[ids[k]];
            m = PointMirror();
            m.p = new_pts[k]; m.ph = new_pts[k]->data; m.seen = g.epoch;
            m.u = m.ph->u; m.v = m.ph->v; m.host_uid = new_pts[k]->host->data->shell->id;
            g.id_of[new_pts[k]] = ids[k];
            pid_out[new_k[k]] = ids[k];
        }
        g.points_inserted += new_pts.size();
    }
    L.lap("frames + insert_points");
    // ---- residuals: inserted (insertResidual :400-412), matcher set since (findMatches, FullSystem.cpp:1121-1133), dropped (dropResidual :578-595) ----
    struct ResEdits { std::vector<int> ins_id, ins_t, ins_st, upd_id, upd_t, upd_st, drp_id, drp_t; std::vector<unsigned char> ins_hm, upd_hm; std::vector<double> ins_m, upd_m; };
    std::vector<ResEdits> part(64);
    int col_of_t[SDVGN_MAX_FRAMES];
    for (int t = 0; t < nF; ++t) col_of_t[t] = g.frame_col[t];
    const int n_part = for_chunks(nAll, [&](int c, size_t k0, size_t k1) {
        ResEdits& E = part[c];
        for (size_t k_ = k0; k_ < k1; ++k_) {
            const EFPoint* p = allPoints[k_];
            if (k_ + 6 < k1) { const auto& v_ = allPoints[k_ + 6]->residualsAll; if (!v_.empty()) __builtin_prefetch(v_.data()); }
            if (k_ + 3 < k1) for (const EFResidual* r_ : allPoints[k_ + 3]->residualsAll) __builtin_prefetch(r_);
            if (k_ + 1 < k1) for (const EFResidual* r_ : allPoints[k_ + 1]->residualsAll) __builtin_prefetch(r_->data);
            const int id = pid_out[k_];
            PointMirror& m = g.pts[id];
            int seen_here = 0;
            for (const EFResidual* r : p->residualsAll) {
                const int t = r->targetIDX, col = col_of_t[t];
                ResMirror& rm = m.r[col];
                const PointFrameResidual* pr = r->data;
                const unsigned char hm = pr->hasMatcher ? 1 : 0;
                const float mx = (float)pr->matcher[0], my = (float)pr->matcher[1];
                if (rm.uid != g.frame_uid[t]) {
                    E.ins_id.push_back(id); E.ins_t.push_back(t); E.ins_st.push_back((int)pr->state_state); E.ins_hm.push_back(hm);
                    E.ins_m.push_back(pr->matcher[0]); E.ins_m.push_back(pr->matcher[1]);
                    rm.uid = g.frame_uid[t]; ++m.n_res;
                } else if (rm.hasMatcher != hm || (hm && (rm.mx != mx || rm.my != my))) {
                    E.upd_id.push_back(id); E.upd_t.push_back(t); E.upd_st.push_back((int)pr->state_state); E.upd_hm.push_back(hm);
                    E.upd_m.push_back(pr->matcher[0]); E.upd_m.push_back(pr->matcher[1]);
                }
                rm.hasMatcher = hm; rm.mx = mx; rm.my = my; rm.seen = g.epoch;
                ++seen_here;
            }
            if (seen_here != m.n_res) {                 // some of this point's device residuals are gone on the host
                for (int t = 0; t < nF; ++t) {
                    ResMirror& rm = m.r[col_of_t[t]];
                    if (rm.uid == g.frame_uid[t] && rm.seen != g.epoch) { E.drp_id.push_back(id); E.drp_t.push_back(t); rm = ResMirror(); --m.n_res; }
                }
                // (columns of frames that left were dropped with their frame)
                int n = 0;
                for (int t = 0; t < nF; ++t) if (m.r[col_of_t[t]].uid == g.frame_uid[t]) ++n;
                m.n_res = n;
            }
        }
    });
    L.lap("residual diff");
    // the chunks' edits in the order of the points (= what one thread would have recorded)
    ResEdits& A = part[0];
    for (int c = 1; c < n_part; ++c) {
        ResEdits& E = part[c];
        auto cat = [](auto& d, const auto& s_) { d.insert(d.end(), s_.begin(), s_.end()); };
        cat(A.ins_id, E.ins_id); cat(A.ins_t, E.ins_t); cat(A.ins_st, E.ins_st); cat(A.ins_hm, E.ins_hm); cat(A.ins_m, E.ins_m);
        cat(A.upd_id, E.upd_id); cat(A.upd_t, E.upd_t); cat(A.upd_st, E.upd_st); cat(A.upd_hm, E.upd_hm); cat(A.upd_m, E.upd_m);
        cat(A.drp_id, E.drp_id); cat(A.drp_t, E.drp_t);
    }
    std::vector<int>&ins_id = A.ins_id, &ins_t = A.ins_t, &ins_st = A.ins_st, &upd_id = A.upd_id, &upd_t = A.upd_t, &upd_st = A.upd_st, &drp_id = A.drp_id, &drp_t = A.drp_t;
    std::vector<unsigned char>&ins_hm = A.ins_hm, &upd_hm = A.upd_hm;
    std::vector<double>&ins_m = A.ins_m, &upd_m = A.upd_m;
    if (!drp_id.empty()) GPU_CK(sdvgn_ef_drop_residuals(g.h, (int)drp_id.size(), drp_id.data(), drp_t.data()));
    if (!ins_id.empty()) GPU_CK(sdvgn_ef_insert_residuals(g.h, (int)ins_id.size(), ins_id.data(), ins_t.data(), ins_st.data(), ins_hm.data(), ins_m.data()));
    if (!upd_id.empty()) GPU_CK(sdvgn_ef_update_residuals(g.h, (int)upd_id.size(), upd_id.data(), upd_t.data(), upd_st.data(), upd_hm.data(), upd_m.data()));
    g.res_inserted += ins_id.size(); g.res_dropped += drp_id.size(); g.res_updated += upd_id.size();
    L.lap("residual edit calls");
    // ---- step 2: what the host loop owns and may have moved since the last call ----
    double vs[4], vmz[4];
    for (int k = 0; k < 4; ++k) { vs[k] = Hcalib.value_scaled[k]; vmz[k] = Hcalib.value_minus_value_zero[k]; }
    GPU_CK(sdvgn_ef_set_calib(g.h, vs, vmz));
    std::vector<double> ev(7 * nF), st(10 * nF), sz(10 * nF);
    std::vector<float> abx(nF);
    for (int t = 0; t < nF; ++t) {
        const FrameHessian* fh = ef->frames[t]->data;
        pose7(fh->worldToCam_evalPT, &ev[7 * t]);
        for (int k = 0; k < 10; ++k) { st[10 * t + k] = fh->state[k]; sz[10 * t + k] = fh->state_zero[k]; }
        abx[t] = fh->ab_exposure;
    }
    GPU_CK(sdvgn_ef_update_frames(g.h, nF, ev.data(), st.data(), sz.data(), abx.data()));
    L.lap("calib + frames");
    GPU_CK(sdvgn_ef_make_idx(g.h));
    L.lap("make_idx");
    const int n = CPARS + 6 * nF;
    std::vector<double> HMr((size_t)n * n), bMr(n);
    for (int r = 0; r < n; ++r) { bMr[r] = ef->bM[r]; for (int c = 0; c < n; ++c) HMr[(size_t)r * n + c] = ef->HM(r, c); }
    GPU_CK(sdvgn_ef_set_marg_prior(g.h, HMr.data(), bMr.data()));
    GPU_CK(sdvgn_ef_compute_nullspaces(g.h));             // FullSystem::getNullspaces (:548-588): a function of the linearisation points
    GPU_CK(sdvgn_ef_set_adjoints(g.h));
    GPU_CK(sdvgn_ef_set_precalc(g.h));
    L.lap("prior + nullspaces + adjoints + precalc");
    L.print("sync");
    ++g.syncs;
}

}  // namespace sdvgn_dropin

// test / bench hooks (C linkage)
extern "C" unsigned long long sdvgn_dropin_opt_calls(const void* fs) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_windows.find((const FullSystem*)fs);
    return it == g_windows.end() ? 0 : it->second.calls;
}
// [0] key-frame images uploaded, [1] points inserted, [2] points removed, [3] residuals inserted, [4] dropped, [5] updated (all since creation);
// [6..8] microseconds of the last call: graph walk + edits + commit | optimize + finish on the device | write-back into the reference's objects
extern "C" void sdvgn_dropin_opt_stats(const void* fs, double out9[9]) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_windows.find((const FullSystem*)fs);
    for (int i = 0; i < 9; ++i) out9[i] = 0;
    if (it == g_windows.end()) return;
    const GpuWindow& g = it->second;
    out9[0] = (double)g.frames_uploaded; out9[1] = (double)g.points_inserted; out9[2] = (double)g.points_removed; out9[3] = (double)g.res_inserted;
    out9[4] = (double)g.res_dropped; out9[5] = (double)g.res_updated; out9[6] = g.us_sync; out9[7] = g.us_gpu; out9[8] = g.us_writeback;
}
extern "C" void sdvgn_dropin_opt_release(const void* fs) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_windows.find((const FullSystem*)fs);
    if (it == g_windows.end()) return;
    if (it->second.h) sdvgn_ef_destroy(it->second.h);
    g_windows.erase(it);
}

namespace sdv_loam {

float FullSystem::optimize(int mnumOptIts) {
    if (frameHessians.size() < 2) return 0;                                   // FullSystemOptimize.cpp:347-349
    if (frameHessians.size() < 3) mnumOptIts = 100;
    if (frameHessians.size() < 4) mnumOptIts = 75;
    Laps LP;
    const std::vector<EFPoint*> points = all_points(ef);
    LP.lap("all_points");
    const int nF = ef->nFrames, n = CPARS + 6 * nF, nP = (int)points.size();
    const auto t_0 = std::chrono::steady_clock::now();
    GpuWindow& g = window_for(this, nP);
    ++g.calls;
    std::vector<int> pid;
    sync_window(g, ef, points, Hcalib, pid);
    g.us_sync = us_since(t_0);

    // ---- the loop (:353-458) and the tail's linearizeAll(true) (:460-470) on the device ----
    const auto t_1 = std::chrono::steady_clock::now();
    const int cap = mnumOptIts > 0 ? mnumOptIts : 1, stride = 7 + n + 1;
    std::vector<double> trace((size_t)cap * stride, 0.0);
    // `canbreak && iteration >= setting_minOptIterations` (:456): the library's loop has the reference's default (1) built in and an all-bodies mode; any other
    // value would silently run a different number of bodies than the reference -- refuse it
    if (setting_minOptIterations != 1 && setting_minOptIterations < mnumOptIts)
        fs_die("setting_minOptIterations other than 1 (or >= the iteration count) is not supported by this binding", setting_minOptIterations);
    const int fixed = setting_minOptIterations >= mnumOptIts ? 1 : 0;
    const int its = sdvgn_ef_optimize(g.h, mnumOptIts, fixed, trace.data(), stride, cap);
    if (its < 0) fs_die("sdvgn_ef_optimize", its);
    if (!setting_debugout_runquiet) {     // the reference's console lines (:414-425), from the device's trace
        printf("OPTIMIZE %d pts (GPU window: %llu key-frame images uploaded so far)!\n", ef->nPoints, g.frames_uploaded);
        for (int i = 0; i < its; ++i) {
            const double* tr = &trace[(size_t)i * stride];
            // (printOptRes' per-iteration extras -- the average residual, the counts, the newest frame's a / b -- are not traced: n/a)
            printf("%s %d (L %.2f, dir n/a, ss 1.0): \tA(%f)=(AV n/a). Num: A(n/a) + M(%'d); ab n/a!\n", tr[2] != 0 ? "ACCEPT" : "REJECT", i, log10(tr[1]), tr[3], ef->resInM);
        }
    }
    // the loop's final state -> the reference's objects, by the reference's own setters (the tail below starts from it)
    std::vector<double> st(10 * (size_t)nF);
    std::vector<float> idp(nP);
    double vs[4];
    GPU_CK(sdvgn_ef_get_state(g.h, vs, st.data(), idp.data()));
    {
        VecC v; v << vs[0], vs[1], vs[2], vs[3];
        Hcalib.setValueScaled(v);                                             // (also value, value_minus_value_zero and the float views)
    }
    for (int t = 0; t < nF; ++t) {
        FrameHessian* fh = ef->frames[t]->data;
        Vec10 s; for (int k = 0; k < 10; ++k) s[k] = st[10 * (size_t)t + k];
        fh->setState(s);
    }
    double lastE = 0;
    std::vector<float> relbs(nP);
    std::vector<int> ngood(nP);
    std::vector<unsigned char> removed((size_t)nF * nP);
    GPU_CK(sdvgn_ef_optimize_finish(g.h, &lastE, relbs.data(), ngood.data(), removed.data()));
    std::vector<unsigned char> nogood(nP);
    GPU_CK(sdvgn_ef_get_point_nogood(g.h, nogood.data()));
    g.us_gpu = us_since(t_1);

    // ---- step 4: what FullSystem::optimize leaves behind ----
    const auto t_2 = std::chrono::steady_clock::now();
    Laps LW;
    {   // the tail on the reference's objects, with the reference's own members (:460-466)
        Vec10 newStateZero = Vec10::Zero();
        newStateZero.segment<2>(6) = frameHessians.back()->get_state().segment<2>(6);
        frameHessians.back()->setEvalPT(frameHessians.back()->PRE_worldToCam, newStateZero);
        EFDeltaValid = false;
        EFAdjointsValid = false;
        ef->setAdjointsF(&Hcalib);
        setPrecalcValues();
    }
    LW.lap("setEvalPT + setAdjointsF + setPrecalcValues");
    std::vector<int> ids(nP);
    GPU_CK(sdvgn_ef_get_point_ids(g.h, ids.data()));
    std::vector<int> idx_of_id(g.pts.size(), -1);
    for (int i = 0; i < nP; ++i) idx_of_id[ids[i]] = i;
    std::vector<float> pts9(9 * (size_t)nP);
    GPU_CK(sdvgn_ef_get_points(g.h, pts9.data()));
    const size_t slots = (size_t)nF * nP;
    std::vector<unsigned char> ex(slots), act(slots);
    std::vector<signed char> sst(slots), snew(slots);
    std::vector<float> en(slots), enn(slots), ewo(slots);
    GPU_CK(sdvgn_ef_get_residual_table(g.h, ex.data(), sst.data(), snew.data(), en.data(), enn.data(), ewo.data(), act.data()));
    {
        std::vector<float> th(nF);
        GPU_CK(sdvgn_ef_get_frame_energy_th(g.h, th.data()));
        frameHessians.back()->frameEnergyTH = th[nF - 1];                      // setNewFrameEnergyTH moves the newest frame's only (:63-97)
    }
    for (int t = 0; t < nF; ++t) {     // FrameHessian::step of the last solveSystemF (EnergyFunctional.cpp:748-752): -x
        FrameHessian* fh = ef->frames[t]->data;
        if (its > 0) { const double* x = &trace[(size_t)(its - 1) * stride + 7]; for (int k = 0; k < 6; ++k) fh->step[k] = -x[CPARS + 6 * t + k]; fh->step.tail<4>().setZero(); }
    }
    if (its > 0) {
        const double* x = &trace[(size_t)(its - 1) * stride + 7];
        ef->lastX = VecX(n);
        for (int i = 0; i < n; ++i) ef->lastX[i] = x[i];
        Hcalib.step = -ef->lastX.head<CPARS>();
    }
    {
        int resInA = 0;
        GPU_CK(sdvgn_ef_get_top_acc(g.h, nullptr, &resInA));
        ef->resInA = resInA;
    }
    LW.lap("getters");
    // per point: independent of every other point except for the toRemove surgery, which is collected and done below in the points' order
    std::vector<std::vector<size_t>> with_removed(64);
    const int n_part = for_chunks(points.size(), [&](int c, size_t k0, size_t k1) {
        for (size_t k_ = k0; k_ < k1; ++k_) {
            EFPoint* p = points[k_];
            if (k_ + 8 < k1) __builtin_prefetch(points[k_ + 8]);
            if (k_ + 6 < k1) __builtin_prefetch(points[k_ + 6]->data);
            if (k_ + 4 < k1) { const auto& v_ = points[k_ + 4]->data->residuals; if (!v_.empty()) __builtin_prefetch(v_.data()); }
            if (k_ + 2 < k1) for (PointFrameResidual* r_ : points[k_ + 2]->data->residuals) __builtin_prefetch(r_);
            if (k_ + 1 < k1) for (PointFrameResidual* r_ : points[k_ + 1]->data->residuals) __builtin_prefetch(r_->efResidual);
            const int id = pid[k_], i = idx_of_id[id];
            PointHessian* ph = p->data;
            const float* o = &pts9[9 * (size_t)i];
            ph->setIdepth(idp[i]);
            ph->setIdepthZero(idp[i]);                                             // doStepFromBackup / loadSateBackup keep both equal (:218,:279)
            ph->step = o[8];
            p->Hdd_accAF = o[0]; p->bd_accAF = o[1];
            for (int k = 0; k < 4; ++k) p->Hcd_accAF[k] = o[2 + k];
            p->HdiF = o[6]; p->bdSumF = o[7];
            // AccumulatedSCHessianSSE::addPoint of the last solveSystemF (AccumulatedSCHessian.cpp:12-28)
            if (nogood[i]) ph->maxRelBaseline = 0;                                 // (some solve of the loop met the point without an active residual: :14-21)
            if (o[6] == 0.0f) { ph->idepth_hessian = 0; ph->maxRelBaseline = 0; }
            else { float H = p->Hdd_accAF + p->Hdd_accLF + p->priorF; if (H < 1e-10) H = 1e-10; ph->idepth_hessian = H; }
            // linearizeAll(true), the isNew bookkeeping (FullSystemOptimize.cpp:34-47)
            if (relbs[i] > ph->maxRelBaseline) ph->maxRelBaseline = relbs[i];
            ph->numGoodResiduals += ngood[i];
            bool any_removed = false;
            for (size_t k = 0; k < ph->residuals.size(); ++k) {
                PointFrameResidual* r = ph->residuals[k];
                const int t = r->efResidual->targetIDX;
                const size_t s = (size_t)t * nP + i;
                if (!ex[s] && !removed[s]) fs_die("a residual of the host graph does not exist on the device", -1);
                r->state_NewEnergyWithOutlier = ewo[s];
                r->state_NewState = (ResState)snew[s];
                r->state_NewEnergy = enn[s];
                r->setState((ResState)sst[s]);
                r->state_energy = en[s];
                r->efResidual->isActiveAndIsGoodNEW = act[s] != 0;
                {   // centerProjectedTo as linearize leaves it (Residuals.cpp:90-97), by the reference's own projection
                    FrameFramePrecalc* pc = &(r->host->targetPrecalc[r->target->idx]);
                    float drescale, u, v, Ku, Kv, new_idepth; Vec3f KliP;
                    if (r->hasMatcher && projectPoint(ph->u, ph->v, ph->idepth_zero_scaled, 0, 0, &Hcalib, pc->PRE_RTll_0, pc->PRE_tTll_0, drescale, u, v, Ku, Kv, KliP, new_idepth))
                        r->centerProjectedTo = Vec3f(Ku, Kv, new_idepth);
                }
                if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].second = r->state_state;          // :128-134
                else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].second = r->state_state;
                any_removed = any_removed || removed[s];
            }
            if (any_removed) with_removed[c].push_back(k_);
        }
    });
    LW.lap("points + residuals");
    for (int c = 0; c < n_part; ++c)
        for (size_t k_ : with_removed[c]) {                                         // the toRemove list (:136-155)
            PointHessian* ph = points[k_]->data;
            const int id = pid[k_], i = idx_of_id[id];
            PointMirror& m = g.pts[id];
            for (size_t k = 0; k < ph->residuals.size(); ++k) {
                PointFrameResidual* r = ph->residuals[k];
                const int t = r->efResidual->targetIDX;
                if (!removed[(size_t)t * nP + i]) continue;
                if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = 0;
                else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = 0;
                ef->dropResidual(r->efResidual);
                deleteOut<PointFrameResidual>(ph->residuals, (int)k);
                --k;
                ResMirror& rm = m.r[g.frame_col[t]];
                if (rm.uid == g.frame_uid[t]) { rm = ResMirror(); --m.n_res; }
            }
        }
    LW.lap("toRemove");
    activeResiduals.clear();     // (the reference leaves pointers to the residuals it has just deleted in here; nothing reads the vector before the next optimize rebuilds it)
    Vec3 lastEnergy(lastE, 0, 0);
    if (!std::isfinite((double)lastEnergy[0])) { printf("KF Tracking failed: LOST!\n"); isLost = true; }      // :472-476
    statistics_lastFineTrackRMSE = sqrtf((float)(lastEnergy[0] / ef->resInA));
    if (calibLog != 0) {
        (*calibLog) << Hcalib.value_scaled.transpose() << " " << frameHessians.back()->get_state_scaled().transpose() << " " << sqrtf((float)(lastEnergy[0] / ef->resInA))
                    << " " << ef->resInM << "\n";
        calibLog->flush();
    }
    {
        boost::unique_lock<boost::mutex> crlock(shellPoseMutex);
        for (FrameHessian* fh : frameHessians) {
            fh->shell->camToWorld = fh->PRE_camToWorld;
            fh->shell->aff_g2l = fh->aff_g2l();
        }
    }
    LW.lap("rest");
    LW.print("write-back");
    g.us_writeback = us_since(t_2);
    return sqrtf((float)(lastEnergy[0] / (ef->resInA)));
}

}  // namespace sdv_loam
