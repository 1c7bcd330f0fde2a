// oracle/dropin/dropin_shared.hpp -- TEST INFRASTRUCTURE: what the reference-side bindings of oracle/dropin/*.cpp share -- the GPU window that stays on
// the device for a FullSystem (defined in FullSystemOptimizeGPU.cpp) and the GPU tracker of a CoarseTracker (CoarseTrackerGPU.cpp).  In a real integration
// these are members (`GpuWindow gpu` of FullSystem / EnergyFunctional, `sdvgn_tracker* gpu` of CoarseTracker); here they sit in registries keyed by the
// object, because the reference's headers are compiled unmodified.
#pragma once
#include "FullSystem/FullSystem.h"
#include "FullSystem/CoarseTracker.h"
#include "FullSystem/HessianBlocks.h"
#include "OptimizationBackend/EnergyFunctional.h"
#include "OptimizationBackend/EnergyFunctionalStructs.h"

extern "C" {
#include "sdvgn.h"
}

#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

namespace sdvgn_dropin {

using namespace sdv_loam;

[[noreturn]] inline void die(const char* who, const char* what, int rc) {
    fprintf(stderr, "%s: %s failed: %s (%d)\n", who, what, sdvgn_error_string(rc), rc);
    abort();
}

struct ResMirror { int uid = -1; unsigned char hasMatcher = 0; float mx = 0, my = 0; int seen = 0; };   // the device's residual point -> frame column k
struct PointMirror {
    const EFPoint* p = nullptr; const PointHessian* ph = nullptr;
    float u = 0, v = 0; int host_uid = -1;      // (a deleted point's addresses may be handed out again: the identity is the pair of objects AND what they describe)
    int seen = 0, n_res = 0;
    ResMirror r[SDVGN_MAX_FRAMES];
};
struct GpuWindow {
    sdvgn_ef* h = nullptr;
    int w = 0, hgt = 0, max_points = 0;
    std::vector<const FrameHessian*> frames;        // device frame order
    std::vector<int> frame_uid, frame_col;          // FrameShell::id, mirror column of every device frame
    int col_uid[SDVGN_MAX_FRAMES];                  // uid of the frame that owns mirror column k (-1: free)
    std::unordered_map<const EFPoint*, int> id_of;  // EFPoint -> library point id
    std::vector<PointMirror> pts;                   // by id
    int epoch = 0;
    unsigned long long calls = 0, frames_uploaded = 0, points_inserted = 0, points_removed = 0, res_inserted = 0, res_dropped = 0, res_updated = 0;
    unsigned long long immature_calls = 0, immature_points = 0, syncs = 0;
    double us_sync = 0, us_gpu = 0, us_writeback = 0;   // of the last optimize call
    GpuWindow() { for (int& c : col_uid) c = -1; }
};

// FullSystemOptimizeGPU.cpp
GpuWindow& window_for(const FullSystem* fs, int nP);
std::vector<EFPoint*> all_points(const EnergyFunctional* ef);
// the EnergyFunctional graph as it is now, diffed against what the device holds and committed (sdvgn_ef_make_idx); pid_out[k] = library id of allPoints[k]
void sync_window(GpuWindow& g, EnergyFunctional* ef, const std::vector<EFPoint*>& allPoints, CalibHessian& Hcalib, std::vector<int>& pid_out);

// CoarseTrackerGPU.cpp: the tracker handle of a CoarseTracker (created on first use), with `fh`'s pyramid as its new frame
sdvgn_tracker* tracker_handle(CoarseTracker* ct);
void tracker_set_new_frame(CoarseTracker* ct, FrameHessian* fh);
// a handle of ANY CoarseTracker whose current new frame is `fh` (NULL: none) -- the pyramid of a key-frame that was tracked as a frame is still on the device
sdvgn_tracker* tracker_holding(const FrameHessian* fh);
// the template on the device came from sdvgn_tracker_make_coarse_depth: trackNewestCoarse must not upload the host arrays over it
void tracker_mark_ref_on_device(CoarseTracker* ct);

}  // namespace sdvgn_dropin
