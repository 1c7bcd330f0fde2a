// oracle/ref_common.hpp -- TEST INFRASTRUCTURE, shared by the ref_glue_*.cpp entry points around the REFERENCE's own code (oracle/Makefile,
// target `ref`).  The reference keeps image size, intrinsics and the pyramid depth in process-wide globals (util/globalCalib.cpp,
// util/settings.cpp); every handle of the glue re-installs its own values on entry so that handles of different shapes can coexist.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "FullSystem/FullSystem.h"
#include "FullSystem/CoarseTracker.h"
#include "FullSystem/ImmaturePoint.h"
#include "OptimizationBackend/AccumulatedSCHessian.h"
#include "OptimizationBackend/AccumulatedTopHessian.h"
#include "OptimizationBackend/EnergyFunctional.h"
#include "OptimizationBackend/EnergyFunctionalStructs.h"
#include "util/globalCalib.h"

namespace refglue {
using namespace sdv_loam;

struct Globals {
    int w, h, levels;
    float fx, fy, cx, cy;
};

// util/globalCalib.cpp setGlobalCalib(w, h, K) decides the pyramid depth itself (halving while both sizes are even, :18-25), which gives one
// level for the odd KITTI widths; the synthetic configurations of SURVEY.md 8 name the level count (w >> l, h >> l) explicitly.  So: the
// reference's own function first, then the levels it did not fill, by the statements of its loop (:61-78).
inline void install(const Globals& g) {
    const bool quiet = true;
    int saved = -1;
    if (quiet) { fflush(stdout); saved = dup(1); FILE* f = fopen("/dev/null", "w"); dup2(fileno(f), 1); fclose(f); }
    Eigen::Matrix3f K = Eigen::Matrix3f::Zero();
    K(0, 0) = g.fx; K(1, 1) = g.fy; K(0, 2) = g.cx; K(1, 2) = g.cy; K(2, 2) = 1;
    setGlobalCalib(g.w, g.h, K);
    if (quiet) { fflush(stdout); dup2(saved, 1); close(saved); }
    for (int level = pyrLevelsUsed; level < g.levels; ++level) {
        wG[level] = g.w >> level;
        hG[level] = g.h >> level;
        fxG[level] = fxG[level - 1] * 0.5;
        fyG[level] = fyG[level - 1] * 0.5;
        cxG[level] = (cxG[0] + 0.5) / ((int)1 << level) - 0.5;
        cyG[level] = (cyG[0] + 0.5) / ((int)1 << level) - 0.5;
        KG[level] << fxG[level], 0.0, cxG[level], 0.0, fyG[level], cyG[level], 0.0, 0.0, 1.0;
        KiG[level] = KG[level].inverse();
        fxiG[level] = KiG[level](0, 0);
        fyiG[level] = KiG[level](1, 1);
        cxiG[level] = KiG[level](0, 2);
        cyiG[level] = KiG[level](1, 2);
    }
    pyrLevelsUsed = g.levels;
    // log files / console chatter of the reference's front end; no arithmetic depends on them
    setting_logStuff = false;
    setting_debugout_runquiet = true;
    setting_render_displayCoarseTrackingFull = false;
    multiThreading = false;   // the reference's default (settings.cpp:164)
}

// pose7 = Sophus data() layout [qx qy qz qw | tx ty tz]; the quaternion is stored as given (the SE3 constructors would re-normalise it)
inline SE3 pose_from7(const double* p) {
    SE3 T;
    double* q = T.so3().data();
    q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
    T.translation() = Vec3(p[4], p[5], p[6]);
    return T;
}
inline void pose_to7(const SE3& T, double* p) {
    const double* q = T.so3().data();
    p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; p[3] = q[3];
    p[4] = T.translation()[0]; p[5] = T.translation()[1]; p[6] = T.translation()[2];
}

// runs f with stdout captured into a string (the reference reports the optimiser's accept / reject decisions only through printf)
template <typename F> inline std::string capture_stdout(F f) {
    fflush(stdout);
    char path[] = "/tmp/refglue_XXXXXX";
    const int fd = mkstemp(path);
    const int saved = dup(1);
    dup2(fd, 1);
    f();
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    std::string out;
    lseek(fd, 0, SEEK_SET);
    char buf[4096];
    ssize_t n;
    while ((n = read(fd, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
    close(fd);
    unlink(path);
    return out;
}
}  // namespace refglue
