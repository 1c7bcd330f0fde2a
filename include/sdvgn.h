/* sdvgn.h -- C ABI of libsdvgn, the MI355X (gfx950) implementation of SDV-LOAM's Gauss-Newton hot path.
 *
 * The reference (ZikangYuan/SDV-LOAM, C++11/Eigen, CPU only) has no plugin / FFI layer: the hot path sits
 * behind a few C++ member functions.  Each entry point below names the reference
 * member it replaces (file:line relative to the reference root).  INTEGRATION.md shows the C++ shim
 * (`CoarseTrackerGPU`, `EnergyFunctionalGPU`) a maintainer adds so that FullSystem.cpp:419 and
 * FullSystemOptimize.cpp:512 stay untouched.
 *
 * Conventions
 *   - plain C types only; no torch / Eigen / Sophus types cross this boundary.
 *   - every function returns 0 on success, a negative hipError_t (as -(int)err) on a HIP failure, or one of
 *     the SDVGN_E_* codes; nothing throws.  A library built without a usable GPU fails at *_create.
 *   - `pose7` is exactly Sophus `SE3d::data()`: [qx qy qz qw tx ty tz] (Eigen quaternion coeff order).
 *   - one handle owns one HIP stream and is single-threaded, matching the reference's mutex discipline
 *     (coarseTracker under trackMutex, coarseTracker_forNewKF under mapMutex; FullSystem.h:277-327).
 *   - pointers are HOST pointers unless the parameter name ends in `_dev`.
 */
#ifndef SDVGN_H
#define SDVGN_H

#ifdef __cplusplus
extern "C" {
#endif

#define SDVGN_OK 0
#define SDVGN_E_ARG (-10001)      /* bad argument (level out of range, null pointer, n too large ...) */
#define SDVGN_E_STATE (-10002)    /* call order violated (e.g. track before set_ref / set_new_image) */
#define SDVGN_E_NODEVICE (-10003) /* no HIP device / wrong architecture */

#define SDVGN_MAX_LEVELS 6 /* PYR_LEVELS, src/util/settings.h */

const char* sdvgn_version(void);
const char* sdvgn_error_string(int code);

/* ===================================================================================================
 * Coarse tracker -- replaces class CoarseTracker (src/FullSystem/CoarseTracker.h:17-107)
 * =================================================================================================== */
typedef struct sdvgn_tracker sdvgn_tracker;

/* CoarseTracker::CoarseTracker(int w,int h)   CoarseTracker.cpp:34-69.
 * `levels` = pyrLevelsUsed (globalCalib.cpp:25-30); level l is (w0>>l) x (h0>>l) (CoarseTracker.cpp:89-90).
 * `max_points` bounds pc_n[lvl] (the reference allocates w*h per level); `max_batch` bounds the number of
 * pose hypotheses of sdvgn_tracker_track_batch.  `stream` is a hipStream_t or NULL (library creates one). */
int sdvgn_tracker_create(sdvgn_tracker** out, int device, int w0, int h0, int levels, int max_points, int max_batch,
                         void* stream);
/* CoarseTracker::~CoarseTracker   CoarseTracker.cpp:70-75 */
void sdvgn_tracker_destroy(sdvgn_tracker* t);

/* globals read by the tracker: setting_huberTH, setting_coarseCutoffTH, setting_affineOptModeA/B
 * (src/util/settings.cpp:93-94,101,112).  Defaults: 6, 20, 0, 0 (launch/run.launch mode=1). */
int sdvgn_tracker_set_settings(sdvgn_tracker* t, float huberTH, float coarseCutoffTH, float affineOptModeA,
                               float affineOptModeB);

/* CoarseTracker::makeK(CalibHessian*)   CoarseTracker.cpp:77-106  (level-0 fx,fy,cx,cy = HCalib->fxl()...) */
int sdvgn_tracker_make_K(sdvgn_tracker* t, float fx, float fy, float cx, float cy);
int sdvgn_tracker_get_K(sdvgn_tracker* t, int lvl, float fxfycxcy[4], float Ki9[9]);

/* Result of CoarseTracker::makeCoarseDepthL0 / makeCoarseDepthForFirstFrame (CoarseTracker.cpp:108-425):
 * the compacted reference template pc_u, pc_v, pc_idepth, pc_color [pc_n[lvl]] of one level
 * (CoarseTracker.h:87-91).  The splat/dilate itself stays on the host. */
int sdvgn_tracker_set_ref(sdvgn_tracker* t, int lvl, int n, const float* pc_u, const float* pc_v,
                          const float* pc_idepth, const float* pc_color);
/* lastRef->ab_exposure and lastRef_aff_g2l set in setCoarseTrackingRef / setCTRefForFirstFrame
 * (CoarseTracker.cpp:636-660) */
int sdvgn_tracker_set_ref_frame(sdvgn_tracker* t, float ab_exposure, double aff_a, double aff_b);

/* newFrame (trackNewestCoarse's FrameHessian*): its pyramid dIp[lvl] and ab_exposure.
 * _image: level-0 float image 0..255 (w0*h0); the {I,dx,dy} pyramid is built on the GPU -- replaces
 *         FrameHessian::makeImages (src/FullSystem/HessianBlocks.cpp:107-167) for the tracker's use.
 * _pyr:   take the reference's own AoS Eigen::Vector3f dIp[lvl] (w_l*h_l*3 floats) unchanged. */
int sdvgn_tracker_set_new_image(sdvgn_tracker* t, const float* image_lvl0, float ab_exposure);
int sdvgn_tracker_set_new_image_dev(sdvgn_tracker* t, const float* image_lvl0_dev, float ab_exposure);
int sdvgn_tracker_set_new_pyr(sdvgn_tracker* t, int lvl, const float* dIp_aos3, float ab_exposure);
int sdvgn_tracker_get_pyr(sdvgn_tracker* t, int lvl, float* dIp_aos3_out);

/* Vec6 CoarseTracker::calcRes(int lvl,const SE3&,AffLight,float cutoffTH)   CoarseTracker.cpp:486-634.
 * out6 = {E, numTermsInE, flowT, 0, flowRT, saturatedRatio}. */
int sdvgn_tracker_calc_res(sdvgn_tracker* t, int lvl, const double pose7[7], double aff_a, double aff_b,
                           float cutoffTH, double out6[6]);
/* void CoarseTracker::calcGSSSE(int lvl,Mat88&,Vec8&,const SE3&,AffLight)   CoarseTracker.cpp:427-484.
 * The reference reads the buf_warped_* planes left by the preceding calcRes at the same pose; the GPU
 * recomputes them in the same fused pass (cutoffTH must be the one used by that calcRes).
 * H88 row-major, scaled like CoarseTracker.cpp:472-483. */
int sdvgn_tracker_calc_gs(sdvgn_tracker* t, int lvl, const double pose7[7], double aff_a, double aff_b,
                          float cutoffTH, double H88[64], double b8[8]);
/* calcRes + calcGSSSE in one launch (one "Gauss-Newton iteration" of the tracker). */
int sdvgn_tracker_res_and_gs(sdvgn_tracker* t, int lvl, const double pose7[7], double aff_a, double aff_b,
                             float cutoffTH, double out6[6], double H88[64], double b8[8]);
/* Parity hook: per reference point i of the last calc_* call, the 8 floats the reference appends to
 * buf_warped_{idepth,u,v,dx,dy,residual,weight,refColor} (CoarseTracker.cpp:588-599), uncompacted, plus
 * status[i]: 0 = skipped (out of bounds / non-finite), 1 = in E and in the warped buffers, 2 = saturated.
 * terms is [8][n] plane-major. */
int sdvgn_tracker_get_point_terms(sdvgn_tracker* t, int lvl, float* terms, int* status);

/* bool CoarseTracker::trackNewestCoarse(FrameHessian*, SE3& lastToNew_out, AffLight& aff_g2l_out,
 *        int coarsestLvl, Vec5 minResForAbort, Output3DWrapper*)     CoarseTracker.cpp:662-838.
 * Returns 1 (true) / 0 (false) / <0 error.  lastResiduals[5] and lastFlowIndicators[3] are the side outputs
 * the caller reads (CoarseTracker.h:62-65, FullSystem.cpp:436-458). NaN in minResForAbort disables the abort
 * exactly as in the reference (`x > 1.5*NaN` is false). */
int sdvgn_tracker_track(sdvgn_tracker* t, double pose7_io[7], double aff_io[2], int coarsestLvl,
                        const double minResForAbort[5], double lastResiduals[5], double lastFlowIndicators[3]);
/* Same LM loop for B independent starting poses against the same reference/new frame -- the <= 31 hypotheses
 * FullSystem::trackNewCoarse tries one after the other (FullSystem.cpp:341-414).  One workgroup per
 * hypothesis, whole coarse-to-fine loop on the device.  ok[i] = 1/0 like the bool above. */
int sdvgn_tracker_track_batch(sdvgn_tracker* t, int B, double* pose7_io, double* aff_io, int coarsestLvl,
                              const double* minResForAbort, double* lastResiduals, double* lastFlowIndicators,
                              int* ok);
/* LM trial log of the last sdvgn_tracker_track call: rows of 15 doubles
 * {lvl, iteration, lambda, accept, incScaled[8], E_new, n_new, cutoffRepeat}; returns rows written. */
int sdvgn_tracker_get_trace(sdvgn_tracker* t, double* rows, int cap);

/* Throughput form of res_and_gs: B problems (poses) in ONE launch on level `lvl`; results stay on the device
 * (out_dev: B x 80 doubles = {out6[6], H88[64], b8[8], pad[2]}).  Used by bench.py for the roofline run. */
int sdvgn_tracker_res_and_gs_batch(sdvgn_tracker* t, int lvl, int B, const double* pose7, const double* aff,
                                   float cutoffTH, double* out_dev);
void* sdvgn_tracker_stream(sdvgn_tracker* t);

#ifdef __cplusplus
}
#endif
#endif /* SDVGN_H */
