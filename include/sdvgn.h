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
 *   - one handle works on one HIP stream and is single-threaded, matching the reference's mutex discipline
 *     (coarseTracker under trackMutex, coarseTracker_forNewKF under mapMutex; FullSystem.h:277-327).  `stream` = NULL at *_create:
 *     tracker handles share one library stream per device, window (ef) handles another -- a stream per handle would be a hardware
 *     queue per handle, and the first launch on a queue that has been idle costs 0.2-0.5 ms.  Pass streams to overlap two handles.
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
 * pose hypotheses of sdvgn_tracker_track_batch.  `stream` is a hipStream_t or NULL (the library's shared tracker stream of that device). */
int sdvgn_tracker_create(sdvgn_tracker** out, int device, int w0, int h0, int levels, int max_points, int max_batch,
                         void* stream);
/* CoarseTracker::~CoarseTracker   CoarseTracker.cpp:70-75 */
void sdvgn_tracker_destroy(sdvgn_tracker* t);

/* globals read by the tracker: setting_huberTH, setting_coarseCutoffTH, setting_affineOptModeA/B
 * (src/util/settings.cpp:93-94,101,112).  Defaults: 6, 20, 0, 0 (launch/run.launch mode=1). */
int sdvgn_tracker_set_settings(sdvgn_tracker* t, float huberTH, float coarseCutoffTH, float affineOptModeA,
                               float affineOptModeB);

/* Tolerance study of BASELINE.json configs[4] (not a reference feature): 0 = fp32 (default, the product path),
 * 1 = fp16 pyramid, 2 = + fp16 Jacobian/residual operands, 3 = + fp16 accumulation.  Affects calc_gs / res_and_gs /
 * track (host-driven) only; calc_res (parity hook) and track_batch always run in fp32.
 * 4 = fp32 on a GATHER-FRIENDLY device copy of the target pyramid: per pixel one 64-byte record with the {I,dx,dy} of its 2x2 neighbourhood,
 * so that the four taps of a lookup come from one 128-byte line instead of two image rows (the reference's row-major AoS float3,
 * HessianBlocks.cpp:135-155, stays the hand-over format of sdvgn_tracker_set_new_pyr).  Results are bit-identical to mode 0 -- only
 * addresses change; HBM traffic of independent tracking problems drops ~2.5x.  4.3x the device memory of a pyramid. */
int sdvgn_tracker_set_precision(sdvgn_tracker* t, int mode);
/* Device pointer of level lvl's record copy (mode 4), built on demand: the per-problem image pointer sdvgn_tracker_res_and_gs_multi takes
 * when the calling handle is in mode 4. */
const void* sdvgn_tracker_records_dev(sdvgn_tracker* t, int lvl);
/* Arithmetic of the fused calcRes + calcGSSSE kernel (k_res_gs: calc_gs / res_and_gs / res_and_gs_batch / res_and_gs_multi / track, the
 * host-driven paths): 0 = the reference's float arithmetic operation by operation (default; per-point terms bit-identical to the CPU path),
 * 1 = tolerance mode: fused multiply-adds and reciprocal-based divisions (v_rcp_f32 + one Newton step).  BASELINE.json's contract for this
 * path is 1e-4 relative on pose increments; mode 1 meets it (tests/test_tracker_gpu.py) with ~40 % fewer vector instructions per point. */
int sdvgn_tracker_set_arith(sdvgn_tracker* t, int mode);
/* Workgroups per pose hypothesis of sdvgn_tracker_track_batch: 0 = automatic (one pass of 256-lane workgroups over the largest level, at
 * most 32, as long as the whole launch is resident at once; otherwise one 1024-lane workgroup per hypothesis), -1 = always one workgroup
 * (k_track), 1..32 = that many (k_track_team).  Results do not depend on it beyond the summation order of the 52 totals.
 * sdvgn_tracker_get_team: what the last track_batch call used (0 = k_track).
 * The workgroups of a team wait for each other inside the kernel, so a team launch needs its whole grid resident.  The team size is limited to
 * three quarters of what the device can hold of the kernel at once (occupancy query x CUs), which leaves room for what runs beside it (the
 * back end on its own stream, other handles); if members still fail to meet -- the device was busier than that -- their poll gives up after
 * ~1 s, the batch is re-run on the one-workgroup kernel and the call succeeds (sdvgn_tracker_get_team then reports 0,
 * sdvgn_tracker_get_team_fallbacks counts such re-runs). */
int sdvgn_tracker_set_team(sdvgn_tracker* t, int team);
int sdvgn_tracker_get_team(sdvgn_tracker* t);
int sdvgn_tracker_get_team_fallbacks(sdvgn_tracker* t);

/* CoarseTracker::makeK(CalibHessian*)   CoarseTracker.cpp:77-106  (level-0 fx,fy,cx,cy = HCalib->fxl()...) */
int sdvgn_tracker_make_K(sdvgn_tracker* t, float fx, float fy, float cx, float cy);
int sdvgn_tracker_get_K(sdvgn_tracker* t, int lvl, float fxfycxcy[4], float Ki9[9]);

/* Result of CoarseTracker::makeCoarseDepthL0 / makeCoarseDepthForFirstFrame (CoarseTracker.cpp:108-425):
 * the compacted reference template pc_u, pc_v, pc_idepth, pc_color [pc_n[lvl]] of one level
 * (CoarseTracker.h:87-91), for callers that build it themselves; sdvgn_tracker_make_coarse_depth builds it on the device. */
int sdvgn_tracker_set_ref(sdvgn_tracker* t, int lvl, int n, const float* pc_u, const float* pc_v,
                          const float* pc_idepth, const float* pc_color);
/* CoarseTracker::makeCoarseDepthL0(frameHessians) (CoarseTracker.cpp:258-425) and makeCoarseDepthForFirstFrame (:108-256) -- the body
 * of setCoarseTrackingRef / setCTRefForFirstFrame (:636-660) -- on the device: builds pc_u / pc_v / pc_idepth / pc_color of every
 * level (what sdvgn_tracker_set_ref uploads otherwise) from the tuples the reference splats, in the reference's order:
 *   makeCoarseDepthL0:            per point either (int)u, (int)v, idepth (:268-277) or (int)(centerProjectedTo + 0.5f), centerProjectedTo[2]
 *                                 (:284-292), weight = sqrtf(1e-3 / (efPoint->HdiF + 1e-12))
 *   makeCoarseDepthForFirstFrame: (int)(u + 0.5f), (int)(v + 0.5f), idepth, same weight (:114-125)
 * ref_pyr_dev: lastRef->dIp[lvl] as device pointers (AoS {I,dx,dy}), e.g. another tracker handle's pyramid; NULL = this handle's
 * current new frame.  The tracker must have been created with max_points >= the number of template points (the reference
 * allocates w*h per level).  Raster order and every value are identical to the reference's loop; one documented deviation: the
 * dilation's out-of-range neighbour reads at the first / last processed pixel (undefined behaviour, :339-342) count as "no value". */
int sdvgn_tracker_make_coarse_depth(sdvgn_tracker* t, int n, const int* u, const int* v, const float* new_idepth, const float* weight,
                                    const float* const* ref_pyr_dev);
/* read back the reference template of a level (parity hook); returns pc_n[lvl]; with u == NULL only the count */
int sdvgn_tracker_get_ref(sdvgn_tracker* t, int lvl, float* u, float* v, float* idepth, float* color);

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
/* The same launch over B INDEPENDENT problems: problem b uses the reference template pc_dev[b] (sdvgn_tracker_ref_dev of any handle, level
 * `lvl`) and the level image img_dev[b] (sdvgn_tracker_pyr_dev of any handle) -- many frames / agents in one launch.  All problems must
 * have this handle's level geometry, intrinsics, exposure / affine reference and template point count. */
int sdvgn_tracker_res_and_gs_multi(sdvgn_tracker* t, int lvl, int B, const void* const* pc_dev, const void* const* img_dev,
                                   const double* pose7, const double* aff, float cutoffTH, double* out_dev);
/* device pointer of the packed reference template {u,v,idepth,colour} of level `lvl` (float4 per point) */
const void* sdvgn_tracker_ref_dev(sdvgn_tracker* t, int lvl);
void* sdvgn_tracker_stream(sdvgn_tracker* t);
/* Device pointer of level `lvl` of the current new-frame pyramid (dIp[lvl], AoS {I,dx,dy}, (w0>>lvl)*(h0>>lvl)*3 floats), NULL if none
 * was set.  For handing the image to the other handles without a copy: sdvgn_reproj_set_cur_level(.., dIp_aos3_dev),
 * sdvgn_tracker_make_coarse_depth(.., ref_pyr_dev).  Valid until the next set_new_image / set_new_pyr / destroy of this handle. */
const float* sdvgn_tracker_pyr_dev(sdvgn_tracker* t, int lvl);

/* CoarseTracker::structPoseEstimation(SE3& curToWorld, overlap_pts)   CoarseTracker.cpp:949-1004, called right after
 * trackNewestCoarse on every frame (FullSystem.cpp:483-489); with calculateRes (:840-872), calculateWeight (:874-889),
 * calcHandb (:891-947) and point2world / world2frame / pixel2unit (ResidualProjections.h:61-103).
 * overlap_pts is flattened: for match i the map point's host pixel u[i], v[i], its `idepth` (PointHessian::idepth, NOT
 * idepth_scaled), the index host_idx[i] of its host key-frame in host_pose7 (n_hosts x 7, `host->shell->camToWorld.data()`)
 * and the matched pixel obs[2i], obs[2i+1] (`it->second`).  Needs make_K (level-0 fx,fy,cx,cy,fxi,fyi; wM3G = w-3, hM3G = h-3).
 * n <= 4096.  curToWorld7 is updated in place by accepted steps only (the reference's SE3& out-parameter).
 * trace (optional, 10 x sdvgn_struct_trace_stride() doubles): per iteration [it, lambda, resOld, resNew, accept, inc(6), num,
 * extrapFac, |inc|, 0, 0]; final_res (optional) = the last accepted mean squared pixel error.
 * Returns the number of LM iterations run (0..10) or a negative error.  The reference's quirks are kept: in-place cumulative
 * damping (:959), H,b rebuilt at the pose BEFORE an accepted step (:983), no num==0 guard on the first energy (:951-952), the
 * 1 - u^2 / -(1 - v^2) Jacobian entries (:919,:925); the
 * reference function has no return statement (:1003) and its caller ignores the value. */
int sdvgn_tracker_struct_pose(sdvgn_tracker* t, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                              int n_hosts, const double* host_pose7, const double* obs, double* curToWorld7, double* trace,
                              double* final_res);
/* parity hook: one calcHandb (:891-947) + calculateRes (:840-872) at worldToCur7; H36 row-major, energy = sum of squared pixel
 * errors (not divided by num). */
int sdvgn_tracker_struct_res_hb(sdvgn_tracker* t, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                                int n_hosts, const double* host_pose7, const double* obs, const double* worldToCur7, double* H36,
                                double* b6, double* energy, int* num);
int sdvgn_struct_trace_stride(void);

/* ImmaturePoint::traceOn(frame, hostToFrame_KRKi, hostToFrame_Kt, hostToFrame_affine, HCalib)   ImmaturePoint.cpp:47-353,
 * for ALL immature points of all key-frames in one launch -- the loop of FullSystem::traceNewCoarse (FullSystem.cpp:519-553).
 * The frame traced on is the tracker's current new frame (set_new_image / set_new_pyr level 0 = fh->dI).
 * set_points registers the per-point data that does not change between frames (ImmaturePoint.h:33-75: u, v, energyTH, gradH as
 * row-major 2x2, color[8], weights[8], and the index of its host key-frame, < 16); call it when immature points are created or
 * deleted.  trace_points runs the search: per host h the caller passes exactly what traceNewCoarse computes -- KRKi9[h] = K *
 * hostToNew.rotationMatrix().cast<float>() * K.inverse() (row-major), Kt3[h] = K * hostToNew.translation().cast<float>(),
 * aff2[h] = AffLight::fromToVecExposure(host, new).cast<float>() -- and the per-point state arrays, updated in place:
 * idepth_min, idepth_max, quality, status (= lastTraceStatus, enum ImmaturePointStatus 0..5, ImmaturePoint.h:20-30),
 * lastTraceUV (2 per point), lastTracePixelInterval.  Points with status IPS_OOB are left untouched (:49). */
int sdvgn_tracker_trace_set_points(sdvgn_tracker* t, int n, const float* u, const float* v, const float* energyTH, const float* gradH4,
                                   const float* color8, const float* weights8, const int* host_idx);
int sdvgn_tracker_trace_points(sdvgn_tracker* t, int n_hosts, const float* KRKi9, const float* Kt3, const float* aff2, float* idepth_min,
                               float* idepth_max, float* quality, int* status, float* lastTraceUV2, float* lastTracePixelInterval);

/* ===================================================================================================
 * Reprojector -- the per-candidate work of class Reprojector (src/FullSystem/Reprojector.h:17-112); SURVEY.md 8f row 2
 * =================================================================================================== */
typedef struct sdvgn_reproj sdvgn_reproj;

/* Reprojector::Reprojector(CalibHessian*, FrameHessian* newframe, std::vector<FrameHessian*>&)   Reprojector.cpp:81-87.
 * The handle outlives one frame: key-frames are (re)registered with set_frame, the new frame with set_cur / set_cur_level.
 * max_frames <= 16 (index space of host_idx / ref_idx), max_points bounds n of sdvgn_reproj_match. */
int sdvgn_reproj_create(sdvgn_reproj** out, int device, int w0, int h0, int levels, int max_frames, int max_points, void* stream);
void sdvgn_reproj_destroy(sdvgn_reproj* r);
void* sdvgn_reproj_stream(sdvgn_reproj* r);
/* K_ = [fxl 0 cxl; 0 fyl cyl; 0 0 1] in double (:85-86); K_.inverse() is taken once, in Eigen's closed cofactor form */
int sdvgn_reproj_set_calib(sdvgn_reproj* r, float fx, float fy, float cx, float cy);
/* key-frame idx: shell->camToWorld.data(), its level-0 image `dI` (AoS {I,dx,dy}, w0*h0*3 floats) either as a host buffer
 * (copied) or as a device pointer that stays valid (e.g. the image the back-end handle already holds) -- pass exactly one, or
 * neither to keep the image registered earlier; ab_exposure and shell->aff_g2l (a,b) feed AffLight::fromToVecExposure (:253-255) */
int sdvgn_reproj_set_frame(sdvgn_reproj* r, int idx, const double* camToWorld7, const float* dI_aos3, const float* dI_aos3_dev,
                           float ab_exposure, double aff_a, double aff_b);
/* the new frame: pose, exposure, aff_g2l; and its pyramid dIp[lvl] (host copy or device pointer, e.g. sdvgn_tracker's) */
int sdvgn_reproj_set_cur(sdvgn_reproj* r, const double* camToWorld7, float ab_exposure, double aff_a, double aff_b);
int sdvgn_reproj_set_cur_level(sdvgn_reproj* r, int lvl, const float* dIp_aos3, const float* dIp_aos3_dev);
/* For n candidate points (u, v, idepth = PointHessian::idepth, host_idx = index of pt->host, ref_idx = index of the frame the
 * reference patch is taken from (pt->host when the window has more than two frames, :240-251), type 0 = CORNER / 1 = EDGELET):
 *   px0[2n], cell[n]   Reprojector::reprojectPoint (:602-616): projection into the new frame; grid cell index
 *                      (cell_size 25, ceil(w/25) columns) or -1 when outside the 8-pixel border (the point is not a candidate)
 *   quality[n]         |(dx,dy)| of host->dI at (int)(v*w+u): the key of pointQualityComparator (:186-194)
 *   success[n], px[2n] Reprojector::findMatchDirect (:236-291) started from px0: getWarpMatrixAffine, getBestSearchLevel,
 *                      warpAffine, createPatchFromPatchWithBorder, align2D / align1D (10 iterations max); px = the aligned
 *                      position when success, evaluated for every candidate with cell >= 0
 *   level[n]           search level used (optional, may be NULL)
 * The caller replays the reference's selection (per-cell sort by quality, random cell order, first success per cell, stop after
 * 0.8*setting_desiredImmatureDensity matches, :117-156,196-234) on these arrays.  Deviation: a candidate whose inverse affine
 * warp is NaN fails here; the reference would align it against the previous candidate's stale patch (:64-68). */
int sdvgn_reproj_match(sdvgn_reproj* r, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const int* ref_idx,
                       const int* type, double* px0, int* cell, float* quality, int* success, double* px, int* level);

/* ===================================================================================================
 * Sliding-window back end -- replaces the data-parallel part of class EnergyFunctional
 * (src/OptimizationBackend/EnergyFunctional.h:36-138) and of FullSystem::linearizeAll
 * (src/FullSystem/FullSystemOptimize.cpp:99-159) on a flattened copy of the EF graph.
 *
 * The shim flattens the pointer graph once per makeIDX() (EnergyFunctional.cpp:761-782):
 * frames in EF order, points grouped by host frame, one residual row per (point, target frame).
 * =================================================================================================== */
typedef struct sdvgn_ef sdvgn_ef;

#define SDVGN_MAX_FRAMES 8 /* setting_maxFrames = 7 in the reference (src/util/settings.cpp:46-47) */

/* EnergyFunctional::EnergyFunctional + the level-0 images of the window.  w,h = wG[0],hG[0].  `stream`: a hipStream_t, NULL (the library's
 * shared window stream of that device) or SDVGN_STREAM_OWN (a stream of the handle's own: windows that are optimised side by side,
 * sdvgn_ef_optimize_batch). */
#define SDVGN_STREAM_OWN ((void*)1)
int sdvgn_ef_create(sdvgn_ef** out, int device, int w, int h, int max_points, void* stream);
void sdvgn_ef_destroy(sdvgn_ef* ef);
void* sdvgn_ef_stream(sdvgn_ef* ef);

/* CalibHessian: value_scaled = {fx,fy,cx,cy} and value_minus_value_zero (HessianBlocks.h:262-330). */
int sdvgn_ef_set_calib(sdvgn_ef* ef, const double value_scaled[4], const double value_minus_value_zero[4]);
/* FrameHessian / EFFrame state of the nF window frames (insertFrame, EFFrame::takeData, FrameHessian::setState):
 * evalPT7 = worldToCam_evalPT (Sophus data() layout), state10/state_zero10 = FrameHessian::state / state_zero
 * (unscaled), frameID (0 => initial pose prior, HessianBlocks.h:220-232), ab_exposure, frameEnergyTH. */
int sdvgn_ef_set_frames(sdvgn_ef* ef, int nF, const double* evalPT7, const double* state10, const double* state_zero10,
                        const int* frameID, const float* ab_exposure, const float* frameEnergyTH);
/* A new frame set INVALIDATES the window tables: points and residuals must be set again (sdvgn_ef_set_points / _set_residuals;
 * entry points that need them return SDVGN_E_STATE until then), and the marginalisation prior HM, bM is reset to zero -- a caller that
 * carries a prior over re-installs it with sdvgn_ef_set_marg_prior.
 *
 * FrameHessian::frameEnergyTH of all nF frames as the last linearizeAll left them: FullSystem::setNewFrameEnergyTH
 * (FullSystemOptimize.cpp:63-97) runs inside every linearizeAll (:122) -- here inside sdvgn_ef_linearize_all, every linearisation
 * of sdvgn_ef_optimize and sdvgn_ef_optimize_finish -- and rewrites the NEWEST frame's threshold from the 70th percentile of
 * state_NewEnergyWithOutlier over the active residuals that target it (device-side exact selection, k_ef_select_th). */
int sdvgn_ef_get_frame_energy_th(sdvgn_ef* ef, float* frameEnergyTH /* [nF] */);
/* FrameHessian::dI (level-0 AoS {I,dx,dy}, w*h*3 floats) of frame idx; _raw builds it on the GPU from the float
 * image like FrameHessian::makeImages (HessianBlocks.cpp:107-167). */
int sdvgn_ef_set_frame_image(sdvgn_ef* ef, int idx, const float* dI_aos3);
int sdvgn_ef_set_frame_image_raw(sdvgn_ef* ef, int idx, const float* image);
/* PointHessian / EFPoint (insertPoint, EFPoint::takeData): host frame index, integer pixel u,v, idepth,
 * idepth_zero, color[8], weights[8] (ImmaturePoint.cpp:14-27), hasDepthPrior, isFromSensor.  Points must be
 * grouped by host frame (ascending). */
int sdvgn_ef_set_points(sdvgn_ef* ef, int nP, const int* host, const float* u, const float* v, const float* idepth,
                        const float* idepth_zero, const float* color8, const float* weights8,
                        const unsigned char* hasDepthPrior, const unsigned char* isFromSensor);
/* PointFrameResidual / EFResidual (insertResidual): point index, target frame index, state_state (ResState),
 * hasMatcher + matcher pixel (Residuals.cpp:44-58), isLinearized, isActiveAndIsGoodNEW.  At most one residual per
 * (point,target). */
int sdvgn_ef_set_residuals(sdvgn_ef* ef, int nR, const int* point, const int* target, const int* state_state,
                           const unsigned char* hasMatcher, const double* matcher_xy, const unsigned char* isLinearized,
                           const unsigned char* isActive);
/* EFResidual::takeDataF (EnergyFunctionalStructs.cpp:15-25) for a caller that keeps PointFrameResidual::linearize on the host: the
 * Jacobians the EnergyFunctional side owns (efResidual->J: resF 2, Jpdxi[0] 6, Jpdxi[1] 6, Jpdc[0] 4, Jpdc[1] 4, Jpdd 2 -- the layout of
 * sdvgn_ef_get_residual_J) and res_toZeroF (NULL: zeros) of the residuals given to sdvgn_ef_set_residuals, in that order.  JpJdF is formed
 * here like takeDataF forms it.  After this call sdvgn_ef_solve_system runs EnergyFunctional::solveSystemF on exactly these values -- the
 * drop-in for a host loop that is otherwise unchanged (oracle/dropin/EnergyFunctionalGPU.cpp). */
int sdvgn_ef_set_residual_jacobians(sdvgn_ef* ef, int nR, const float* J24 /*[nR][24]*/, const float* res_toZero2 /*[nR][2] or NULL*/);
/* EnergyFunctional::HM, bM (marginalisation prior) and lastNullspaces_pose + _scale (EnergyFunctional.h:96-119). */
int sdvgn_ef_set_marg_prior(sdvgn_ef* ef, const double* HM, const double* bM);
int sdvgn_ef_set_nullspaces(sdvgn_ef* ef, int k, const double* vectors);
/* FullSystem::getNullspaces (FullSystemOptimize.cpp:548-588, called by FullSystem::solveSystem :504-513 before every solveSystemF) from the
 * frames' linearisation points, with the per-frame columns of FrameHessian::setStateZero (HessianBlocks.cpp:57-76): installs the 6 pose + 1
 * scale vectors that orthogonalize() projects out.  Host arithmetic (fp64, a dozen SE(3) exp / log per frame); call it after set_frames. */
int sdvgn_ef_compute_nullspaces(sdvgn_ef* ef);
/* the installed vectors, [k][4+6nF] row-major; returns k */
int sdvgn_ef_get_nullspaces(sdvgn_ef* ef, double* out, int cap_vectors);

/* FullSystem::setPrecalcValues (FrameFramePrecalc::set for every pair, HessianBlocks.cpp:169-195) +
 * EnergyFunctional::setDeltaF (EnergyFunctional.cpp:131-156). */
int sdvgn_ef_set_precalc(sdvgn_ef* ef);
/* Finish handing a window over: whatever the setters above left on the host side -- the window constants of the device-side solve
 * (adjoints, priors, marginalisation prior, null-space vectors), the frame states and the calibration value -- is copied to the device
 * now and the stream is drained, so that the next sdvgn_ef_optimize call starts from HBM-resident inputs and issues no host-to-device
 * copy.  Optional: optimize uploads what is still pending itself. */
int sdvgn_ef_make_resident(sdvgn_ef* ef);
/* EnergyFunctional::setAdjointsF (EnergyFunctional.cpp:21-71). */
int sdvgn_ef_set_adjoints(sdvgn_ef* ef);

/* Vec3 FullSystem::linearizeAll(false)[0]: PointFrameResidual::linearize over all non-linearised residuals
 * (FullSystemOptimize.cpp:99-118, Residuals.cpp:60-224).  energy_out = stats[0]. */
int sdvgn_ef_linearize_all(sdvgn_ef* ef, double* energy_out);
/* FullSystem::applyRes_Reductor(true): PointFrameResidual::applyRes over the same residuals (Residuals.cpp:252-275). */
int sdvgn_ef_apply_res(sdvgn_ef* ef);
/* void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian*)   EnergyFunctional.cpp:650-759
 * (accumulateAF/LF/SCF, HM/bM, damped Jacobi-preconditioned LDLT, orthogonalize for iteration >= 2, resubstitute).
 * x_out[4+6nF] = lastX; frame/calib steps are -x, point steps stay on the device (sdvgn_ef_get_points). */
int sdvgn_ef_solve_system(sdvgn_ef* ef, int iteration, double lambda, double* x_out);
/* 0: the last device-side solve used the fast unpivoted LDL^T (the system was positive definite, as it is by construction); 2: a pivot was
 * not positive / finite (an indefinite system, e.g. an indefinite marginalisation prior) and the solve fell back to Eigen's diagonally
 * pivoted LDL^T (LDLT.h: pivot = largest remaining |diagonal|; pseudo-inverse of D), which like the reference's
 * `HFinalScaled.ldlt().solve()` (EnergyFunctional.cpp:743) returns a finite x on such a system.  No failure path. */
int sdvgn_ef_get_solve_status(sdvgn_ef* ef);
/* FullSystem::optimize (sdvgn_ef_optimize, no trace) on B INDEPENDENT windows in one call -- several maps / agents / sub-maps on one GPU.
 * One loop body of one window is a chain of six short launches that occupies a fraction of the chip for ~65 us.  Default: the B loops run as ONE
 * launch sequence (sdvgn_ef_optimize_lockstep below).  Windows that form does not take, and every window when SDVGN_BATCH_THREADS is set in the
 * environment, run as B sdvgn_ef_optimize calls on a library-owned pool of host threads; their chains overlap on the device when every handle
 * has its own stream (create with SDVGN_STREAM_OWN).  its_out[b] (may be NULL) = return value of window b's sdvgn_ef_optimize.  Returns
 * SDVGN_OK or the error of a window.  Results of every window are those of its own sdvgn_ef_optimize call, bit for bit. */
int sdvgn_ef_optimize_batch(sdvgn_ef* const* handles, int B, int mnumOptIts, int flags, int* its_out);
/* The form sdvgn_ef_optimize_batch takes by default: the B loops as ONE launch sequence -- every kernel of a loop body launched once for
 * all windows (the window is a grid index), the accept / reject decision, lambda and the choice of state copies of every window kept in
 * device memory, no host round trip inside the call (csrc/backend_lockstep.inc).  Single-rank windows in the product's default mode only
 * (flags: bit0 = exactly mnumOptIts bodies); SDVGN_E_ARG otherwise (sdvgn_ef_optimize_batch then falls back to one host thread per window).
 * trace (may be NULL): [B][trace_cap][trace_stride] doubles, the rows of sdvgn_ef_optimize's trace per window.  Per window the results are
 * those of its own sdvgn_ef_optimize call, bit for bit (FullSystemOptimize.cpp:344-458).  Thread-safe across disjoint sets of handles: the
 * library keeps two launch-sequence pools (staging + stream) per device, so two host threads can each have a call in flight on one device (a third
 * waits); measured, two sequences side by side are 10-15 % slower in aggregate than one sequence over all the windows. */
int sdvgn_ef_optimize_lockstep(sdvgn_ef* const* handles, int B, int mnumOptIts, int flags, int* its_out, double* trace, int trace_stride,
                               int trace_cap);
/* Arithmetic of k_ef_linearize (PointFrameResidual::linearize, Residuals.cpp:60-224): 0 = the reference's float arithmetic operation by
 * operation (default: IEEE divisions / square roots, no contraction; J, energies and residual states bit-identical with the CPU path),
 * 1 = tolerance mode: fused multiply-adds and the hardware's 1-ulp reciprocal / square root (v_rcp_f32, v_sqrt_f32) for the 34 divisions and
 * 18 square roots per residual.  BASELINE.json's contract for this path is 1e-4 relative on the increments; mode 1 keeps it, with residual
 * states identical except at threshold ties (tests/test_backend_gpu.py).  Affects every later linearise of the handle. */
int sdvgn_ef_set_arith(sdvgn_ef* ef, int mode);
/* doStepFromBackup / backupState / loadSateBackup for the per-point idepths (FullSystemOptimize.cpp:165-321):
 * mode 0: backup = idepth; mode 1: idepth = idepth_zero = backup + stepfac*step; mode 2: idepth = idepth_zero = backup */
int sdvgn_ef_point_step(sdvgn_ef* ef, int mode, float stepfacD);
/* new frame states after a step (FrameHessian::setState) -- follow with sdvgn_ef_set_precalc */
int sdvgn_ef_set_frame_states(sdvgn_ef* ef, const double* state10);

/* The loop of FullSystem::optimize (FullSystemOptimize.cpp:344-458): resetOOB, linearizeAll + calcLEnergy/calcMEnergy,
 * then per iteration backupState / solveSystem / doStepFromBackup / linearizeAll / accept (applyRes, lambda*0.25) or
 * reject (loadSateBackup, re-linearise, lambda*100), break on a tiny step.  One loop body = one "Gauss-Newton
 * iteration" of BASELINE.json's metric.  trace rows: {iteration, lambda, accepted, E, E_L, E_M, canbreak, x[4+6nF]}.
 * Everything runs on the device, the accept test included (the host supplies its parts of the comparison, mirrors the frame states and
 * keeps the books); a body is six launches, the next body's accumulate is queued before the verdict of the current one is known, and
 * setNewFrameEnergyTH / the re-classification after a rejected step ride in launches of the following body -- results are those of the
 * literal order of operations (flags bit1), bit for bit.
 * Returns the number of iterations run (>= 0) or an error (< 0). */
int sdvgn_ef_optimize(sdvgn_ef* ef, int mnumOptIts, int flags /* bit0: run exactly mnumOptIts bodies (bench); bit1: re-linearise after a
                      rejected step literally like FullSystemOptimize.cpp:446-449 instead of switching back to the kept state_New* set
                      (same results, bit for bit -- tests/test_backend_gpu.py); bit2 (opt-in): the body that follows a rejected step
                      works on the state its predecessor's normal equations were built on, so it re-uses the stitched HA/bA/Hsc/bsc and the
                      per-point Schur terms (only lambda changed) instead of accumulating again -- same trace and final state, bit for bit;
                      bit3 (measurement): a HIP event pair around every k_ef_linearize launch of the call, sdvgn_ef_get_linearize_times;
                      bit4 (A/B, tests): do NOT solve the rejected case of every body ahead.  By default, while a trial step is linearised, one
                      workgroup on the library's side stream factors the system the body has just solved with the damping a rejection would
                      bring (lambda * 100, iteration + 1: FullSystemOptimize.cpp:446-458 solves the SAME normal equations again); a rejected
                      step is then followed by resubstitute + step + linearise only.  Same trace and final state, bit for bit.  Where the
                      side stream is not served beside the main one (a serialising profiler) the look-ahead gives up after a bounded wait
                      and the loop solves the rejected case itself; nothing fails.
                      Environment: SDVGN_FUSED_APPLY=1 lets the linearise leave applyRes in second copies of the planes it writes (what
                      sdvgn_ef_optimize_lockstep always does) instead of apply workgroups in the statistics launch -- same results */,
                      double* trace, int trace_stride, int trace_cap);
/* trace rows (continued): if trace_stride > 7 + (4+6nF), column 7 + (4+6nF) holds the newest frame's frameEnergyTH as the trial
 * linearizeAll of that iteration left it.
 *
 * Tail of FullSystem::optimize, FullSystemOptimize.cpp:460-470: frameHessians.back()->setEvalPT(PRE_worldToCam, state with only the
 * affine part kept), ef->setAdjointsF, setPrecalcValues, lastEnergy = linearizeAll(true) -- i.e. linearize + applyRes(true) for every
 * active residual, the isNew bookkeeping of its point (:34-47), setNewFrameEnergyTH, and the toRemove list (:136-155): residuals
 * that are not active afterwards are dropped (their slots cease to exist on the device as well).
 * lastEnergy_out = lastEnergy[0]; relbs_max[nP]: per point the largest relBS over its surviving residuals (caller:
 * maxRelBaseline = max(maxRelBaseline, relbs_max[p])); ngood_inc[nP]: numGoodResiduals increments; removed[nR] (order of
 * sdvgn_ef_set_residuals): 1 = the residual is in toRemove.  Any output may be NULL.  The caller reads the new linearisation point back
 * with sdvgn_ef_get_state / its own FrameHessian (evalPT = PRE_worldToCam of the newest frame) and the states with
 * sdvgn_ef_get_residual_state (lastResiduals bookkeeping, :128-134). */
int sdvgn_ef_optimize_finish(sdvgn_ef* ef, double* lastEnergy_out, float* relbs_max, int* ngood_inc, unsigned char* removed);
/* AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:12-21) zeroes PointHessian::maxRelBaseline (and idepth_hessian) of a point it finds
 * WITHOUT an active residual -- in every solveSystemF of the loop, not only the last one: a point that has none in one iteration and regains one later
 * (applyRes of an accepted step) restarts its running maximum.  out[nP] (dense index): 1 if some solve of the LAST sdvgn_ef_optimize call found the point
 * so; a caller that mirrors maxRelBaseline applies `if (out[p]) maxRelBaseline = 0` BEFORE folding sdvgn_ef_optimize_finish's relbs_max in.  Returns nP. */
int sdvgn_ef_get_point_nogood(sdvgn_ef* ef, unsigned char* out);
/* (no member of the reference: its loop cannot fail this way)  A compute call that returned SDVGN_E_STATE because a workgroup gave up a bounded intra-launch wait,
 * or because a commit carried a fixed linearisation over, leaves a STICKY error word on the handle: every later compute call fails until the window is loaded again
 * through sdvgn_ef_set_frames.  A caller that keeps the window on the device from key-frame to key-frame can instead clear the word here once it has put the
 * window right (dropped / re-sent what the failed call concerned, then sdvgn_ef_set_adjoints + sdvgn_ef_set_precalc): returns the word as it was (0: nothing was
 * raised; bit 0 the accept verdict, bit 1 the solution, bit 3 a fixed linearisation in a commit), and the next solve accumulates anew. */
int sdvgn_ef_clear_error(sdvgn_ef* ef);
/* ---- key-frame cycle around optimize: marginalisation (SURVEY 8 row b2 mode 2, EnergyFunctional.cpp:434-597) -------------------------
 * void EFResidual::fixLinearizationF(EnergyFunctional*)   EnergyFunctionalStructs.cpp:45-55, for every ACTIVE residual of the points with
 * mask[p] != 0 (FullSystem::flagPointsForRemoval calls it after re-linearising + applying those residuals, FullSystem.cpp:771-783 --
 * here: sdvgn_ef_linearize_all + sdvgn_ef_apply_res): res_toZeroF = resF - (Jpdxi . adHTdeltaF + Jpdc . cDeltaF + Jpdd * deltaF),
 * isLinearized = true. */
int sdvgn_ef_fix_linearization(sdvgn_ef* ef, const unsigned char* mask /* [nP] */);
/* PointFrameResidual::resetOOB (Residuals.h:70-76: state_NewEnergy = state_energy = 0, state_NewState = OUTLIER, state_state = IN) for the
 * non-linearised residuals of the points with mask[p] != 0 (NULL: all points, the head of FullSystem::optimize :353-364). */
int sdvgn_ef_reset_oob(sdvgn_ef* ef, const unsigned char* mask /* [nP] or NULL */);
/* void EnergyFunctional::marginalizePointsF()  EnergyFunctional.cpp:514-576 (+ dropPointsF :578-597) under the reference's
 * setting_solverMode (SOLVER_ORTHOGONALIZE_X_LATER: no null-space branch): for the points with marg[p] != 0 (stateFlag == PS_MARGINALIZE)
 * priorF *= setting_idepthFixPriorMargFac, AccumulatedTopHessianSSE::addPoint<2> + AccumulatedSCHessianSSE::addPoint(p, false) on the
 * device, stitch without priors, HM += setting_margWeightFac * (M - Msc), bM likewise.  Those points and the ones with drop[p] != 0
 * (PS_DROP; drop may be NULL) then leave the window (their residual slots cease to exist).  Single-GPU entry point. */
int sdvgn_ef_marginalize_points(sdvgn_ef* ef, const unsigned char* marg /* [nP] */, const unsigned char* drop /* [nP] or NULL */);
/* EnergyFunctional::HM, bM as they stand ((4+6nF)^2 and 4+6nF doubles; zeros if none was set). */
int sdvgn_ef_get_marg_prior(sdvgn_ef* ef, double* HM, double* bM);
/* void EnergyFunctional::marginalizeFrame(EFFrame*)  EnergyFunctional.cpp:434-512, the algebra on HM / bM: frame idx is moved to the end,
 * its prior added (:468-469), the Schur complement taken on the preconditioned system (:471-493).  Host-only and pure: outputs are
 * (4+6(nF-1))^2 and 4+6(nF-1) doubles; the caller rebuilds the window without the frame (sdvgn_ef_set_frames ...) and installs them with
 * sdvgn_ef_set_marg_prior, like the reference re-indexes its frames (:495-511). */
int sdvgn_ef_marginalize_frame(sdvgn_ef* ef, int idx, double* HM_out, double* bM_out);
/* EFResidual::res_toZeroF (nR x 2) and isLinearized (nR), in the order of sdvgn_ef_set_residuals (tests). */
int sdvgn_ef_get_res_toZero(sdvgn_ef* ef, float* res_toZero2, unsigned char* isLinearized);

/* ---- the window kept resident from key-frame to key-frame (csrc/backend_window.inc) -------------------------------------------------
 * The reference never rebuilds its graph; between two FullSystem::optimize calls it MUTATES it (EnergyFunctional.h:51-58).  These entry
 * points mirror those members, so that a key-frame costs one image upload and a few hundred kB of edits instead of every table and every
 * image again (the whole-plane setters above).  Edits are collected on the host; sdvgn_ef_make_idx -- EnergyFunctional::makeIDX, the
 * reference's own commit point (EnergyFunctional.cpp:761-782) -- applies them ON THE DEVICE (one gather pass over the per-point planes and the
 * flags / state / matcher planes; new points and residual edits from 96- / 16-byte records the entry points leave in pinned memory and send
 * at once -- the argument arrays may be reused when a call returns -- resolved to table slots by the commit's kernels).  Until the commit
 * every other entry point sees the window of the last commit -- with ONE exception: the image slot of a frame removed in this session is free at once, and a
 * sdvgn_ef_insert_frame of the same session may upload into it (with SDVGN_MAX_FRAMES frames in the window there is no other slot); a compute call between that
 * insert and the commit would read the new image for the removed frame.  The reference's key-frame (marginalizeFrame ... insertFrame ... makeIDX, then optimize)
 * has no such call; run sdvgn_ef_make_idx first if yours does.  After it the tables are bit-identical with a reload of the same graph in the
 * same order through the setters above.
 *   Points are addressed by ids (sdvgn_ef_insert_points returns them; a window loaded through sdvgn_ef_set_points has id = index) that stay
 *   valid until the point is removed; the id of a removed point may be returned for a new point after the NEXT commit, never before (an edit
 *   recorded on a point that leaves before the commit is void), so ids stay below 2 x max_points for any length of run.
 *   Frames are addressed by their CURRENT index (like EFFrame::idx, renumbered by a removal), residuals by (point id, target frame index); an
 *   edit towards a frame that is removed before the commit is void.
 *   Inside a host frame the points keep EFFrame::points' order: insertPoint appends, removePoint moves the LAST point into the hole
 *   (EnergyFunctional.cpp:414-432, 597-620).  sdvgn_ef_get_point_ids gives the dense order after a commit; per-point getters and the
 *   slot tables (slot = target * nP + dense index) follow it.
 *   Surviving points keep what the device holds (the inverse depths of the last optimize), surviving residuals their state_state / matcher /
 *   isActive; Jacobians and energies restart from zero like after sdvgn_ef_set_residuals (the next optimize re-linearises everything).  A
 *   residual with a fixed linearisation cannot be carried over a commit (the reference's live inside ONE flagPointsForRemoval ->
 *   marginalizePointsF pair, FullSystem.cpp:771-800): the next compute call fails with SDVGN_E_STATE.  Single-rank handles only.
 *
 * EFFrame* EnergyFunctional::insertFrame(FrameHessian*, CalibHessian*)  EnergyFunctional.cpp:352-398: the frame gets index nF (returned);
 * HM / bM grow by six zero rows / columns (:363-368); its level-0 image -- dI_aos3 (FrameHessian::dI) or, if NULL, built on the device from
 * `image` like FrameHessian::makeImages -- goes to a free image slot now (asynchronous on the handle's stream; the buffer may be reused
 * after the next sdvgn_ef_make_idx). */
int sdvgn_ef_insert_frame(sdvgn_ef* ef, const double* evalPT7, const double* state10, const double* state_zero10, int frameID, float ab_exposure,
                          float frameEnergyTH, const float* dI_aos3, const float* image);
/* void EnergyFunctional::marginalizeFrame(EFFrame*)  EnergyFunctional.cpp:434-512: HM, bM <- the Schur complement without the frame
 * (sdvgn_ef_marginalize_frame's algebra on the edited window; or the caller's own HM_new / bM_new, (4+6(nF-1))^2 and 4+6(nF-1) doubles),
 * the frames behind it move down by one (:495-503), its image slot is free.  Points it still hosts and residuals that still target it
 * leave with it (the reference has dropped or marginalised them before: FullSystemMarginalize.cpp). */
int sdvgn_ef_remove_frame(sdvgn_ef* ef, int idx, const double* HM_new, const double* bM_new);
/* FrameHessian::setState / setEvalPT of the current frames ([nF][7], [nF][10], [nF][10], [nF]) without touching the tables: what the host
 * loop changed since the last optimize (e.g. makeKeyFrame's setEvalPT_scaled of the new frame, FullSystem.cpp:1047).  Follow with
 * sdvgn_ef_make_idx, sdvgn_ef_set_adjoints, sdvgn_ef_set_precalc. */
int sdvgn_ef_update_frames(sdvgn_ef* ef, int nF, const double* evalPT7, const double* state10, const double* state_zero10, const float* ab_exposure);
/* EFPoint* EnergyFunctional::insertPoint(PointHessian*)  :414-432, n points (arguments as sdvgn_ef_set_points); point_id_out[n] (may be NULL) */
int sdvgn_ef_insert_points(sdvgn_ef* ef, int n, const int* host, const float* u, const float* v, const float* idepth, const float* idepth_zero,
                           const float* color8, const float* weights8, const unsigned char* hasDepthPrior, const unsigned char* isFromSensor,
                           int* point_id_out);
/* void EnergyFunctional::removePoint(EFPoint*)  :597-620 (what dropPointsF :578-595 and marginalizePointsF :568-574 do to their points):
 * the points leave with all their residuals */
int sdvgn_ef_remove_points(sdvgn_ef* ef, int n, const int* point_id);
/* EFResidual* EnergyFunctional::insertResidual(PointFrameResidual*)  :400-412: state_state, hasMatcher + matcher pixel as in
 * sdvgn_ef_set_residuals; a new residual is neither linearised nor active.  Between two commits a (point, target) pair may be inserted once,
 * updated once and dropped once; the commit applies all inserts, then all updates, then all drops. */
int sdvgn_ef_insert_residuals(sdvgn_ef* ef, int n, const int* point_id, const int* target, const int* state_state, const unsigned char* hasMatcher,
                              const double* matcher_xy);
/* state_state and matcher of residuals that exist already (PointFrameResidual::findMatches gives an old residual its matcher when the
 * key-frame that brings one arrives, FullSystem.cpp:1121-1133); residuals that do not exist are left alone */
int sdvgn_ef_update_residuals(sdvgn_ef* ef, int n, const int* point_id, const int* target, const int* state_state, const unsigned char* hasMatcher,
                              const double* matcher_xy);
/* void EnergyFunctional::dropResidual(EFResidual*)  :578-595 */
int sdvgn_ef_drop_residuals(sdvgn_ef* ef, int n, const int* point_id, const int* target);
/* void EnergyFunctional::makeIDX()  :761-782: the edits since the last commit become the window.  Follow with sdvgn_ef_set_marg_prior (only if
 * the prior changed outside sdvgn_ef_remove_frame, e.g. marginalizePointsF on the host), sdvgn_ef_set_adjoints, sdvgn_ef_set_precalc. */
int sdvgn_ef_make_idx(sdvgn_ef* ef);
/* id of the point at every dense index [nP]; returns nP */
int sdvgn_ef_get_point_ids(sdvgn_ef* ef, int* id_of_index);
/* the residual tables by slot (slot = target * nP + dense point index), nF * nP entries each, any output may be NULL: exists, state_state,
 * state_NewState, state_energy, state_NewEnergy, state_NewEnergyWithOutlier, isActiveAndIsGoodNEW.  Returns nF * nP.  (On a handle whose
 * window was edited in place sdvgn_ef_optimize_finish's `removed` output is such a table too: there is no caller-side residual list.) */
int sdvgn_ef_get_residual_table(sdvgn_ef* ef, unsigned char* exists, signed char* state_state, signed char* state_new, float* energy, float* energy_new,
                                float* energy_with_outlier, unsigned char* isActive);

/* state after optimize: CalibHessian::value_scaled, FrameHessian::state (nF x 10), PointHessian::idepth (nP) */
int sdvgn_ef_get_state(sdvgn_ef* ef, double* value_scaled4, double* state10, float* idepth);
/* wall time (microseconds, host steady_clock) of every loop body of the last sdvgn_ef_optimize call; returns their number */
int sdvgn_ef_get_iteration_times(sdvgn_ef* ef, double* us, int cap);
/* number of accepted steps of the last sdvgn_ef_optimize call (what a trace would show, without asking for one) */
int sdvgn_ef_get_accepted_steps(sdvgn_ef* ef);
/* the look-ahead of the last sdvgn_ef_optimize call (flags bit4 switches it off; it also stands down where the side stream is not served beside
 * the main one): rejected cases solved ahead on the side stream, and loop bodies that started from such a solution */
int sdvgn_ef_get_look_ahead(sdvgn_ef* ef, int* launched, int* used);
/* durations (milliseconds, HIP events on the library's stream) of the k_ef_linearize launches of the last sdvgn_ef_optimize call that ran
 * with flags bit3, in launch order (the first is the call's initial linearizeAll); returns their number */
int sdvgn_ef_get_linearize_times(sdvgn_ef* ef, float* ms, int cap);
/* Device pointer of key-frame idx's level-0 image (dI, AoS {I,dx,dy}) held by the window, for sdvgn_reproj_set_frame(.., dI_aos3_dev);
 * NULL if idx is out of range or the handle is host-only.  Valid until that frame's image is replaced or the handle is destroyed. */
const float* sdvgn_ef_frame_image_dev(sdvgn_ef* ef, int idx);

/* FullSystem::optimizeImmaturePoint(ImmaturePoint*, int minObs, ImmaturePointTemporaryResidual*)   FullSystemOptPoint.cpp:18-185,
 * with ImmaturePoint::linearizeResidual (ImmaturePoint.cpp:410-477), for n immature points in one launch (the loop of
 * activatePointsMT_Reductor, FullSystem.cpp:555-566).  Uses the window loaded on the handle: calibration, frame images and the
 * frames' current states (PRE_RTll / PRE_tTll / PRE_aff_mode of FrameFramePrecalc, HessianBlocks.cpp:169-195).
 * Per point: host index, u, v, idepth_min, idepth_max, energyTH, color[8], weights[8], isFromSensor.  Outputs:
 *   result[i]   0 = the reference returns 0 (not well constrained, :57-63,:83-90), -1 = it returns (PointHessian*)-1
 *               (NaN idepth / fewer than minObs inlier residuals / NaN energyTH, :125-143), 1 = activate
 *   idepth[i]   inverse depth for setIdepth / setIdepthZero (:150-159) when result is 1 (NaN otherwise)
 *   res_state[i*nF + t]  ResState of the temporary residual towards frame t (0 IN, 1 OOB, 2 OUTLIER; -1 for t == host): the
 *               caller creates PointFrameResiduals for the IN ones (:163-181). */
int sdvgn_ef_optimize_immature(sdvgn_ef* e, int n, const int* host, const float* u, const float* v, const float* idepth_min,
                               const float* idepth_max, const float* energyTH, const float* color8, const float* weights8,
                               const unsigned char* isFromSensor, int minObs, int* result, float* idepth, int* res_state);

/* parity / read-back hooks */
int sdvgn_ef_dim(sdvgn_ef* ef);
int sdvgn_ef_get_system(sdvgn_ef* ef, double* HA, double* bA, double* Hsc, double* bsc, double* HFinal, double* bFinal);
int sdvgn_ef_get_residual_J(sdvgn_ef* ef, int which /*0 new, 1 EF*/, float* out24 /*[nR][24]*/);
/* (which = 1 is the reference's efResidual->J for residuals the EnergyFunctional uses, i.e. active ones; which = 0 is `J` as the last
 * linearize left it, valid until the next applyRes.  applyRes swaps the two buffers of every existing, not fixed residual -- the reference's
 * takeDataF only where the new state is IN --, so the rows of a residual that is not active afterwards carry no meaning: no consumer reads them.) */
int sdvgn_ef_get_residual_state(sdvgn_ef* ef, int* state_state, int* state_new, float* energy_new,
                                float* energy_with_outlier, unsigned char* isActive);
int sdvgn_ef_get_points(sdvgn_ef* ef, float* out9 /*[nP][Hdd_accAF,bd_accAF,Hcd_accAF x4,HdiF,bdSumF,step]*/);
int sdvgn_ef_get_top_acc(sdvgn_ef* ef, double* out /*[nF*nF][11*11], index h + nF*t*/, int* resInA);
/* device pointer + element count of the packed accumulator buffer that cfg4 all-reduces across ranks (doubles):
 * top Gram [nF*nF][121] (live 11x11) followed by SC Gram [nF][1431] (upper triangle of the live 53x53) followed by resInA. */
int sdvgn_ef_accumulators_dev(sdvgn_ef* ef, double** buf_dev, int* count);
/* solve_system split for multi-GPU: accumulate only (fills the packed buffer), then finish (stitch, solve, resubstitute)
 * after the caller has all-reduced the buffer. */
int sdvgn_ef_accumulate(sdvgn_ef* ef);
/* Sharded window, ONE collective per loop body (configs[3]; north_star: a single all-reduce per Gauss-Newton iteration).  With a collective
 * buffer installed -- `buf_dev`: device memory of >= 2 x sdvgn_ef_collective_stride(ef) doubles (e.g. a torch tensor the all-reduce callback
 * can address), or NULL for a library allocation -- sdvgn_ef_optimize of a sharded handle (sdvgn_ef_init_rccl / sdvgn_ef_set_allreduce)
 * sends ONE message per loop body, + one per call: [packed accumulators | 4 statistics | max_points quantile candidates].  It applies the
 * trial linearisation and accumulates it BEFORE the accept test (whose energy rides in the same message): an accepted step costs nothing
 * extra, a rejected one takes the apply back and recomputes the rank-local per-point planes.  Results are those of the two-collective loop
 * bit for bit.  sdvgn_ef_collective_count: all-reduces the handle has issued so far. */
int sdvgn_ef_collective_stride(sdvgn_ef* ef);
int sdvgn_ef_set_collective_buffer(sdvgn_ef* ef, double* buf_dev, int capacity);
unsigned long long sdvgn_ef_collective_count(sdvgn_ef* ef);
int sdvgn_ef_finish_solve(sdvgn_ef* ef, int iteration, double lambda, double* x_out);
/* host part of solveSystemF on a caller-supplied (e.g. all-reduced) accumulator buffer in host memory; works on a
 * host-only handle (sdvgn_ef_create with device = -1: no kernels, only frames/adjoints/priors and this function). */
int sdvgn_ef_stitch_solve_host(sdvgn_ef* ef, const double* acc_host, int iteration, double lambda, double* x_out);
int sdvgn_ef_accumulator_count(sdvgn_ef* ef);
/* cfg4 plumbing: let the caller own the two reducible device buffers (e.g. torch tensors), and register the
 * collective: fn(user, buf_dev, count) must sum buf_dev[0..count) over all ranks in stream order (RCCL all-reduce).
 * With a callback registered, sdvgn_ef_solve_system / sdvgn_ef_optimize all-reduce the packed accumulators (once per
 * GN iteration) and the 4 energy/step statistics (once per linearizeAll). */
int sdvgn_ef_set_external_buffers(sdvgn_ef* ef, double* acc_dev, int acc_capacity, double* stats_dev, int stats_capacity);
/* stats_dev: 4 + max_points doubles -- {energy, L-energy, sum step^2, sum |idepth|} followed by this rank's candidates of
 * setNewFrameEnergyTH's quantile (energy + 1 per point, 0 = none); ONE all-reduce per linearizeAll carries both. */
int sdvgn_ef_set_allreduce(sdvgn_ef* ef, void (*fn)(void* user, double* buf_dev, int count), void* user);
/* The same collectives issued by the library itself through RCCL (ncclAllReduce on the handle's stream, in place, ncclDouble /
 * ncclSum) -- no callback, no Python in the iteration.  librccl.so is resolved at run time (the instance already loaded by
 * torch.distributed when there is one).  Rank 0 obtains a 128-byte id with sdvgn_rccl_unique_id and distributes it by any means;
 * every rank then calls sdvgn_ef_init_rccl (collective, like ncclCommInitRank).  Takes precedence over set_allreduce;
 * id128 == NULL drops the communicator again. */
int sdvgn_rccl_unique_id(unsigned char* out128);
int sdvgn_ef_init_rccl(sdvgn_ef* ef, const unsigned char* id128, int rank, int world);
/* Handles of one process initialised with the same id share ONE communicator (reference-counted; the last handle to drop it destroys it):
 * only the first sdvgn_ef_init_rccl with an id is collective.  An id whose communicator is gone must not be used again (like any
 * ncclUniqueId); sdvgn_rccl_comm_alive tells whether this process still holds the communicator of `id128` (1) or not (0). */
int sdvgn_rccl_comm_alive(const unsigned char* id128);
/* number of ranks of the RCCL communicator behind this handle's collectives (ncclCommCount); 0: no communicator (single GPU / callback path) */
int sdvgn_ef_rccl_ranks(sdvgn_ef* ef);
/* restrict this rank's work to host frames [h0,h1) (cfg4: frames sharded across GPUs); default all. */
int sdvgn_ef_set_host_range(sdvgn_ef* ef, int h0, int h1);

#ifdef __cplusplus
}
#endif
#endif /* SDVGN_H */
