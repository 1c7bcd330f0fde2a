// tracker.hip -- host side of the GPU coarse tracker + its C ABI (include/sdvgn.h).
//
// Mirrors class CoarseTracker (src/FullSystem/CoarseTracker.h:17-107): makeK, the pc_* reference template,
// calcRes / calcGSSSE (fused into k_res_gs), and the coarse-to-fine Levenberg-Marquardt driver
// trackNewestCoarse (CoarseTracker.cpp:662-838).  The photometric work runs in the kernels of
// tracker_kernels.hpp; the 8x8 LDLT, SE3 exp and accept/reject logic live in gnmath.hpp and run either on the
// host (sdvgn_tracker_track: one fused launch + one 640-byte read-back per LM trial) or on the device
// (k_track, sdvgn_tracker_track_batch: whole loop in one launch, one workgroup per pose hypothesis).
#include "../../include/sdvgn.h"
#include "gnmath.hpp"
#include "tracker_kernels.hpp"
#include "waitflag.hpp"
#include "devmem.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <vector>

#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return -(int)_e;        \
    } while (0)

using namespace sdvgn;

namespace sdvgn {
#include "tracker_track_kernel.inc"
#include "tracker_struct_pose.inc"
#include "tracker_trace_points.inc"
#include "tracker_coarse_depth.inc"
}

struct sdvgn_tracker {
    int device = 0;
    int levels = 0;
    int w[SDVGN_MAX_LEVELS], h[SDVGN_MAX_LEVELS];
    int max_points = 0, max_batch = 0, max_chunks = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // intrinsics pyramid (makeK)
    float fx[SDVGN_MAX_LEVELS], fy[SDVGN_MAX_LEVELS], cx[SDVGN_MAX_LEVELS], cy[SDVGN_MAX_LEVELS];
    float Ki[SDVGN_MAX_LEVELS][9];
    bool haveK = false;

    // settings
    float huberTH = 6.f, coarseCutoffTH = 20.f, affineOptModeA = 0.f, affineOptModeB = 0.f;

    // reference template pc_* (packed {u,v,idepth,color}) and reference frame state
    float4* pc_dev[SDVGN_MAX_LEVELS] = {};
    int pc_n[SDVGN_MAX_LEVELS] = {};
    float ref_exposure = 1.f;
    double ref_a = 0, ref_b = 0;

    // new frame
    float* pyr_dev[SDVGN_MAX_LEVELS] = {};  // AoS {I,dx,dy}
    __half* pyr_half_dev[SDVGN_MAX_LEVELS] = {};  // precision study only: fp16 {I,dx,dy,0}, built lazily
    bool half_valid = false;
    float4* pyr_rec_dev[SDVGN_MAX_LEVELS] = {};   // PREC_F32_REC: 64-byte neighbourhood records per pixel (k_pyr_to_records), built lazily
    bool rec_valid = false;
    int precision = PREC_F32;
    int arith = 0;                    // 0: the reference's arithmetic (default); 1: tolerance mode (FMA + rcp divisions), sdvgn_tracker_set_arith
    ProblemPtrs* ptrs_dev = nullptr;  // per-problem template / image pointers of sdvgn_tracker_res_and_gs_multi (max_batch entries)
    float* img_stage_dev = nullptr;         // level-0 float image staging
    float new_exposure = 1.f;
    bool haveNew = false;

    // work buffers
    LevelParams* params_dev = nullptr;
    LevelParams* params_host = nullptr;  // pinned
    float* partial_dev = nullptr;
    double* out_dev = nullptr;
    double* out_host = nullptr;          // pinned
    int* flag_host = nullptr;            // pinned completion flag of the host-driven trial (waitflag.hpp)
    int flag_seq = 0;
    float* terms_dev = nullptr;
    int* status_dev = nullptr;
    int terms_lvl = -1;
    TrackState* track_host = nullptr;    // pinned
    TeamMem* team_dev = nullptr;         // k_track_team: partial rows + counters per hypothesis (allocated on first use, zeroed)
    int team_cap = 0;
    int cu_count = 0;                    // multiProcessorCount of the device (queried once)
    int team_capacity = 0;               // workgroups of k_track_team the device holds at once (occupancy query x CUs)
    int team_fallbacks = 0;              // batches re-run on k_track because a team's members did not meet
    int team_mode = 0;                   // sdvgn_tracker_set_team: 0 automatic, -1 always k_track (one workgroup), T >= 1 fixed team size
    int last_team = 0;                   // team size of the last track_batch call (0: k_track)
    int track_seq = 0;                   // sequence number of track_batch calls (TrackState::done)
    unsigned team_seq = 0;               // launch number of k_track_team (20 bits; part of every exchanged word's tag)

    // side outputs of the last track() call
    std::vector<double> trace;

    // ImmaturePoint::traceOn (tracker_trace_points.inc): static per-point data resident on the device, dynamic state in pinned memory
    int tp_n = 0, tp_cap = 0;
    float* tp_static_dev = nullptr;     // u | v | energyTH | gradH(4) | color(8) | weights(8) | host_idx : 23 words per point
    void* tp_state_host = nullptr;      // pinned: idepth_min | idepth_max | quality | status | lastTraceUV(2) | interval : 7 words per point

    // makeCoarseDepthL0 (tracker_coarse_depth.inc): per-level maps, row bookkeeping, host-side ordered pre-accumulation
    float* cd_maps = nullptr;            // idepth | wsum | wdil for all levels
    int* cd_rows = nullptr;              // rowcount | rowoff
    int* cd_n_host = nullptr;            // pinned: pc_n per level, written by k_cd_scan
    void* cd_stage = nullptr;            // pinned: unique pixels (pix | val | wgt), read by k_cd_scatter directly
    int cd_stage_cap = 0;
    std::vector<int> cd_slot, cd_stamp;  // dense pixel -> slot map with a generation stamp (no per-call clearing)
    int cd_gen = 0;

    // structPoseEstimation (tracker_struct_pose.inc): packed input staging and the result block, both pinned host memory
    void* sp_stage_host = nullptr;   // pinned, read by the kernel directly
    size_t sp_cap_bytes = 0;
    StructIO* sp_io_host = nullptr;  // pinned, read and written by the kernel directly
};

static int chunks_for(const sdvgn_tracker* t, int n, int B) {
    // one point per lane while the launch is small; otherwise enough workgroups to cover the chip ~8x
    int c = (n + 255) / 256;
    if (c < 1) c = 1;
    const int want = (2048 + B - 1) / B;
    if (c > want) c = want < 1 ? 1 : want;
    if (c > t->max_chunks) c = t->max_chunks;
    return c;
}

static void fill_params(const sdvgn_tracker* t, int lvl, const double* pose7, double aff_a, double aff_b,
                        float cutoffTH, LevelParams& P) {
    // head of calcRes (:499-513) and calcGSSSE (:431-434)
    double R[9];
    gn::rotation_matrix(pose7, R);
    float Rf[9];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
    const float* Ki = t->Ki[lvl];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P.RKi[i * 3 + j] = (Rf[i * 3 + 0] * Ki[j] + Rf[i * 3 + 1] * Ki[3 + j]) + Rf[i * 3 + 2] * Ki[6 + j];
    for (int i = 0; i < 3; ++i) P.t[i] = (float)pose7[4 + i];
    for (int i = 0; i < 9; ++i) P.Ki[i] = Ki[i];
    P.fx = t->fx[lvl]; P.fy = t->fy[lvl]; P.cx = t->cx[lvl]; P.cy = t->cy[lvl];
    double ab[2];
    gn::aff_from_to(t->ref_exposure, t->new_exposure, t->ref_a, t->ref_b, aff_a, aff_b, ab);
    P.affLL0 = (float)ab[0]; P.affLL1 = (float)ab[1];
    P.b0 = (float)t->ref_b;
    P.cutoff = cutoffTH;
    P.huber = t->huberTH;
    P.maxEnergy = 2 * t->huberTH * cutoffTH - t->huberTH * t->huberTH;
    P.wl = t->w[lvl]; P.hl = t->h[lvl]; P.lvl = lvl; P.n = t->pc_n[lvl];
}

// launches k_res_gs + k_finalize for B problems; results in t->out_dev
// zero_copy (the host-driven single-trial path): the kernels read the 152-B LevelParams record straight from the pinned host
// buffer and k_finalize stores its 640 B of results straight into pinned host memory -- no copy engine in the loop (a small
// hipMemcpyAsync costs 10-20 us of fixed latency each way, the PCIe transfers themselves well under 1 us).
static int ensure_records(sdvgn_tracker* t) {
    if (t->rec_valid) return SDVGN_OK;
    for (int l = 0; l < t->levels; ++l) {
        const int npix = t->w[l] * t->h[l];
        if (!t->pyr_rec_dev[l]) HIPCHK(SDVGN_DMALLOC(&t->pyr_rec_dev[l], sizeof(float4) * 4 * (size_t)npix));
        k_pyr_to_records<<<(npix + 255) / 256, 256, 0, t->stream>>>(t->pyr_dev[l], t->pyr_rec_dev[l], t->w[l], t->h[l]);
    }
    HIPCHK(hipGetLastError());
    t->rec_valid = true;
    return SDVGN_OK;
}

static int launch_res_gs(sdvgn_tracker* t, int lvl, int B, const double* pose7, const double* aff, float cutoffTH,
                         bool write_terms, double* out_dev, bool zero_copy = false, const ProblemPtrs* ptrs = nullptr) {
    if (!t->haveK || !t->haveNew) return SDVGN_E_STATE;
    if (lvl < 0 || lvl >= t->levels || B < 1 || B > t->max_batch) return SDVGN_E_ARG;
    if (write_terms && B != 1) return SDVGN_E_ARG;
    for (int b = 0; b < B; ++b) fill_params(t, lvl, pose7 + 7 * b, aff[2 * b], aff[2 * b + 1], cutoffTH, t->params_host[b]);
    const LevelParams* params = zero_copy ? t->params_host : t->params_dev;
    if (!zero_copy) HIPCHK(hipMemcpyAsync(t->params_dev, t->params_host, sizeof(LevelParams) * B, hipMemcpyHostToDevice, t->stream));
    const int n = t->pc_n[lvl];
    const int chunks = chunks_for(t, n, B);
    dim3 grid(chunks, B), block(256);
    if (t->precision == PREC_F32_REC) {
        // fp32 on the gather-friendly record copy of the pyramid: bit-identical results, one 128-byte line per lookup
        if (write_terms) return SDVGN_E_ARG;
        int rcr = ensure_records(t);
        if (rcr) return rcr;
        const float* rimg = reinterpret_cast<const float*>(t->pyr_rec_dev[lvl]);
        k_res_gs<false, PREC_F32_REC><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], rimg, params, t->partial_dev, nullptr, nullptr, ptrs);
    } else if (t->precision != PREC_F32) {
        // tolerance study (configs[4]): same kernel on an fp16 pyramid / fp16 operands / fp16 accumulator
        if (write_terms) return SDVGN_E_ARG;
        if (!t->half_valid) {
            for (int l = 0; l < t->levels; ++l) {
                const int npix = t->w[l] * t->h[l];
                if (!t->pyr_half_dev[l]) HIPCHK(SDVGN_DMALLOC(&t->pyr_half_dev[l], sizeof(__half) * 4 * (size_t)npix));
                k_pyr_to_half<<<(npix + 255) / 256, 256, 0, t->stream>>>(t->pyr_dev[l], t->pyr_half_dev[l], npix);
            }
            t->half_valid = true;
        }
        const float* himg = reinterpret_cast<const float*>(t->pyr_half_dev[lvl]);
        if (t->precision == PREC_H_PYR) k_res_gs<false, PREC_H_PYR><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], himg, params, t->partial_dev, nullptr, nullptr);
        else if (t->precision == PREC_H_OPER) k_res_gs<false, PREC_H_OPER><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], himg, params, t->partial_dev, nullptr, nullptr);
        else k_res_gs<false, PREC_H_ACC><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], himg, params, t->partial_dev, nullptr, nullptr);
    } else if (write_terms) {
        if (t->arith == 0) k_res_gs<true><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], t->pyr_dev[lvl], params, t->partial_dev, t->terms_dev, t->status_dev);
        else k_res_gs<true, PREC_F32, 1><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], t->pyr_dev[lvl], params, t->partial_dev, t->terms_dev, t->status_dev);
        t->terms_lvl = lvl;
    } else {
        if (t->arith == 0) k_res_gs<false><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], t->pyr_dev[lvl], params, t->partial_dev, nullptr, nullptr, ptrs);
        else k_res_gs<false, PREC_F32, 1><<<grid, block, 0, t->stream>>>(t->pc_dev[lvl], t->pyr_dev[lvl], params, t->partial_dev, nullptr, nullptr, ptrs);
    }
    if (zero_copy && B == 1) k_finalize<<<B, 128, 0, t->stream>>>(t->partial_dev, chunks, out_dev, t->flag_host, ++t->flag_seq);
    else k_finalize<<<B, 128, 0, t->stream>>>(t->partial_dev, chunks, out_dev);
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

static int res_gs_sync(sdvgn_tracker* t, int lvl, const double* pose7, double a, double b, float cutoffTH,
                       bool write_terms, double* out6, double* H, double* bv) {
    const double aff[2] = {a, b};
    int rc = launch_res_gs(t, lvl, 1, pose7, aff, cutoffTH, write_terms, t->out_host, /*zero_copy=*/true);
    if (rc) return rc;
    HIPCHK(wait_flag(t->flag_host, t->flag_seq, t->stream));
    if (write_terms) HIPCHK(hipStreamSynchronize(t->stream));   // the per-point parity planes are read back by a later call
    if (out6) std::memcpy(out6, t->out_host, 6 * sizeof(double));
    if (H) std::memcpy(H, t->out_host + 6, 64 * sizeof(double));
    if (bv) std::memcpy(bv, t->out_host + 70, 8 * sizeof(double));
    return SDVGN_OK;
}

// ---- trackNewestCoarse, host-driven (CoarseTracker.cpp:662-838) ---------------------------------------
static int track_host_driven(sdvgn_tracker* t, double* pose7_io, double* aff_io, int coarsestLvl, const double* minRes,
                             double* lastResiduals, double* lastFlow) {
    if (coarsestLvl < 0 || coarsestLvl >= 5 || coarsestLvl >= t->levels) return SDVGN_E_ARG;  // assert :672
    for (int i = 0; i < 5; ++i) lastResiduals[i] = NAN;
    for (int i = 0; i < 3; ++i) lastFlow[i] = 1000;
    t->trace.clear();
    const int maxIterations[] = {10, 20, 50, 50, 50};
    const float lambdaExtrapolationLimit = 0.001f;
    gn::Pose cur;
    gn::pose_load(cur, pose7_io);
    double aff_a = aff_io[0], aff_b = aff_io[1];
    bool haveRepeated = false;
    double p7[7];

    for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
        double H[64], b[8], resOld[6];
        float levelCutoffRepeat = 1;
        gn::pose_store(cur, p7);
        int rc = res_gs_sync(t, lvl, p7, aff_a, aff_b, t->coarseCutoffTH * levelCutoffRepeat, false, resOld, H, b);
        if (rc) return rc;
        while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
            levelCutoffRepeat *= 2;
            rc = res_gs_sync(t, lvl, p7, aff_a, aff_b, t->coarseCutoffTH * levelCutoffRepeat, false, resOld, H, b);
            if (rc) return rc;
        }
        float lambda = 0.01f;
        for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
            double inc[8], incScaled[8];
            lm_step(H, b, lambda, lambdaExtrapolationLimit, t->affineOptModeA, t->affineOptModeB, inc, incScaled);
            const gn::Pose cand = gn::compose(gn::exp_se3(incScaled), cur);
            const double a_new = aff_a + incScaled[6], b_new = aff_b + incScaled[7];
            double resNew[6], Hn[64], bn[8];
            gn::pose_store(cand, p7);
            rc = res_gs_sync(t, lvl, p7, a_new, b_new, t->coarseCutoffTH * levelCutoffRepeat, false, resNew, Hn, bn);
            if (rc) return rc;
            const bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
            {
                double row[15] = {(double)lvl, (double)iteration, (double)lambda, accept ? 1.0 : 0.0};
                for (int i = 0; i < 8; ++i) row[4 + i] = incScaled[i];
                row[12] = resNew[0]; row[13] = resNew[1]; row[14] = levelCutoffRepeat;
                t->trace.insert(t->trace.end(), row, row + 15);
            }
            if (accept) {
                std::memcpy(H, Hn, sizeof(H));
                std::memcpy(b, bn, sizeof(b));
                std::memcpy(resOld, resNew, sizeof(resOld));
                aff_a = a_new; aff_b = b_new;
                cur = cand;
                lambda *= 0.5;
            } else {
                lambda *= 4;
                if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
            }
            double nrm = 0;
            for (int i = 0; i < 8; ++i) nrm += inc[i] * inc[i];
            if (!(sqrt(nrm) > 1e-3)) break;
        }
        lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
        for (int i = 0; i < 3; ++i) lastFlow[i] = resOld[2 + i];
        if (lastResiduals[lvl] > 1.5 * minRes[lvl]) return 0;
        if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    }
    gn::pose_store(cur, pose7_io);
    aff_io[0] = aff_a; aff_io[1] = aff_b;
    return track_final_checks(t->affineOptModeA, t->affineOptModeB, t->ref_exposure, t->new_exposure, t->ref_a, t->ref_b,
                              aff_io) ? 1 : 0;
}

// =========================================== C ABI =====================================================
extern "C" {

const char* sdvgn_version(void) { return "sdvgn 0.1 (gfx950)"; }

const char* sdvgn_error_string(int code) {
    switch (code) {
        case SDVGN_OK: return "ok";
        case SDVGN_E_ARG: return "bad argument";
        case SDVGN_E_STATE: return "call order violated";
        case SDVGN_E_NODEVICE: return "no usable HIP device";
        default: return code < 0 ? hipGetErrorString((hipError_t)(-code)) : "unknown";
    }
}

int sdvgn_tracker_create(sdvgn_tracker** out, int device, int w0, int h0, int levels, int max_points, int max_batch,
                         void* stream) {
    if (!out || levels < 1 || levels > SDVGN_MAX_LEVELS || w0 < 16 || h0 < 16 || max_points < 1 || max_batch < 1)
        return SDVGN_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return SDVGN_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    sdvgn_tracker* t = new (std::nothrow) sdvgn_tracker();
    if (!t) return SDVGN_E_ARG;
    t->device = device; t->levels = levels; t->max_points = max_points; t->max_batch = max_batch;
    t->max_chunks = 256;
    for (int l = 0; l < levels; ++l) { t->w[l] = w0 >> l; t->h[l] = h0 >> l; }
    if (stream) t->stream = (hipStream_t)stream;
    else {
        // trackers without a caller's stream share one library stream per device (not the back end's: a frame is tracked while a window is
        // optimised).  One stream per handle would mean one hardware queue per handle, and the first launch on a queue that has been idle
        // pays its re-activation (measured 0.2-0.5 ms on the back-end handles, backend.hip)
        static std::mutex mu;
        static hipStream_t shared[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        if (device < 0 || device >= 64) { delete t; return SDVGN_E_ARG; }
        if (!shared[device]) HIPCHK(hipStreamCreateWithFlags(&shared[device], hipStreamNonBlocking));
        t->stream = shared[device];
    }
    for (int l = 0; l < levels; ++l) {
        HIPCHK(SDVGN_DMALLOC(&t->pc_dev[l], sizeof(float4) * max_points));
        HIPCHK(SDVGN_DMALLOC(&t->pyr_dev[l], sizeof(float) * 3 * (size_t)t->w[l] * t->h[l]));
        HIPCHK(hipMemsetAsync(t->pyr_dev[l], 0, sizeof(float) * 3 * (size_t)t->w[l] * t->h[l], t->stream));
    }
    HIPCHK(SDVGN_DMALLOC(&t->img_stage_dev, sizeof(float) * (size_t)w0 * h0));
    HIPCHK(SDVGN_DMALLOC(&t->params_dev, sizeof(LevelParams) * max_batch));
    HIPCHK(SDVGN_HMALLOC(&t->params_host, sizeof(LevelParams) * max_batch));
    HIPCHK(SDVGN_DMALLOC(&t->partial_dev, sizeof(float) * kNRed * (size_t)t->max_chunks * max_batch));
    HIPCHK(SDVGN_DMALLOC(&t->out_dev, sizeof(double) * kOutStride * max_batch));
    HIPCHK(SDVGN_HMALLOC(&t->out_host, sizeof(double) * kOutStride * max_batch));
    HIPCHK(SDVGN_HMALLOC((void**)&t->flag_host, 64));
    *t->flag_host = 0;
    HIPCHK(SDVGN_DMALLOC(&t->terms_dev, sizeof(float) * 8 * (size_t)max_points));
    HIPCHK(SDVGN_DMALLOC(&t->status_dev, sizeof(int) * (size_t)max_points));
    HIPCHK(SDVGN_HMALLOC(&t->track_host, sizeof(TrackState) * max_batch));
    std::memset(t->track_host, 0, sizeof(TrackState) * max_batch);
    HIPCHK(hipStreamSynchronize(t->stream));
    *out = t;
    return SDVGN_OK;
}

void sdvgn_tracker_destroy(sdvgn_tracker* t) {
    if (!t) return;
    hipSetDevice(t->device);
    hipStreamSynchronize(t->stream);
    const bool dbg = getenv("SDVGN_PROFILE") != nullptr;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && dbg) fprintf(stderr, "[sdvgn] tracker destroy: %s -> %d (%s)\n", what, (int)e, hipGetErrorString(e)); };
#define DFREE(p) chk(SDVGN_DFREE(p), "SDVGN_DFREE(" #p ")")
#define HFREE(p) chk(SDVGN_HFREE(p), "SDVGN_HFREE(" #p ")")
    for (int l = 0; l < t->levels; ++l) { DFREE(t->pc_dev[l]); DFREE(t->pyr_dev[l]); if (t->pyr_half_dev[l]) DFREE(t->pyr_half_dev[l]); if (t->pyr_rec_dev[l]) DFREE(t->pyr_rec_dev[l]); }
    DFREE(t->img_stage_dev); DFREE(t->params_dev); DFREE(t->ptrs_dev); HFREE(t->params_host); DFREE(t->partial_dev);
    DFREE(t->out_dev); HFREE(t->out_host); HFREE(t->flag_host); DFREE(t->terms_dev); DFREE(t->status_dev);
    HFREE(t->track_host); DFREE(t->team_dev);
    HFREE(t->sp_stage_host); HFREE(t->sp_io_host);
    DFREE(t->tp_static_dev); HFREE(t->tp_state_host);
    DFREE(t->cd_maps); DFREE(t->cd_rows); HFREE(t->cd_n_host); HFREE(t->cd_stage);
#undef DFREE
#undef HFREE
    if (t->own_stream) chk(hipStreamDestroy(t->stream), "hipStreamDestroy");
    (void)hipGetLastError();   // a failed free must not surface as the "last error" of some later, unrelated launch
    delete t;
}

void* sdvgn_tracker_stream(sdvgn_tracker* t) { return t ? (void*)t->stream : nullptr; }
const float* sdvgn_tracker_pyr_dev(sdvgn_tracker* t, int lvl) {
    if (!t || lvl < 0 || lvl >= t->levels || !t->haveNew) return nullptr;
    // the consumer runs on another handle's stream: make sure the pyramid kernels of this handle have finished
    if (hipSetDevice(t->device) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess) return nullptr;
    return t->pyr_dev[lvl];
}

int sdvgn_tracker_set_settings(sdvgn_tracker* t, float huberTH, float coarseCutoffTH, float affA, float affB) {
    if (!t) return SDVGN_E_ARG;
    t->huberTH = huberTH; t->coarseCutoffTH = coarseCutoffTH; t->affineOptModeA = affA; t->affineOptModeB = affB;
    return SDVGN_OK;
}

int sdvgn_tracker_set_team(sdvgn_tracker* t, int team) {
    if (!t || team < -1 || team > kTeamMax) return SDVGN_E_ARG;
    t->team_mode = team;
    return SDVGN_OK;
}
int sdvgn_tracker_get_team(sdvgn_tracker* t) { return t ? t->last_team : SDVGN_E_ARG; }
int sdvgn_tracker_get_team_fallbacks(sdvgn_tracker* t) { return t ? t->team_fallbacks : SDVGN_E_ARG; }

int sdvgn_tracker_set_arith(sdvgn_tracker* t, int mode) {
    if (!t || mode < 0 || mode > 1) return SDVGN_E_ARG;
    t->arith = mode;
    return SDVGN_OK;
}

int sdvgn_tracker_set_precision(sdvgn_tracker* t, int mode) {
    if (!t || mode < 0 || mode > 4) return SDVGN_E_ARG;
    t->precision = mode == 4 ? PREC_F32_REC : mode;
    return SDVGN_OK;
}
const void* sdvgn_tracker_records_dev(sdvgn_tracker* t, int lvl) {
    if (!t || lvl < 0 || lvl >= t->levels || !t->haveNew) return nullptr;
    if (hipSetDevice(t->device) != hipSuccess || ensure_records(t) != SDVGN_OK || hipStreamSynchronize(t->stream) != hipSuccess) return nullptr;
    return t->pyr_rec_dev[lvl];
}

int sdvgn_tracker_make_K(sdvgn_tracker* t, float fx, float fy, float cx, float cy) {  // CoarseTracker.cpp:77-106
    if (!t) return SDVGN_E_ARG;
    t->fx[0] = fx; t->fy[0] = fy; t->cx[0] = cx; t->cy[0] = cy;
    for (int l = 1; l < t->levels; ++l) {
        t->fx[l] = t->fx[l - 1] * 0.5;
        t->fy[l] = t->fy[l - 1] * 0.5;
        t->cx[l] = (t->cx[0] + 0.5) / ((int)1 << l) - 0.5;
        t->cy[l] = (t->cy[0] + 0.5) / ((int)1 << l) - 0.5;
    }
    for (int l = 0; l < t->levels; ++l) {
        const float K[9] = {t->fx[l], 0, t->cx[l], 0, t->fy[l], t->cy[l], 0, 0, 1};
        gn::inverse3f(K, t->Ki[l]);
    }
    t->haveK = true;
    return SDVGN_OK;
}

int sdvgn_tracker_get_K(sdvgn_tracker* t, int lvl, float k4[4], float Ki9[9]) {
    if (!t || lvl < 0 || lvl >= t->levels || !t->haveK) return SDVGN_E_ARG;
    k4[0] = t->fx[lvl]; k4[1] = t->fy[lvl]; k4[2] = t->cx[lvl]; k4[3] = t->cy[lvl];
    std::memcpy(Ki9, t->Ki[lvl], sizeof(float) * 9);
    return SDVGN_OK;
}

int sdvgn_tracker_set_ref(sdvgn_tracker* t, int lvl, int n, const float* u, const float* v, const float* idepth,
                          const float* color) {
    if (!t || lvl < 0 || lvl >= t->levels || n < 0 || n > t->max_points) return SDVGN_E_ARG;
    if (n > 0 && (!u || !v || !idepth || !color)) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    std::vector<float4> packed(n);
    for (int i = 0; i < n; ++i) packed[i] = make_float4(u[i], v[i], idepth[i], color[i]);
    if (n) HIPCHK(hipMemcpyAsync(t->pc_dev[lvl], packed.data(), sizeof(float4) * n, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    t->pc_n[lvl] = n;
    return SDVGN_OK;
}

int sdvgn_tracker_set_ref_frame(sdvgn_tracker* t, float exposure, double a, double b) {
    if (!t) return SDVGN_E_ARG;
    t->ref_exposure = exposure; t->ref_a = a; t->ref_b = b;
    return SDVGN_OK;
}

static int build_pyramid(sdvgn_tracker* t, const float* img_dev) {
    for (int l = 0; l < t->levels; ++l) {
        const int wl = t->w[l], hl = t->h[l];
        const int qw = (wl + 1) >> 1, qh = (hl + 1) >> 1;
        dim3 grid((qw + 255) / 256, qh), block(256);
        const int has_next = (l + 1 < t->levels) ? 1 : 0;
        k_pyr_level<<<grid, block, 0, t->stream>>>(l == 0 ? img_dev : nullptr, t->pyr_dev[l],
                                                   has_next ? t->pyr_dev[l + 1] : nullptr, wl, hl, has_next);
    }
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

int sdvgn_tracker_set_new_image(sdvgn_tracker* t, const float* image, float exposure) {
    if (!t || !image) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipMemcpyAsync(t->img_stage_dev, image, sizeof(float) * (size_t)t->w[0] * t->h[0], hipMemcpyHostToDevice, t->stream));
    int rc = build_pyramid(t, t->img_stage_dev);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(t->stream));  // `image` may be pageable: do not return before the copy is done
    t->new_exposure = exposure; t->haveNew = true; t->half_valid = false; t->rec_valid = false;
    return SDVGN_OK;
}

int sdvgn_tracker_set_new_image_dev(sdvgn_tracker* t, const float* image_dev, float exposure) {
    if (!t || !image_dev) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    int rc = build_pyramid(t, image_dev);
    if (rc) return rc;
    t->new_exposure = exposure; t->haveNew = true; t->half_valid = false; t->rec_valid = false;
    return SDVGN_OK;
}

int sdvgn_tracker_set_new_pyr(sdvgn_tracker* t, int lvl, const float* aos3, float exposure) {
    if (!t || !aos3 || lvl < 0 || lvl >= t->levels) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipMemcpyAsync(t->pyr_dev[lvl], aos3, sizeof(float) * 3 * (size_t)t->w[lvl] * t->h[lvl], hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    t->new_exposure = exposure; t->haveNew = true; t->half_valid = false; t->rec_valid = false;
    return SDVGN_OK;
}

int sdvgn_tracker_get_pyr(sdvgn_tracker* t, int lvl, float* out) {
    if (!t || !out || lvl < 0 || lvl >= t->levels) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipMemcpyAsync(out, t->pyr_dev[lvl], sizeof(float) * 3 * (size_t)t->w[lvl] * t->h[lvl], hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return SDVGN_OK;
}

int sdvgn_tracker_calc_res(sdvgn_tracker* t, int lvl, const double pose7[7], double a, double b, float cutoffTH, double out6[6]) {
    if (!t || !pose7 || !out6) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    return res_gs_sync(t, lvl, pose7, a, b, cutoffTH, true, out6, nullptr, nullptr);
}

int sdvgn_tracker_calc_gs(sdvgn_tracker* t, int lvl, const double pose7[7], double a, double b, float cutoffTH, double H[64], double b8[8]) {
    if (!t || !pose7 || !H || !b8) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    return res_gs_sync(t, lvl, pose7, a, b, cutoffTH, false, nullptr, H, b8);
}

int sdvgn_tracker_res_and_gs(sdvgn_tracker* t, int lvl, const double pose7[7], double a, double b, float cutoffTH,
                             double out6[6], double H[64], double b8[8]) {
    if (!t || !pose7) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    return res_gs_sync(t, lvl, pose7, a, b, cutoffTH, false, out6, H, b8);
}

int sdvgn_tracker_res_and_gs_batch(sdvgn_tracker* t, int lvl, int B, const double* pose7, const double* aff, float cutoffTH,
                                   double* out_dev) {
    if (!t || !pose7 || !aff) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    return launch_res_gs(t, lvl, B, pose7, aff, cutoffTH, false, out_dev ? out_dev : t->out_dev);
}

int sdvgn_tracker_res_and_gs_multi(sdvgn_tracker* t, int lvl, int B, const void* const* pc_dev, const void* const* img_dev, const double* pose7,
                                   const double* aff, float cutoffTH, double* out_dev) {
    if (!t || !pose7 || !aff || !pc_dev || !img_dev || B < 1 || B > t->max_batch) return SDVGN_E_ARG;
    if (t->precision != PREC_F32 && t->precision != PREC_F32_REC) return SDVGN_E_STATE;   // (record mode: img_dev[b] = sdvgn_tracker_records_dev of problem b)
    HIPCHK(hipSetDevice(t->device));
    if (!t->ptrs_dev) HIPCHK(SDVGN_DMALLOC(&t->ptrs_dev, sizeof(ProblemPtrs) * t->max_batch));
    std::vector<ProblemPtrs> hp(B);
    for (int b = 0; b < B; ++b) { hp[b].pc = (const float4*)pc_dev[b]; hp[b].img = (const float*)img_dev[b]; if (!hp[b].pc || !hp[b].img) return SDVGN_E_ARG; }
    HIPCHK(hipMemcpyAsync(t->ptrs_dev, hp.data(), sizeof(ProblemPtrs) * B, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));   // hp goes out of scope
    return launch_res_gs(t, lvl, B, pose7, aff, cutoffTH, false, out_dev ? out_dev : t->out_dev, false, t->ptrs_dev);
}

const void* sdvgn_tracker_ref_dev(sdvgn_tracker* t, int lvl) {
    if (!t || lvl < 0 || lvl >= t->levels) return nullptr;
    if (hipSetDevice(t->device) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess) return nullptr;
    return t->pc_dev[lvl];
}

int sdvgn_tracker_get_point_terms(sdvgn_tracker* t, int lvl, float* terms, int* status) {
    if (!t || !terms || !status || lvl != t->terms_lvl) return SDVGN_E_STATE;
    HIPCHK(hipSetDevice(t->device));
    const int n = t->pc_n[lvl];
    HIPCHK(hipMemcpyAsync(terms, t->terms_dev, sizeof(float) * 8 * (size_t)n, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipMemcpyAsync(status, t->status_dev, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return SDVGN_OK;
}

int sdvgn_tracker_track(sdvgn_tracker* t, double pose7_io[7], double aff_io[2], int coarsestLvl, const double minRes[5],
                        double lastResiduals[5], double lastFlow[3]) {
    if (!t || !pose7_io || !aff_io || !minRes || !lastResiduals || !lastFlow) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    return track_host_driven(t, pose7_io, aff_io, coarsestLvl, minRes, lastResiduals, lastFlow);
}

int sdvgn_tracker_get_trace(sdvgn_tracker* t, double* rows, int cap) {
    if (!t) return SDVGN_E_ARG;
    const int n = (int)(t->trace.size() / 15);
    const int m = n < cap ? n : cap;
    if (rows && m > 0) std::memcpy(rows, t->trace.data(), sizeof(double) * 15 * m);
    return n;
}

int sdvgn_tracker_track_batch(sdvgn_tracker* t, int B, double* pose7_io, double* aff_io, int coarsestLvl,
                              const double* minRes, double* lastResiduals, double* lastFlow, int* ok) {
    if (!t || !pose7_io || !aff_io || !lastResiduals || !lastFlow || !ok || B < 1 || B > t->max_batch) return SDVGN_E_ARG;
    if (coarsestLvl < 0 || coarsestLvl >= 5 || coarsestLvl >= t->levels) return SDVGN_E_ARG;
    if (!t->haveK || !t->haveNew) return SDVGN_E_STATE;
    HIPCHK(hipSetDevice(t->device));
    TrackConst tc;
    tc.levels = t->levels;
    for (int l = 0; l < t->levels; ++l) {
        tc.w[l] = t->w[l]; tc.h[l] = t->h[l]; tc.fx[l] = t->fx[l]; tc.fy[l] = t->fy[l]; tc.cx[l] = t->cx[l]; tc.cy[l] = t->cy[l];
        std::memcpy(tc.Ki[l], t->Ki[l], sizeof(float) * 9);
        tc.pc[l] = t->pc_dev[l]; tc.pc_n[l] = t->pc_n[l]; tc.img[l] = t->pyr_dev[l];
    }
    tc.huberTH = t->huberTH; tc.coarseCutoffTH = t->coarseCutoffTH; tc.affA = t->affineOptModeA; tc.affB = t->affineOptModeB;
    tc.ref_exposure = t->ref_exposure; tc.new_exposure = t->new_exposure; tc.ref_a = t->ref_a; tc.ref_b = t->ref_b;
    tc.coarsestLvl = coarsestLvl;
    for (int b = 0; b < B; ++b) {
        TrackState& s = t->track_host[b];
        std::memcpy(s.pose, pose7_io + 7 * b, sizeof(double) * 7);
        s.aff[0] = aff_io[2 * b]; s.aff[1] = aff_io[2 * b + 1];
        for (int i = 0; i < 5; ++i) s.minRes[i] = minRes ? minRes[5 * b + i] : NAN;
        s.ok = 0; s.ntrials = 0;
    }
    // no copy-engine transfers: the constants are a kernel argument, the per-hypothesis state blocks stay in pinned host memory.
    // Team size: enough workgroups of 256 lanes for one pass over the largest level (at most kTeamMax), as long as every workgroup of
    // the launch is resident at once (two 256-lane workgroups per CU at its register count: 512 on the 256 CUs of an MI355X); beyond that: one
    // workgroup per hypothesis.
    const int seq = ++t->track_seq;
    int T = 0;
    {
        int nmax = 0;
        for (int l = 0; l <= coarsestLvl; ++l) nmax = t->pc_n[l] > nmax ? t->pc_n[l] : nmax;
        const int Bpad = (B + 7) & ~7;
        int want = t->team_mode > 0 ? t->team_mode : (nmax + kTeamThreads - 1) / kTeamThreads;
        if (want > kTeamMax) want = kTeamMax;
        // resident capacity: what the device can hold of THIS kernel at once (occupancy query: workgroups per CU at its register / LDS
        // use x CUs), less a quarter as headroom for whatever runs beside it -- the back end has a stream of its own precisely so that a
        // window is optimised while a frame is tracked, and a caller may run several handles or ranks on one device.  The whole grid must
        // fit, and so must one dispatch window of 8 interleaved teams (a partitioned device with few CUs gets small teams or the
        // one-workgroup kernel).  If members still fail to meet (the poll gives up, ok == -2) the batch is re-run on k_track below.
        if (t->cu_count <= 0) {
            hipDeviceProp_t prop;
            t->cu_count = (hipGetDeviceProperties(&prop, t->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_track_team, kTeamThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            t->team_capacity = per_cu * t->cu_count;
        }
        // (an explicit team size -- sdvgn_tracker_set_team(1..32), tests -- may use the whole resident capacity; the automatic choice leaves a quarter)
        const int capacity = t->team_mode > 0 ? t->team_capacity : (3 * t->team_capacity) / 4;
        if (want > capacity / Bpad) want = capacity / Bpad;
        if (Bpad > 512) want = 0;      // the exchange rows are allocated for at most 512 hypotheses
        // a fixed request of 1 runs the team kernel with a single member (tests); automatic mode needs at least two
        if (t->team_mode >= 0 && (want >= 2 || (t->team_mode > 0 && want >= 1))) T = want;
    }
    t->last_team = T;
    if (T >= 1) {
        if (t->team_cap < B) {
            HIPCHK(hipStreamSynchronize(t->stream));
            if (t->team_dev) HIPCHK(SDVGN_DFREE(t->team_dev));
            t->team_dev = nullptr; t->team_cap = 0;
            const int cap = t->max_batch < 512 ? t->max_batch : 512;   // at most 512 resident workgroups
            HIPCHK(SDVGN_DMALLOC(&t->team_dev, sizeof(TeamMem) * (size_t)cap));
            HIPCHK(hipMemsetAsync(t->team_dev, 0, sizeof(TeamMem) * (size_t)cap, t->stream));
            t->team_cap = cap;
        }
        const int Bpad = (B + 7) & ~7;
        if (++t->team_seq >= (1u << 20)) {   // tags would repeat: start over on zeroed rows
            HIPCHK(hipMemsetAsync(t->team_dev, 0, sizeof(TeamMem) * (size_t)t->team_cap, t->stream));
            t->team_seq = 1;
        }
        k_track_team<<<Bpad * T, kTeamThreads, 0, t->stream>>>(tc, t->track_host, t->team_dev, B, T, t->team_seq << 12, seq);
    } else {
        k_track<<<B, kTrackThreads, 0, t->stream>>>(tc, t->track_host, seq);
    }
    HIPCHK(hipGetLastError());
    // every hypothesis' posting workgroup publishes `seq` behind its results: spin on the flags instead of a stream synchronisation
    for (int b = 0; b < B; ++b) HIPCHK(wait_flag(&t->track_host[b].done, seq, t->stream));
    if (T >= 1) {
        bool dead = false;
        for (int b = 0; b < B; ++b) dead = dead || t->track_host[b].ok == -2;
        if (dead) {
            // an exchange gave up: some member of a team was not resident while the others polled (the device was busier than the headroom
            // allows for).  Nothing of the batch is used; it runs again on the one-workgroup kernel, which needs no co-residency.
            ++t->team_fallbacks;
            HIPCHK(hipStreamSynchronize(t->stream));
            HIPCHK(hipMemsetAsync(t->team_dev, 0, sizeof(TeamMem) * (size_t)t->team_cap, t->stream));   // half-written exchange rows
            for (int b = 0; b < B; ++b) {
                TrackState& s = t->track_host[b];
                std::memcpy(s.pose, pose7_io + 7 * b, sizeof(double) * 7);
                s.aff[0] = aff_io[2 * b]; s.aff[1] = aff_io[2 * b + 1];
                for (int i = 0; i < 5; ++i) s.minRes[i] = minRes ? minRes[5 * b + i] : NAN;
                s.ok = 0; s.ntrials = 0;
            }
            const int seq2 = ++t->track_seq;
            t->last_team = 0;
            k_track<<<B, kTrackThreads, 0, t->stream>>>(tc, t->track_host, seq2);
            HIPCHK(hipGetLastError());
            for (int b = 0; b < B; ++b) HIPCHK(wait_flag(&t->track_host[b].done, seq2, t->stream));
        }
    }
    for (int b = 0; b < B; ++b) {
        const TrackState& s = t->track_host[b];
        std::memcpy(pose7_io + 7 * b, s.pose, sizeof(double) * 7);
        aff_io[2 * b] = s.aff[0]; aff_io[2 * b + 1] = s.aff[1];
        for (int i = 0; i < 5; ++i) lastResiduals[5 * b + i] = s.lastRes[i];
        for (int i = 0; i < 3; ++i) lastFlow[3 * b + i] = s.flow[i];
        ok[b] = s.ok;
    }
    if (getenv("SDVGN_PROFILE")) {
        const TrackState& s = t->track_host[0];
        fprintf(stderr, "[sdvgn profile] team %d (its stamps: 10 ns ticks); ", T);
        fprintf(stderr, "[sdvgn profile] k_track hyp0: state machine + solves %lld, %lld solves, evaluations %lld over %lld evals, %d trials\n", s.dbg_cycles[0], s.dbg_cycles[1],
                s.dbg_cycles[2], s.dbg_cycles[3], s.ntrials);
    }
    return SDVGN_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// structPoseEstimation
// ---------------------------------------------------------------------------------------------------------------
static int struct_pose_run(sdvgn_tracker* t, int mode, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                           int n_hosts, const double* host_pose7, const double* obs, const double* pose7_in) {
    if (!t || !t->haveK) return t ? SDVGN_E_STATE : SDVGN_E_ARG;
    if (n < 0 || n > kStructMaxN || n_hosts < 1 || !host_pose7 || !pose7_in) return SDVGN_E_ARG;
    if (n > 0 && (!u || !v || !idepth || !host_idx || !obs)) return SDVGN_E_ARG;
    for (int i = 0; i < n; ++i) if (host_idx[i] < 0 || host_idx[i] >= n_hosts) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    // packed staging: u | v | idepth | host_idx | obs (float2) | hostRt (12 floats per host)
    const size_t np = (size_t)((n + 3) & ~3);
    const size_t bytes = np * 4 * 6 + (size_t)n_hosts * 12 * 4;
    if (bytes > t->sp_cap_bytes) {
        HIPCHK(hipStreamSynchronize(t->stream));
        SDVGN_HFREE(t->sp_stage_host);
        t->sp_stage_host = nullptr; t->sp_cap_bytes = 0;
        const size_t cap = bytes * 2 + 4096;
        HIPCHK(SDVGN_HMALLOC(&t->sp_stage_host, cap));
        t->sp_cap_bytes = cap;
    }
    if (!t->sp_io_host) HIPCHK(SDVGN_HMALLOC((void**)&t->sp_io_host, sizeof(StructIO)));
    float* hs = (float*)t->sp_stage_host;
    float* hu = hs, *hv = hs + np, *hid = hs + 2 * np;
    int* hh = (int*)(hs + 3 * np);
    float* hobs = hs + 4 * np;
    float* hRt = hs + 6 * np;
    for (int i = 0; i < n; ++i) {
        hu[i] = u[i]; hv[i] = v[i]; hid[i] = idepth[i]; hh[i] = host_idx[i];
        hobs[2 * i] = (float)obs[2 * i]; hobs[2 * i + 1] = (float)obs[2 * i + 1];      // `it->second.cast<float>()`
    }
    for (int k = 0; k < n_hosts; ++k) {   // camToWorld.rotationMatrix().cast<float>(), translation().cast<float>() (:850-851)
        double R[9];
        gn::rotation_matrix(host_pose7 + 7 * k, R);
        for (int i = 0; i < 9; ++i) hRt[12 * k + i] = (float)R[i];
        for (int i = 0; i < 3; ++i) hRt[12 * k + 9 + i] = (float)host_pose7[7 * k + 4 + i];
    }
    std::memset(t->sp_io_host, 0, sizeof(StructIO));
    for (int i = 0; i < 7; ++i) t->sp_io_host->pose[i] = pose7_in[i];
    const int seq = ++t->track_seq;
    // zero-copy: the kernel reads the 29 kB of packed inputs once, coalesced, straight from pinned host memory and keeps its
    // result block there too -- two copy-engine round trips (10-20 us each) would cost more than the whole kernel's arithmetic
    StructConst C;
    C.fx = t->fx[0]; C.fy = t->fy[0]; C.cx = t->cx[0]; C.cy = t->cy[0];
    C.fxi = t->Ki[0][0]; C.fyi = t->Ki[0][4];                    // fxi[0] = Ki[0](0,0) (:101-102)
    C.wM3G = (float)(t->w[0] - 3); C.hM3G = (float)(t->h[0] - 3); // globalCalib.cpp:46-47
    C.n = n; C.n_hosts = n_hosts; C.mode = mode;
    const float* ds = (const float*)t->sp_stage_host;
    if (n <= kStructSmallN)
        k_struct_pose<kStructSmallThreads, kStructSmallPPL><<<1, kStructSmallThreads, 0, t->stream>>>(C, ds, ds + np, ds + 2 * np, (const int*)(ds + 3 * np), ds + 6 * np,
                                                                                                      (const float2*)(ds + 4 * np), t->sp_io_host, seq);
    else
        k_struct_pose<kStructThreads, kStructPPL><<<1, kStructThreads, 0, t->stream>>>(C, ds, ds + np, ds + 2 * np, (const int*)(ds + 3 * np), ds + 6 * np,
                                                                                       (const float2*)(ds + 4 * np), t->sp_io_host, seq);
    HIPCHK(hipGetLastError());
    HIPCHK(wait_flag(&t->sp_io_host->done, seq, t->stream));   // published behind the results: no stream synchronisation
    return SDVGN_OK;
}

int sdvgn_tracker_struct_pose(sdvgn_tracker* t, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                              int n_hosts, const double* host_pose7, const double* obs, double* curToWorld7, double* trace,
                              double* final_res) {
    if (!curToWorld7) return SDVGN_E_ARG;
    const int rc = struct_pose_run(t, 0, n, u, v, idepth, host_idx, n_hosts, host_pose7, obs, curToWorld7);
    if (rc < 0) return rc;
    const StructIO* io = t->sp_io_host;
    for (int i = 0; i < 7; ++i) curToWorld7[i] = io->pose[i];
    if (trace) std::memcpy(trace, io->trace, sizeof(double) * kStructTraceStride * (size_t)io->its);
    if (final_res) *final_res = io->final_res;
    if (getenv("SDVGN_PROFILE"))
        fprintf(stderr, "[sdvgn] k_struct_pose cycles: pass %lld reduce %lld logicA %lld solve %lld logicB %lld total %lld (its %d)\n", io->dbg_cycles[0],
                io->dbg_cycles[1], io->dbg_cycles[2], io->dbg_cycles[3], io->dbg_cycles[4], io->dbg_cycles[5], io->its);
    return io->its;
}

int sdvgn_tracker_struct_res_hb(sdvgn_tracker* t, int n, const float* u, const float* v, const float* idepth, const int* host_idx,
                                int n_hosts, const double* host_pose7, const double* obs, const double* worldToCur7, double* H36,
                                double* b6, double* energy, int* num) {
    if (!H36 || !b6 || !energy || !num) return SDVGN_E_ARG;
    const int rc = struct_pose_run(t, 1, n, u, v, idepth, host_idx, n_hosts, host_pose7, obs, worldToCur7);
    if (rc < 0) return rc;
    const StructIO* io = t->sp_io_host;
    std::memcpy(H36, io->H, sizeof(double) * 36);
    std::memcpy(b6, io->b, sizeof(double) * 6);
    *energy = io->energy; *num = io->num;
    return SDVGN_OK;
}

int sdvgn_struct_trace_stride(void) { return kStructTraceStride; }

// ---------------------------------------------------------------------------------------------------------------
// ImmaturePoint::traceOn for all immature points (first part)
// ---------------------------------------------------------------------------------------------------------------
int sdvgn_tracker_trace_set_points(sdvgn_tracker* t, int n, const float* u, const float* v, const float* energyTH, const float* gradH4,
                                   const float* color8, const float* weights8, const int* host_idx) {
    if (!t || n < 0) return SDVGN_E_ARG;
    if (n > 0 && (!u || !v || !energyTH || !gradH4 || !color8 || !weights8 || !host_idx)) return SDVGN_E_ARG;
    for (int i = 0; i < n; ++i) if (host_idx[i] < 0 || host_idx[i] >= kTraceMaxHosts) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    if ((int)np > t->tp_cap) {
        HIPCHK(hipStreamSynchronize(t->stream));
        SDVGN_DFREE(t->tp_static_dev); SDVGN_HFREE(t->tp_state_host);
        t->tp_static_dev = nullptr; t->tp_state_host = nullptr; t->tp_cap = 0;
        const size_t cap = np + np / 2 + 1024;
        HIPCHK(SDVGN_DMALLOC((void**)&t->tp_static_dev, sizeof(float) * 24 * cap));
        HIPCHK(SDVGN_HMALLOC(&t->tp_state_host, sizeof(float) * 8 * cap));
        t->tp_cap = (int)cap;
    }
    t->tp_n = n;
    if (n == 0) return SDVGN_OK;
    const size_t cap = t->tp_cap;
    std::vector<float> st(24 * cap, 0.f);
    float* su = st.data(), *sv = su + cap, *se = sv + cap, *sg = se + cap, *sc = sg + 4 * cap, *sw = sc + 8 * cap;
    int* sh = (int*)(sw + 8 * cap);
    std::memcpy(su, u, 4 * (size_t)n); std::memcpy(sv, v, 4 * (size_t)n); std::memcpy(se, energyTH, 4 * (size_t)n);
    std::memcpy(sg, gradH4, 16 * (size_t)n); std::memcpy(sc, color8, 32 * (size_t)n); std::memcpy(sw, weights8, 32 * (size_t)n);
    std::memcpy(sh, host_idx, 4 * (size_t)n);
    HIPCHK(hipMemcpyAsync(t->tp_static_dev, st.data(), sizeof(float) * 24 * cap, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    return SDVGN_OK;
}

int sdvgn_tracker_trace_points(sdvgn_tracker* t, int n_hosts, const float* KRKi9, const float* Kt3, const float* aff2, float* idepth_min,
                               float* idepth_max, float* quality, int* status, float* lastTraceUV2, float* lastTracePixelInterval) {
    if (!t || n_hosts < 1 || n_hosts > kTraceMaxHosts || !KRKi9 || !Kt3 || !aff2) return SDVGN_E_ARG;
    if (!t->haveNew) return SDVGN_E_STATE;
    const int n = t->tp_n;
    if (n == 0) return SDVGN_OK;
    if (!idepth_min || !idepth_max || !quality || !status || !lastTraceUV2 || !lastTracePixelInterval) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    TraceConst C;
    C.n = n; C.w = t->w[0]; C.h = t->h[0]; C.dI = t->pyr_dev[0];
    std::memset(C.KRKi, 0, sizeof(C.KRKi)); std::memset(C.Kt, 0, sizeof(C.Kt)); std::memset(C.aff, 0, sizeof(C.aff));
    std::memcpy(C.KRKi, KRKi9, sizeof(float) * 9 * n_hosts);
    std::memcpy(C.Kt, Kt3, sizeof(float) * 3 * n_hosts);
    std::memcpy(C.aff, aff2, sizeof(float) * 2 * n_hosts);
    const size_t cap = t->tp_cap;
    float* hs = (float*)t->tp_state_host;
    float* hmin = hs, *hmax = hs + cap, *hq = hs + 2 * cap;
    int* hst = (int*)(hs + 3 * cap);
    float* huv = hs + 4 * cap, *hiv = hs + 6 * cap;
    std::memcpy(hmin, idepth_min, 4 * (size_t)n); std::memcpy(hmax, idepth_max, 4 * (size_t)n); std::memcpy(hq, quality, 4 * (size_t)n);
    std::memcpy(hst, status, 4 * (size_t)n); std::memcpy(huv, lastTraceUV2, 8 * (size_t)n); std::memcpy(hiv, lastTracePixelInterval, 4 * (size_t)n);
    const float* sd = t->tp_static_dev;
    // the dynamic state (28 B per point) is read and written in place in pinned host memory by the kernel: no copy engine
    k_trace_points<<<(n + 63) / 64, 64, 0, t->stream>>>(C, sd, sd + cap, sd + 2 * cap, (const float4*)(sd + 3 * cap), (const float4*)(sd + 7 * cap),
                                                      (const float4*)(sd + 15 * cap), (const int*)(sd + 23 * cap), hmin, hmax, hq, hst, (float2*)huv, hiv);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->stream));
    std::memcpy(idepth_min, hmin, 4 * (size_t)n); std::memcpy(idepth_max, hmax, 4 * (size_t)n); std::memcpy(quality, hq, 4 * (size_t)n);
    std::memcpy(status, hst, 4 * (size_t)n); std::memcpy(lastTraceUV2, huv, 8 * (size_t)n); std::memcpy(lastTracePixelInterval, hiv, 4 * (size_t)n);
    return SDVGN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// makeCoarseDepthL0 / makeCoarseDepthForFirstFrame: reference template on the device
// ---------------------------------------------------------------------------------------------------------------
int sdvgn_tracker_make_coarse_depth(sdvgn_tracker* t, int n, const int* u, const int* v, const float* new_idepth, const float* weight,
                                    const float* const* ref_pyr_dev) {
    if (!t || n < 0 || (n > 0 && (!u || !v || !new_idepth || !weight))) return SDVGN_E_ARG;
    if (!ref_pyr_dev && !t->haveNew) return SDVGN_E_STATE;
    const int L = t->levels, w0 = t->w[0], h0 = t->h[0];
    for (int i = 0; i < n; ++i) if (u[i] < 0 || u[i] >= w0 || v[i] < 0 || v[i] >= h0) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    size_t npix = 0, nrows = 0;
    for (int l = 0; l < L; ++l) { npix += (size_t)t->w[l] * t->h[l]; nrows += t->h[l]; }
    if (!t->cd_maps) {
        HIPCHK(SDVGN_DMALLOC((void**)&t->cd_maps, sizeof(float) * 3 * npix));
        HIPCHK(SDVGN_DMALLOC((void**)&t->cd_rows, sizeof(int) * 2 * nrows));
        HIPCHK(SDVGN_HMALLOC((void**)&t->cd_n_host, sizeof(int) * SDVGN_MAX_LEVELS));
        t->cd_slot.assign((size_t)w0 * h0, 0);
        t->cd_stamp.assign((size_t)w0 * h0, 0);
    }
    if (n > t->cd_stage_cap) {
        HIPCHK(hipStreamSynchronize(t->stream));
        SDVGN_HFREE(t->cd_stage);
        t->cd_stage = nullptr; t->cd_stage_cap = 0;
        const int cap = n + n / 2 + 1024;
        HIPCHK(SDVGN_HMALLOC(&t->cd_stage, (size_t)cap * 12));
        t->cd_stage_cap = cap;
    }
    // splat in tuple order (:266-293): tuples that hit one pixel are summed here, sequentially, exactly like `idepth[0][k] += ...`
    int* s_pix = (int*)t->cd_stage;
    float* s_val = (float*)(s_pix + t->cd_stage_cap);
    float* s_wgt = s_val + t->cd_stage_cap;
    int nu = 0;
    const int gen = ++t->cd_gen;
    for (int i = 0; i < n; ++i) {
        const int k = u[i] + w0 * v[i];
        if (t->cd_stamp[k] != gen) { t->cd_stamp[k] = gen; t->cd_slot[k] = nu; s_pix[nu] = k; s_val[nu] = 0.0f; s_wgt[nu] = 0.0f; ++nu; }
        const int sl = t->cd_slot[k];
        s_val[sl] += new_idepth[i] * weight[i];
        s_wgt[sl] += weight[i];
    }
    CdLevels C;
    C.levels = L;
    {
        float* p = t->cd_maps;
        int r = 0;
        for (int l = 0; l < L; ++l) {
            const size_t np = (size_t)t->w[l] * t->h[l];
            C.w[l] = t->w[l]; C.h[l] = t->h[l];
            C.idepth[l] = p; C.wsum[l] = p + np; C.wdil[l] = p + 2 * np;
            p += 3 * np;
            C.ref[l] = ref_pyr_dev ? ref_pyr_dev[l] : t->pyr_dev[l];
            if (!C.ref[l]) return SDVGN_E_ARG;
            C.pc[l] = t->pc_dev[l];
            C.row0[l] = r; r += t->h[l];
        }
    }
    int* rowcount = t->cd_rows;
    int* rowoff = t->cd_rows + nrows;
    HIPCHK(hipMemsetAsync(C.idepth[0], 0, sizeof(float) * 2 * (size_t)w0 * h0, t->stream));   // idepth[0] and wsum[0] are adjacent
    if (nu) k_cd_scatter<<<(nu + 255) / 256, 256, 0, t->stream>>>(nu, s_pix, s_val, s_wgt, C.idepth[0], C.wsum[0]);
    for (int l = 1; l < L; ++l)
        k_cd_pyr<<<(t->w[l] * t->h[l] + 255) / 256, 256, 0, t->stream>>>(t->w[l], t->h[l], t->w[l - 1], C.idepth[l - 1], C.wsum[l - 1], C.idepth[l], C.wsum[l]);
    k_cd_dilate<<<dim3((w0 * h0 + 255) / 256, L), 256, 0, t->stream>>>(C);
    k_cd_count<<<dim3(h0, L), 256, 0, t->stream>>>(C, rowcount);
    k_cd_scan<<<L, 256, 0, t->stream>>>(C, rowcount, rowoff, t->cd_n_host);
    k_cd_emit<<<dim3(h0, L), 256, 0, t->stream>>>(C, rowoff, t->max_points);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->stream));
    for (int l = 0; l < L; ++l) {
        if (t->cd_n_host[l] > t->max_points) { t->pc_n[l] = 0; return SDVGN_E_ARG; }   // create the tracker with max_points = w*h like the reference (:49-55)
        t->pc_n[l] = t->cd_n_host[l];
    }
    return SDVGN_OK;
}

int sdvgn_tracker_get_ref(sdvgn_tracker* t, int lvl, float* u, float* v, float* idepth, float* color) {
    if (!t || lvl < 0 || lvl >= t->levels) return SDVGN_E_ARG;
    const int n = t->pc_n[lvl];
    if (!u) return n;
    if (!v || !idepth || !color) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(t->device));
    std::vector<float4> p(n);
    HIPCHK(hipStreamSynchronize(t->stream));
    if (n) HIPCHK(hipMemcpy(p.data(), t->pc_dev[lvl], sizeof(float4) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) { u[i] = p[i].x; v[i] = p[i].y; idepth[i] = p[i].z; color[i] = p[i].w; }
    return n;
}
