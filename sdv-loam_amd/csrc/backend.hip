// backend.hip -- host side of the GPU sliding-window back end + its C ABI (include/sdvgn.h, sdvgn_ef_*).
//
// Mirrors the data-parallel part of class EnergyFunctional (src/OptimizationBackend/EnergyFunctional.{h,cpp}) and of
// FullSystem::linearizeAll / applyRes_Reductor (src/FullSystem/FullSystemOptimize.cpp:23-159) on a flattened window:
//   device: linearize, applyRes, per-point sums, the (host,target) 13x13 accumulators and the Schur-complement
//           accumulators (as MFMA Gram matrices), resubstitution of the point steps;
//   host:   FrameFramePrecalc / adjoints / deltas (nF^2 tiny 3x3 / 6x6 products), the double-precision stitch of the
//           (4+6nF)^2 system, the Jacobi-preconditioned LDLT solve and the null-space projection -- all O(nF^3) and
//           independent of the number of points ("tiny; keep on host").
#include "../../include/sdvgn.h"
#include "sdvgn_debug.h"
#include "devmem.hpp"
#include "backend_kernels.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library is resolved at run time (dlopen), so libsdvgn has no link-time RCCL dependency
#include "gnmath.hpp"
#include "backend_solve.inc"
#include "tracker_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <unordered_map>
#include <vector>

#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return -(int)_e;        \
    } while (0)

#define EF_DEVICE(e)                                         \
    do {                                                     \
        if ((e)->host_only) return SDVGN_E_STATE;            \
        HIPCHK(hipSetDevice((e)->device));                   \
    } while (0)

using namespace sdvgn;

namespace {

constexpr int CPARS = 4;
constexpr int kTopP = 256;        // partial buffers: one 16x16 tile per (host,target) and workgroup
constexpr int kScP = 10 * 256;    // partial buffers: 10 upper 16x16 tiles of the 64x64 Gram per host and workgroup
constexpr int kTopE = 121;        // packed accumulators: the live 11x11 of the top tile, row-major
constexpr int kScN = 53;          // live features of the Schur Gram
constexpr int kScE = kScN * (kScN + 1) / 2;   // packed accumulators: upper triangle of 53x53, row-major (1431)
constexpr int kMaxChunks = 64;
const float kScaleXiRot = 1.0f, kScaleXiTrans = 0.5f, kScaleA = 10.0f, kScaleB = 1000.0f;
const float kInitialRotPrior = 1e11f, kInitialTransPrior = 1e10f, kInitialCalibHessian = 5e9f, kIdepthFixPrior = 50 * 50;

struct FrameH {
    gn::Pose evalPT, PRE_worldToCam, PRE_camToWorld;
    double state[10], state_zero[10], state_scaled[10];
    double prior[6], delta[6], delta_prior[6], state_backup[10];
    int frameID;
    float ab_exposure, frameEnergyTH;
};

struct WinEdit;   // backend_window.inc
WinEdit* win_new();
void win_delete(WinEdit*);
void win_forget(WinEdit*);
bool ef_win_is_open(const WinEdit*);

template <typename T>
int dev_alloc_tagged(T** p, size_t n, const char* tag) { return gmem::dmalloc_impl((void**)p, sizeof(T) * (n ? n : 1), alignof(T), tag) == hipSuccess ? 0 : -1; }
#define dev_alloc(p, n) dev_alloc_tagged((p), (n), #p)

}  // namespace

// (arguments of the statistics workgroup -- sum_stats_body below -- and of the accept test it may carry)
struct DecideArgs { double En, EM, rhs; int* accept_dev; int on; unsigned* verdict; unsigned seq; const double* en_em; };
struct StatsLaunch { const double* pe; int nE; const double* pl; int nL; const double* ps; int nS; double* out; volatile int* done_flag; int done_seq; DecideArgs dec; };
struct sdvgn_ef {
    int device = 0, w = 0, h = 0, max_points = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int nF = 0, nP = 0, nR = 0;
    int h0 = 0, h1 = SDVGN_MAX_FRAMES;   // host-frame shard of this rank (only meaningful when shard_set)
    bool shard_set = false;              // sdvgn_ef_set_host_range was called; otherwise the shard is always [0, nF)
    // calib
    double value_scaled[4] = {0, 0, 0, 0}, value_minus_value_zero[4] = {0, 0, 0, 0};
    double value[4] = {0, 0, 0, 0}, value_zero[4] = {0, 0, 0, 0}, value_backup[4] = {0, 0, 0, 0};
    EFConst C{};
    std::vector<FrameH> frames;
    std::vector<int> phost, hostP0, r_slot;
    std::vector<double> adHost, adTarget;      // [h + t*nF][36]
    std::vector<float> adHostF, adTargetF;
    double cPrior[4];
    std::vector<double> HM, bM;
    std::vector<std::vector<double>> nullspaces;
    std::vector<double> ns_N, ns_Npi;   // normalised null-space basis and its pseudo-inverse (orthogonalize_x), cached
    bool ns_dirty = true;
    std::vector<double> HA, bA, Hsc, bsc, HFinal, bFinal, lastX;
    int resInA = 0;
    // device
    EFArrays A{};
    float *pu = nullptr, *pv = nullptr, *pidz = nullptr, *pid = nullptr, *pidepth_backup = nullptr, *ppriorF = nullptr, *pdeltaF = nullptr;
    float4 *pcolor = nullptr, *pweights = nullptr;
    uint8_t* psensor = nullptr;
    uint8_t* rflags = nullptr;
    int8_t *rstate = nullptr, *rstate_new = nullptr;
    // second set of the three state_New* planes: the optimize loop linearises a trial step into the other set and swaps on accept,
    // so that a rejected step needs no re-linearisation (the reference's is a bit-exact recomputation of what the set still holds)
    int8_t* rstate_new2 = nullptr;
    float *renergy_new2 = nullptr, *renergy_wo2 = nullptr;
    int new_cur = 0;   // which set holds the current state_New* values
    // second copies of the four planes applyRes writes (flags, state, energy, JpJd): in the optimize loop the linearise leaves the result of
    // applyRes in them (EFArrays::rflags_w) and an accepted step swaps the pointers -- no applyRes pass, nothing to undo after a rejected step.
    // applied_synced: every slot a linearise does NOT process (no residual / fixed linearisation) holds the same values in both copies; any entry
    // point that writes the first copies outside the loop clears it, the next optimize call copies once (ef_sync_applied)
    uint8_t* rflags_alt = nullptr; int8_t* rstate_alt = nullptr; float *renergy_alt = nullptr, *JpJd_alt = nullptr;
    bool applied_synced = false;
    // device-resident small solve (backend_solve.inc)
    SolveWindow* win_dev = nullptr;        // per-window constants of the solve (adjoints, priors, HM/bM, null-space basis, evalPT ...)
    SolveWindow* win_host = nullptr;       // pinned staging, TWO copies used in turn (ef_sync_window: the host never waits for the stream to refill one)
    int win_flip = 0;
    hipEvent_t up_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [window | state][half]: recorded behind the copy kernel that read that half
    bool up_ev_used[2][2] = {{false, false}, {false, false}};
    bool win_dirty = true;
    SolveState* sstate_dev = nullptr;      // [2]: calib value + frame states, the current set and the trial set of the optimize loop
    SolveState* sstate_host = nullptr;     // pinned staging, two copies used in turn
    int st_flip = 0;
    CalibDev* calib_dev = nullptr;         // [2]: CalibHessian float views of the two state sets
    CalibDev* calib_host = nullptr;        // pinned staging, two copies used in turn
    int st_cur = 0;                        // which of the two sets holds the current state
    double* en_em_dev = nullptr;           // [2 sets][2]: prior energy and M energy of the state in that set (step_energy_body, backend_solve.inc)
    bool state_dirty = true;               // the host mirror changed outside the loop: upload before the next device solve
    ResubX* rx_dev = nullptr;              // xc, xAd of the last solve (the resubstitute workgroups of k_ef_tail_resub read them)
    unsigned long long* xw_dev = nullptr;  // the same + x as tagged words (SolveIO::xw): the in-launch hand-off of k_ef_tail_resub; [kXwTh, +8): thresholds of a select in that launch
    SolveSys* sys_dev = nullptr;           // HA, bA, Hsc, bsc, HFinal, bFinal of the last solve
    SolveOut* sol_host = nullptr;          // pinned: x, step statistics, resInA, status
    // the REJECTED case solved ahead (ef_launch_spec_solve): while a trial step is linearised, one workgroup on the library's side stream factors
    // the system this body solved once more with the damping a rejection would bring (lambda * 100, iteration + 1).  If the step is rejected the
    // next body finds its solution ready and skips accumulate, reduce, stitch and the factorisation; if it is accepted the result is dropped.
    // No events: the side-stream launch waits for the main solve's x words (tagged, SolveIO::wait_xw) and publishes its own results as tagged
    // words in xw_spec[buffer], which the next body's resubstitute / step workgroups poll exactly like they poll a solve of their own launch.
    ResubX* rx_spec = nullptr;             // device: plain copies of xc, xAd of the speculative solve (nobody reads them; the kernel writes both forms)
    unsigned long long* xw_spec = nullptr; // device: [2][512] tagged words, alternating per speculative solve
    SolveOut* sol_spec = nullptr;          // pinned [2]: x, step statistics, status of the speculative solves (flags: flags_host[5 + buffer])
    int seq_spec = 0;                      // tags of the speculative solves are kSpecTag | seq_spec: disjoint from the main solves' (seq_solve)
    hipStream_t side = nullptr;            // the device's shared side stream (not owned)
    int spec_last_buf = -1, spec_last_seq = 0;   // the speculative solve launched last (may still be running)
    SolvePieces* pieces_dev = nullptr;     // [SDVGN_MAX_FRAMES]: per-host shares of HA / bA / Hsc / bsc
    unsigned long long* solve_stamps = nullptr;   // pinned, 16 words: SDVGN_DEBUG_FLAGS bit6 only (phase stamps of the solve workgroup)
    int solve_status = 0;                  // status of the last device solve: 1 = a pivot of the LDL^T was not positive / finite (x = 0)
    int seq_solve = 0;                     // flags_host[3]
    bool sys_on_device = false, sys_fetched = false, sys_valid = false;
    void* fin_dev = nullptr;       // outputs of sdvgn_ef_optimize_finish (relbs_max, ngood_inc, removed), grown on demand
    void* fin_host = nullptr;      // pinned mirror of fin_dev
    void* jstage_dev = nullptr;    // staging of sdvgn_ef_set_residual_jacobians (slots | rows | res_toZero), grown on demand
    size_t jstage_bytes = 0;
    size_t fin_bytes = 0;
    float* th_dev = nullptr;       // frameEnergyTH [2 sets][SDVGN_MAX_FRAMES]: one per state_New* set (setNewFrameEnergyTH after every linearizeAll)
    float* th_log = nullptr;       // pinned ring: threshold of the newest frame after each linearizeAll of the last optimize call (trace, tests)
    int th_log_n = 0;
    size_t stats_cap = 0;          // doubles behind stats_dev: 4 statistics + max_points quantile candidates (sharded path)
    std::vector<double> iter_us;   // wall time of every loop body of the last sdvgn_ef_optimize call (microseconds)
    int n_accepted = 0;            // accepted steps of the last sdvgn_ef_optimize call
    int n_merged_tests = 0, n_pre_acc = 0;      // of the last call: accept tests taken as a workgroup of the next body's accumulate (k_ef_acc_stats) / accumulates queued ahead of the verdict
    int n_spec_launched = 0, n_spec_used = 0;   // of the last call: rejected cases solved ahead on the side stream / bodies that started from such a solution
    // deferred work of the optimize loop (see linearize_launch): the threshold select of the last linearisation and the re-classification
    // after a rejected step ride in later launches as extra workgroups; whatever is still pending is launched on its own by ef_flush_pending
    SelArgs pend_sel{}; bool pend_sel_valid = false;
    int lin_partials = 0, lin_nL = 0;   // of the linearise launched last (linearize_launch_kernels -> linearize_launch_stats)
    int* accept_dev = nullptr;          // verdict of the device-side accept test ([0]; [4]: the tagged verdict word of k_ef_stats_apply)
    unsigned seq_verdict = 0;
    ReclArgs pend_rc{}; bool pend_rc_valid = false;
    bool time_lin = false;         // optimize flags bit3: HIP event pair around every k_ef_linearize launch of the call
    std::vector<hipEvent_t> lin_events;
    size_t lin_ev_used = 0;
    std::vector<float> lin_ms;     // their durations (milliseconds), in launch order
    float2* rmatcher = nullptr;
    float *renergy = nullptr, *renergy_new = nullptr, *renergy_wo = nullptr, *rres_toZero = nullptr, *J = nullptr, *JpJd = nullptr;
    float *pHddA = nullptr, *pbdA = nullptr, *pHcdA = nullptr, *pHddL = nullptr, *pbdL = nullptr, *pHcdL = nullptr, *pHdi = nullptr,
          *pbdSum = nullptr, *pHcd = nullptr, *pstep = nullptr;
    uint8_t* pnogood = nullptr;    // EFArrays::pnogood
    uint8_t nogood_epoch = 0;      // epoch of the last optimize call (0: none yet)
    // the statistics of the call's INITIAL linearizeAll, not launched yet: nothing needs them before the first accept test, so they ride as a workgroup of the first
    // body's accumulate (k_ef_acc_stats) instead of a 5 us launch between that linearise and that accumulate; ef_flush_stats launches them alone if no accumulate comes
    StatsLaunch pend_stats{};
    bool pend_stats_valid = false;
    float* images = nullptr;
    float* img_stage = nullptr;
    int *phost_dev = nullptr, *hostP0_dev = nullptr;
    PrecalcDev* precalc_dev = nullptr;
    PrecalcDev* precalc_host = nullptr;  // pinned
    double* energy_partial = nullptr;
    float *top_partial = nullptr, *sc_partial = nullptr;
    int* nres_partial = nullptr;
    ImmPrecalc* imm_pc_dev = nullptr;       // optimizeImmaturePoint: PRE_RTll / PRE_tTll / PRE_aff_mode per (host,target)
    ImmPrecalc* imm_pc_host = nullptr;      // pinned
    void* imm_stage = nullptr;              // pinned candidate / result staging, grown on demand
    size_t imm_stage_bytes = 0;
    double* acc_dev = nullptr;    // packed: top [nF*nF][256] | sc [nF][2560] | resInA
    double* acc_host = nullptr;   // pinned; the reduce kernels write it DIRECTLY (zero-copy) when no all-reduce is installed
    double* stats_host = nullptr; // pinned, 4 doubles: same for k_ef_sum_stats
    int arith = 0;                 // sdvgn_ef_set_arith: 0 = the reference's arithmetic in k_ef_linearize (default), 1 = tolerance mode (lin_fast)
    bool reuse_system = false;     // set by sdvgn_ef_optimize (flags bit2) for the solve that follows a rejected step: HA/bA/Hsc/bsc and
                                   // the per-point Schur terms on the device are those of the identical state one body earlier
    // second copies of the planes a trial step overwrites (point idepths, precalc table): the optimize loop writes the trial values
    // into them and swaps the pointers, so that loadSateBackup after a rejected step is a pointer swap back, not two launches
    float *pid_alt = nullptr, *pidz_alt = nullptr, *pdeltaF_alt = nullptr;
    PrecalcDev* precalc_alt = nullptr;
    unsigned long long* dbg_stamps = nullptr;     // SDVGN_DEBUG_FLAGS bit5 only (kDbgStampWords words)
    uint8_t *marg_mask_dev = nullptr, *drop_mask_dev = nullptr;   // point masks of fixLinearization / marginalizePointsF
    int resInM = 0;
    const PrecalcDev* precalc_staged = nullptr;   // pinned half the last ef_upload_precalc filled
    bool in_optimize_loop = false;                // finish_solve then also does doStepFromBackup's host part before its launch
    float step_sumT = 0, step_sumR = 0;
    ncclComm_t rccl_comm = nullptr;   // cfg4 with the collectives issued by the library itself (sdvgn_ef_init_rccl)
    int* flags_host = nullptr;    // pinned: [0] top accumulators done, [1] all accumulators done, [2] linearize statistics done
    unsigned* done_ctr = nullptr; // device: workgroup counters for the multi-workgroup publishers (2)
    int seq_top = 0, seq_acc = 0, seq_stats = 0;
    bool acc_in_host = false;     // the last accumulate wrote acc_host directly (acc_dev not updated)
    double* stats_dev = nullptr;   // {linearize energy, L-energy point part, sum step^2, sum |idepth_backup|}
    bool own_acc = true, own_stats = true;
    // sharded window, ONE collective per loop body (sdvgn_ef_set_collective_buffer): two message buffers [acc | 4 statistics | nP quantile
    // candidates]; coll_cur holds the reduced accumulators of the CURRENT state, the other one receives the speculative message of a trial
    double* coll[2] = {nullptr, nullptr};
    size_t coll_stride = 0;        // doubles per message buffer
    int coll_cur = 0;
    bool own_coll = false;
    ApplyBackup apply_bak{nullptr, nullptr, nullptr, nullptr};
    unsigned long long n_collectives = 0;   // all-reduces issued through ef_allreduce (tests)
    void (*allreduce)(void*, double*, int) = nullptr;   // cfg4: sum a device buffer over the ranks (RCCL), in stream order
    void* allreduce_user = nullptr;
    bool host_only = false;
    double* stats_partial = nullptr;
    size_t slots_cap = 0;
    bool havePrecalc = false, haveAdjoints = false;
    bool deltaF_nonzero = false, has_linearized = false;   // when both are false the point part of calcLEnergy is exactly 0
    int precalc_flip = 0;                                  // two pinned staging halves -> no host sync per upload
    // the window edited in place (backend_window.inc): image slot of every frame (identity unless frames were inserted / removed), the edit
    // state, and table_mode: the residual tables are addressed by slot, there is no caller-side residual list (r_slot empty, nR = 0)
    uint8_t img_slot[SDVGN_MAX_FRAMES] = {0, 1, 2, 3, 4, 5, 6, 7};
    WinEdit* win = nullptr;
    bool table_mode = false;
};

struct PhaseTimer {   // SDVGN_PROFILE=1: host wall time per phase of the optimize loop, printed by sdvgn_ef_optimize
    bool on = getenv("SDVGN_PROFILE") != nullptr;
    double acc[16] = {0};
    long cnt[16] = {0};
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void stop(int k) { if (on) { auto t1 = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::micro>(t1 - t0).count(); cnt[k]++; t0 = t1; } }
};
static thread_local PhaseTimer g_pt;   // per thread: sdvgn_ef_optimize_batch runs the loop on several host threads
enum { PT_ACCUM = 0, PT_D2H, PT_STITCH_TOP, PT_STITCH_SC, PT_SOLVE, PT_RESUB, PT_STEP, PT_PRECALC, PT_LIN, PT_APPLY, PT_PREP, PT_N };
static const char* kPtNames[PT_N] = {"launches (acc..solve..resub..linearize..stats)", "wait for x (solve flag)", "-", "-", "-", "-", "host mirror of the step", "-", "wait for the statistics", "decision + apply/restore", "-"};

constexpr size_t kDbgStampWords = (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * 2 * kMaxChunks * 4 * 8;
static size_t acc_count(const sdvgn_ef* e) { return (size_t)e->nF * e->nF * kTopE + (size_t)e->nF * kScE + 1; }

static void frame_set_state(FrameH& f, const double* state) {  // FrameHessian::setState, HessianBlocks.h:131-143
    for (int i = 0; i < 10; ++i) f.state[i] = state[i];
    for (int i = 0; i < 3; ++i) f.state_scaled[i] = kScaleXiTrans * state[i];
    for (int i = 3; i < 6; ++i) f.state_scaled[i] = kScaleXiRot * state[i];
    f.state_scaled[6] = kScaleA * state[6]; f.state_scaled[7] = kScaleB * state[7];
    f.state_scaled[8] = kScaleA * state[8]; f.state_scaled[9] = kScaleB * state[9];
    f.PRE_worldToCam = gn::compose(gn::exp_se3(f.state_scaled), f.evalPT);
    f.PRE_camToWorld = gn::inverse(f.PRE_worldToCam);
}

static void m3f_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}

// point the kernels' state_New* planes at set `write` (0/1) and the read-only "previous state_NewEnergy" at set `prev`;
// the thresholds the next linearise classifies with come from set `th_read` (default: prev), the ones its setNewFrameEnergyTH
// produces go to set `write`
static void ef_select_new_set(sdvgn_ef* e, int write, int prev, int th_read = -1) {
    EFArrays& A = e->A;
    A.rstate_new = write ? e->rstate_new2 : e->rstate_new;
    A.renergy_new = write ? e->renergy_new2 : e->renergy_new;
    A.renergy_wo = write ? e->renergy_wo2 : e->renergy_wo;
    A.renergy_new_prev = prev ? e->renergy_new2 : e->renergy_new;
    if (th_read < 0) th_read = prev;
    A.frameTH_r = e->th_dev + (size_t)th_read * SDVGN_MAX_FRAMES;
    A.frameTH_w = e->th_dev + (size_t)write * SDVGN_MAX_FRAMES;
}
constexpr int kThLog = 1024;
// a new epoch for EFArrays::pnogood ("during THIS optimize call"): the plane is cleared once per 255 calls instead of by a memset launch at the head of every call
static int ef_next_nogood_epoch(sdvgn_ef* e, hipStream_t s) {
    if (e->nogood_epoch == 255) { HIPCHK(hipMemsetAsync(e->pnogood, 0, (size_t)e->max_points, s)); e->nogood_epoch = 0; }
    e->A.nogood_epoch = ++e->nogood_epoch;
    return SDVGN_OK;
}

static void ef_fill_arrays(sdvgn_ef* e) {
    EFArrays& A = e->A;
    A.pu = e->pu; A.pv = e->pv; A.pidz = e->pidz; A.pid = e->pid; A.pcolor = e->pcolor; A.pweights = e->pweights;
    A.ppriorF = e->ppriorF; A.pdeltaF = e->pdeltaF; A.psensor = e->psensor;
    A.rflags = e->rflags; A.rstate = e->rstate; A.rstate_new = e->rstate_new; A.rmatcher = e->rmatcher;
    A.renergy = e->renergy; A.renergy_new = e->renergy_new; A.renergy_wo = e->renergy_wo; A.rres_toZero = e->rres_toZero;
    A.renergy_new_prev = e->renergy_new; e->new_cur = 0;
    A.frameTH_r = e->th_dev; A.frameTH_w = e->th_dev;
    A.J = e->J; A.JpJd = e->JpJd;
    A.pHddA = e->pHddA; A.pbdA = e->pbdA; A.pHcdA = e->pHcdA; A.pHddL = e->pHddL; A.pbdL = e->pbdL; A.pHcdL = e->pHcdL;
    A.pHdi = e->pHdi; A.pbdSum = e->pbdSum; A.pHcd = e->pHcd; A.pstep = e->pstep; A.pnogood = e->pnogood; A.nogood_epoch = e->nogood_epoch;
    A.images = e->images;
    A.img_slots = 0;
    for (int t = 0; t < SDVGN_MAX_FRAMES; ++t) A.img_slots |= (unsigned)(e->img_slot[t] & 7) << (4 * t);
    A.dbg_stamps = e->dbg_stamps;
    A.reset_oob = 0;
    A.rflags_w = nullptr; A.rstate_w = nullptr; A.renergy_w = nullptr; A.JpJd_w = nullptr;
    A.err = e->stats_host ? (unsigned*)(e->stats_host + 6) : nullptr;
}
// ---- applyRes fused into the linearise (EFArrays::rflags_w) ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ef_sync_applied(size_t slots, const uint8_t* __restrict__ fl, const int8_t* __restrict__ st, const float* __restrict__ en,
                                                         const float* __restrict__ jd, uint8_t* __restrict__ fl2, int8_t* __restrict__ st2, float* __restrict__ en2,
                                                         float* __restrict__ jd2) {
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= slots) return;
    const uint8_t f = fl[s];
    fl2[s] = f;
    if (!(f & RF_EXISTS)) return;
    st2[s] = st[s]; en2[s] = en[s];
#pragma unroll
    for (int i = 0; i < 6; ++i) jd2[(size_t)i * slots + s] = jd[(size_t)i * slots + s];
}
static int ef_sync_applied(sdvgn_ef* e) {
    if (e->applied_synced) return SDVGN_OK;
    const size_t slots = (size_t)e->nF * e->nP;
    k_ef_sync_applied<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(slots, e->rflags, e->rstate, e->renergy, e->JpJd, e->rflags_alt, e->rstate_alt,
                                                                              e->renergy_alt, e->JpJd_alt);
    HIPCHK(hipGetLastError());
    e->applied_synced = true;
    return SDVGN_OK;
}
static void ef_flip_applied(sdvgn_ef* e) {      // the second copies become the current ones (an applyRes that has already been computed)
    std::swap(e->rflags, e->rflags_alt); std::swap(e->rstate, e->rstate_alt); std::swap(e->renergy, e->renergy_alt); std::swap(e->JpJd, e->JpJd_alt);
    e->A.rflags = e->rflags; e->A.rstate = e->rstate; e->A.renergy = e->renergy; e->A.JpJd = e->JpJd;
}
static void ef_set_apply_target(sdvgn_ef* e, bool on) {   // the next linearise launch also performs applyRes, into the second copies
    e->A.rflags_w = on ? e->rflags_alt : nullptr; e->A.rstate_w = on ? e->rstate_alt : nullptr;
    e->A.renergy_w = on ? e->renergy_alt : nullptr; e->A.JpJd_w = on ? e->JpJd_alt : nullptr;
}

static void ef_update_const(sdvgn_ef* e) {  // CalibHessian float views, HessianBlocks.h:302-330
    EFConst& C = e->C;
    C.nF = e->nF; C.nP = e->nP; C.w = e->w; C.h = e->h;
    C.fxl = (float)e->value_scaled[0]; C.fyl = (float)e->value_scaled[1];
    C.cxl = (float)e->value_scaled[2]; C.cyl = (float)e->value_scaled[3];
    C.fxli = 1.0f / C.fxl; C.fyli = 1.0f / C.fyl;
    C.wM3G = e->w - 3; C.hM3G = e->h - 3;
    for (int i = 0; i < 4; ++i) C.cDeltaF[i] = (float)e->value_minus_value_zero[i];
    C.huberTH = 6.0f; C.outlierTHSumComponent = 50 * 50;
    C.debug_flags = getenv("SDVGN_DEBUG_FLAGS") ? atoi(getenv("SDVGN_DEBUG_FLAGS")) : 0;
}

// ---------------- host stitch: top (AccumulatedTopHessian.cpp:181-242 + .h:100-113) --------------------------
// Host fp64 algebra of the solve (stitches, LDLT): compiled for FMA and allowed to contract -- it is tolerance-checked double-precision
// glue, not part of the bit-exact float paths (those are built with -ffp-contract=off and stay that way).
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define SDVGN_HOST_FMA __attribute__((target("avx2,fma")))
#else
#define SDVGN_HOST_FMA
#endif

SDVGN_HOST_FMA static void stitch_top(sdvgn_ef* e, const double* G /*[nF*nF][256]*/, bool use_prior = true) {
#pragma clang fp contract(fast)
    const int nF = e->nF, n = CPARS + 6 * nF;
    std::vector<double>& H = e->HA;
    std::vector<double>& b = e->bA;
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    const double sT[6] = {kScaleXiTrans, kScaleXiTrans, kScaleXiTrans, kScaleXiRot, kScaleXiRot, kScaleXiRot};   // adTarget = diag
    for (int k = 0; k < nF * nF; ++k) {
        const int h = k % nF, t = k / nF;
        if (h == t) continue;
        const int hIdx = CPARS + h * 6, tIdx = CPARS + t * 6;
        const double* g11 = G + (size_t)(h * nF + t) * kTopE;   // device pair index = h*nF + t, packed 11x11
        double g[11 * 16];                                       // re-strided to 16 so that the code below indexes g[r*16+c]
        for (int r = 0; r < 11; ++r) for (int c = 0; c < 11; ++c) g[r * 16 + c] = g11[r * 11 + c];
        const double* AH = &e->adHost[(size_t)(h + t * nF) * 36];
        // adTarget = diag(sT): only AH*A66 and (AH*A66)*AH^T are real 6x6 products
        double T1[36];
        for (int i = 0; i < 6; ++i) {
            double row[6] = {0, 0, 0, 0, 0, 0};
            for (int q = 0; q < 6; ++q) {
                const double a = AH[i * 6 + q];
                const double* gq = g + (4 + q) * 16 + 4;
                for (int j = 0; j < 6; ++j) row[j] += a * gq[j];
            }
            for (int j = 0; j < 6; ++j) T1[i * 6 + j] = row[j];
        }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double hh = 0;
                for (int q = 0; q < 6; ++q) hh += T1[i * 6 + q] * AH[j * 6 + q];
                H[(size_t)(hIdx + i) * n + hIdx + j] += hh;
                H[(size_t)(tIdx + i) * n + tIdx + j] += sT[i] * g[(4 + i) * 16 + 4 + j] * sT[j];
                H[(size_t)(hIdx + i) * n + tIdx + j] += T1[i * 6 + j] * sT[j];
            }
        for (int i = 0; i < 6; ++i) {
            double sh[CPARS + 1] = {0, 0, 0, 0, 0};
            for (int q = 0; q < 6; ++q) {
                const double a = AH[i * 6 + q];
                const double* gq = g + (4 + q) * 16;
                for (int j = 0; j < CPARS; ++j) sh[j] += a * gq[j];
                sh[CPARS] += a * gq[10];
            }
            for (int j = 0; j < CPARS; ++j) {
                H[(size_t)(hIdx + i) * n + j] += sh[j];
                H[(size_t)(tIdx + i) * n + j] += sT[i] * g[(4 + i) * 16 + j];
            }
            b[hIdx + i] += sh[CPARS];
            b[tIdx + i] += sT[i] * g[(4 + i) * 16 + 10];
        }
        for (int i = 0; i < CPARS; ++i) {
            for (int j = 0; j < CPARS; ++j) H[(size_t)i * n + j] += g[i * 16 + j];
            b[i] += g[i * 16 + 10];
        }
    }
    // priors (usePrior; marginalizePointsF stitches without them, EnergyFunctional.cpp:545)
    if (use_prior) {
    for (int i = 0; i < CPARS; ++i) { H[(size_t)i * n + i] += e->cPrior[i]; b[i] += e->cPrior[i] * (double)e->C.cDeltaF[i]; }
    for (int h = 0; h < nF; ++h)
        for (int i = 0; i < 6; ++i) {
            H[(size_t)(CPARS + h * 6 + i) * n + CPARS + h * 6 + i] += e->frames[h].prior[i];
            b[CPARS + h * 6 + i] += e->frames[h].prior[i] * e->frames[h].delta_prior[i];
        }
    }
    for (int h = 0; h < nF; ++h) {
        const int hIdx = CPARS + h * 6;
        for (int i = 0; i < CPARS; ++i) for (int j = 0; j < 6; ++j) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i];
        for (int t = h + 1; t < nF; ++t) {
            const int tIdx = CPARS + t * 6;
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[(size_t)(hIdx + i) * n + tIdx + j] += H[(size_t)(tIdx + j) * n + hIdx + i];
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[(size_t)(tIdx + i) * n + hIdx + j] = H[(size_t)(hIdx + j) * n + tIdx + i];
        }
    }
}

// ---------------- host stitch: Schur complement (AccumulatedSCHessian.cpp:64-135 + .h:108-113) ----------------
// G_h is the 64x64 Gram of host h (upper 16x16 tiles); features 6t+i (JpJdF of target t), 48-51 Hcd, 52 bdSum.
// Uses A_h = [AH_h0 | ... | AH_h,nF-1] (6 x 6nF) so that all  AH D AH^T / AH D AT^T / AT D AH^T  terms of one host come
// from B = A_h * D_h (6 x 6nF) instead of nF^2 separate 6x6x6 products.
// Works on the packed upper triangles directly (no 64x64 unpack): D_h symmetric, so  B = A_h D_h  is assembled from each packed
// row q as an AXPY over the columns c >= q plus, by symmetry, a sum over c > q into column q (two accumulator sets).  The terms
// that do not involve AH -- diag(sT) D_h diag(sT) -- are summed over the hosts in packed form and applied once; the AH D AT^T row
// blocks go to R and enter as R + R^T at the end.
SDVGN_HOST_FMA static void stitch_sc(sdvgn_ef* e, const double* Gall /*[nF][1431]*/) {
#pragma clang fp contract(fast)
    const int nF = e->nF, n = CPARS + 6 * nF, nf6 = 6 * nF;
    std::vector<double>& H = e->Hsc;
    std::vector<double>& b = e->bsc;
    const double* adHost = e->adHost.data();
    H.assign((size_t)n * n, 0); b.assign(n, 0);
    const double sT[6] = {kScaleXiTrans, kScaleXiTrans, kScaleXiTrans, kScaleXiRot, kScaleXiRot, kScaleXiRot};  // adTarget = diag
    double Ah[6 * 6 * SDVGN_MAX_FRAMES], B[6 * 6 * SDVGN_MAX_FRAMES];
    double sT48[6 * SDVGN_MAX_FRAMES];
    double R[6 * SDVGN_MAX_FRAMES * 6 * SDVGN_MAX_FRAMES];      // row-block terms  R[iIdx-4 + r][q] = B[r][q] * sT48[q]; H += R + R^T at the end
    double Dsum[48 * 49 / 2 + 48];      // sum over hosts of the packed upper 48x48 (row-major upper, row q has 48-q entries)
    int off[kScN];                      // offset of packed row r: entries (r, r..52)
    { int k = 0; for (int r = 0; r < kScN; ++r) { off[r] = k; k += kScN - r; } }
    for (int q = 0; q < nf6; ++q) sT48[q] = sT[q % 6];
    std::memset(R, 0, sizeof(double) * (size_t)nf6 * nf6);
    std::memset(Dsum, 0, sizeof(Dsum));
    for (int h = 0; h < nF; ++h) {
        const double* gp = Gall + (size_t)h * kScE;
        const int iIdx = CPARS + h * 6;
        for (int j = 0; j < nF; ++j) { const double* AH = &adHost[(size_t)(h + nF * j) * 36]; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Ah[(size_t)r * nf6 + 6 * j + c] = AH[r * 6 + c]; }
        // B = A_h * D with D symmetric, from its packed upper rows: row q contributes  B[:, q..] += A[:, q] u[q..]  (upper incl. diagonal)
        // and  B[:, q] += sum_{c > q} A[:, c] u[c]  (strict lower, by symmetry)
        std::fill(B, B + (size_t)6 * nf6, 0.0);
        double AT8[6 * SDVGN_MAX_FRAMES][8];            // A_h transposed, rows padded to 8
        for (int c = 0; c < nf6; ++c) { for (int r = 0; r < 6; ++r) AT8[c][r] = Ah[(size_t)r * nf6 + c]; AT8[c][6] = AT8[c][7] = 0; }
        for (int q = 0; q < nf6; ++q) {
            const double* u = gp + off[q];          // u[c - q] = D[q][c], c >= q
            const int len = nf6 - q;
            // upper part incl. diagonal: B[:, q..] += A[:, q] u[q..]
            for (int r = 0; r < 6; ++r) {
                const double a = Ah[(size_t)r * nf6 + q];
                if (a == 0.0) continue;
                double* br = &B[(size_t)r * nf6 + q];
                for (int c = 0; c < len; ++c) br[c] += a * u[c];
            }
            // strict lower part by symmetry: B[:, q] += sum_{c > q} u[c] A[:, c]; two accumulator sets (even / odd c)
            double w0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int c = 1;
            for (; c + 1 < len; c += 2) {
                const double u0 = u[c], u1 = u[c + 1];
                const double* a0 = AT8[q + c]; const double* a1 = AT8[q + c + 1];
                for (int r = 0; r < 8; ++r) { w0[r] += u0 * a0[r]; w1[r] += u1 * a1[r]; }
            }
            if (c < len) { const double u0 = u[c]; const double* a0 = AT8[q + c]; for (int r = 0; r < 8; ++r) w0[r] += u0 * a0[r]; }
            for (int r = 0; r < 6; ++r) B[(size_t)r * nf6 + q] += w0[r] + w1[r];
        }
        {   // H[i,i] += B A_h^T, accumulated as 6 rows of 8 (A_h^T rows from AT8)
            double M[6][8];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 8; ++c) M[r][c] = 0;
            for (int q = 0; q < nf6; ++q) {
                const double* aq = AT8[q];
                for (int r = 0; r < 6; ++r) { const double bq = B[(size_t)r * nf6 + q]; for (int c = 0; c < 8; ++c) M[r][c] += bq * aq[c]; }
            }
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[(size_t)(iIdx + r) * n + iIdx + c] += M[r][c];
        }
        for (int r = 0; r < 6; ++r) { double* rrow = &R[(size_t)(6 * h + r) * nf6]; const double* brow = &B[(size_t)r * nf6]; for (int q = 0; q < nf6; ++q) rrow[q] += brow[q] * sT48[q]; }
        // sum of the packed 48x48 parts (AT D AT^T is applied once after the loop)
        { int k = 0; for (int q = 0; q < nf6; ++q) { const double* u = gp + off[q]; const int len = nf6 - q; for (int c = 0; c < len; ++c) Dsum[k + c] += u[c]; k += len; } }
        for (int j = 0; j < nF; ++j) {
            const int jIdx = CPARS + j * 6; const double* AH = &adHost[(size_t)(h + nF * j) * 36];
            for (int r = 0; r < 6; ++r) {
                const double* ur = gp + off[6 * j + r] - (6 * j + r);   // ur[c] = G[6j+r][c], c >= 6j+r
                for (int c = 0; c < CPARS; ++c) H[(size_t)(jIdx + r) * n + c] += sT[r] * ur[48 + c];
                b[jIdx + r] += sT[r] * ur[52];
            }
            double e5[6][5];
            for (int q = 0; q < 6; ++q) { const double* uq = gp + off[6 * j + q] - (6 * j + q); for (int c = 0; c < 5; ++c) e5[q][c] = uq[48 + c]; }
            for (int r = 0; r < 6; ++r) {
                double sh[5] = {0, 0, 0, 0, 0};
                for (int q = 0; q < 6; ++q) { const double a = AH[r * 6 + q]; for (int c = 0; c < 5; ++c) sh[c] += a * e5[q][c]; }
                for (int c = 0; c < CPARS; ++c) H[(size_t)(iIdx + r) * n + c] += sh[c];
                b[iIdx + r] += sh[4];
            }
        }
        for (int r = 0; r < CPARS; ++r) { const double* ur = gp + off[48 + r] - (48 + r); for (int c = r; c < CPARS; ++c) { H[(size_t)r * n + c] += ur[48 + c]; if (c != r) H[(size_t)c * n + r] += ur[48 + c]; } b[r] += ur[52]; }
    }
    // frame block: R + R^T + diag(sT) Dsum diag(sT)
    { int k = 0;
      for (int q = 0; q < nf6; ++q) { const int len = nf6 - q; for (int c = 0; c < len; ++c) { const double v = sT48[q] * Dsum[k + c] * sT48[q + c]; H[(size_t)(CPARS + q) * n + CPARS + q + c] += v; if (c) H[(size_t)(CPARS + q + c) * n + CPARS + q] += v; } k += len; } }
    for (int r = 0; r < nf6; ++r) for (int c = 0; c < nf6; ++c) H[(size_t)(CPARS + r) * n + CPARS + c] += R[(size_t)r * nf6 + c] + R[(size_t)c * nf6 + r];
    for (int h = 0; h < nF; ++h) { const int hIdx = CPARS + h * 6; for (int i = 0; i < CPARS; ++i) for (int j = 0; j < 6; ++j) H[(size_t)i * n + hIdx + j] = H[(size_t)(hIdx + j) * n + i]; }
}

// one-sided Jacobi SVD (m x k, k small) used by the null-space projection
static void svd_jacobi(int m, int k, std::vector<double>& A, std::vector<double>& s, std::vector<double>& V) {
    V.assign((size_t)k * k, 0);
    for (int i = 0; i < k; ++i) V[(size_t)i * k + i] = 1;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                double a = 0, bb = 0, c = 0;
                for (int i = 0; i < m; ++i) { const double x = A[(size_t)i * k + p], y = A[(size_t)i * k + q]; a += x * x; bb += y * y; c += x * y; }
                off = std::max(off, std::fabs(c) / std::sqrt(std::max(a * bb, 1e-300)));
                if (std::fabs(c) < 1e-300) continue;
                const double zeta = (bb - a) / (2 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) { const double x = A[(size_t)i * k + p], y = A[(size_t)i * k + q]; A[(size_t)i * k + p] = cs * x - sn * y; A[(size_t)i * k + q] = sn * x + cs * y; }
                for (int i = 0; i < k; ++i) { const double x = V[(size_t)i * k + p], y = V[(size_t)i * k + q]; V[(size_t)i * k + p] = cs * x - sn * y; V[(size_t)i * k + q] = sn * x + cs * y; }
            }
        if (off < 1e-15) break;
    }
    s.assign(k, 0);
    for (int j = 0; j < k; ++j) {
        double nn = 0;
        for (int i = 0; i < m; ++i) nn += A[(size_t)i * k + j] * A[(size_t)i * k + j];
        s[j] = std::sqrt(nn);
        for (int i = 0; i < m; ++i) A[(size_t)i * k + j] = s[j] > 0 ? A[(size_t)i * k + j] / s[j] : 0;  // A becomes U
    }
}

// EnergyFunctional::orthogonalize(&x, 0)  EnergyFunctional.cpp:615-648.  The normalised null-space basis N and its pseudo-inverse
// Npi (via SVD, singular values below setting_solverModeDelta * max cut) only change with sdvgn_ef_set_nullspaces: computed once.
static void ef_prepare_nullspace(sdvgn_ef* e, int n) {
    const int k = (int)e->nullspaces.size();
    if (!e->ns_dirty && e->ns_N.size() == (size_t)n * k) return;
    std::vector<double> N((size_t)n * k), U, sv, V;
    for (int j = 0; j < k; ++j) {
        double nn = 0;
        for (int i = 0; i < n; ++i) nn += e->nullspaces[j][i] * e->nullspaces[j][i];
        nn = std::sqrt(nn);
        for (int i = 0; i < n; ++i) N[(size_t)i * k + j] = e->nullspaces[j][i] / nn;
    }
    U = N;
    svd_jacobi(n, k, U, sv, V);
    double maxSv = 0;
    for (double v : sv) maxSv = std::max(maxSv, v);
    for (double& v : sv) v = (v > 1e-5 * maxSv) ? 1.0 / v : 0;   // setting_solverModeDelta
    std::vector<double> Npi((size_t)n * k, 0);                    // Npi = U S V^T
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j) { double a = 0; for (int q = 0; q < k; ++q) a += U[(size_t)i * k + q] * sv[q] * V[(size_t)j * k + q]; Npi[(size_t)i * k + j] = a; }
    e->ns_N = N; e->ns_Npi = Npi; e->ns_dirty = false;
}
static void orthogonalize_x(sdvgn_ef* e, std::vector<double>& x) {     // y = x - 0.5 (N Npi^T + Npi N^T) x
    const int n = (int)x.size(), k = (int)e->nullspaces.size();
    if (k == 0) return;
    ef_prepare_nullspace(e, n);
    const std::vector<double>&N = e->ns_N, &Npi = e->ns_Npi;
    std::vector<double> tN(k, 0), tP(k, 0);
    for (int j = 0; j < k; ++j) { double a = 0, c = 0; for (int i = 0; i < n; ++i) { a += N[(size_t)i * k + j] * x[i]; c += Npi[(size_t)i * k + j] * x[i]; } tN[j] = a; tP[j] = c; }
    for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < k; ++j) a += N[(size_t)i * k + j] * tP[j] + Npi[(size_t)i * k + j] * tN[j]; x[i] -= 0.5 * a; }
}

// pinned host -> device by loads of the kernel itself, up to two blocks per launch (8-byte words)
__global__ void __launch_bounds__(256) k_ef_copy_in(const unsigned long long* __restrict__ src0, unsigned long long* __restrict__ dst0, int n0,
                                                    const unsigned long long* __restrict__ src1, unsigned long long* __restrict__ dst1, int n1) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n0; i += gridDim.x * 256) dst0[i] = src0[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n1; i += gridDim.x * 256) dst1[i] = src1[i];
}

// ---- device-resident solve: the window constants and the state set the solve kernel reads (backend_solve.inc) -------------------
static int ef_sync_window(sdvgn_ef* e) {
    if (!e->win_dirty) return 0;
    const int nF = e->nF, n = CPARS + 6 * nF, k = (int)e->nullspaces.size();
    if (k > kMaxNs) return SDVGN_E_ARG;
    // two pinned copies in turn: the one filled now fed the upload before the previous one, which has long passed (checked: its event) -- the host
    // does not wait for the stream here (a key-frame's commit kernels may still be running on it: 50-60 us per key-frame when this was a stream sync)
    const int half = (e->win_flip ^= 1);
    if (e->up_ev_used[0][half]) HIPCHK(hipEventSynchronize(e->up_ev[0][half]));
    SolveWindow& W = e->win_host[half];
    std::memset(&W, 0, sizeof(W));
    W.nF = nF; W.n = n; W.ns_k = k;
    if (e->haveAdjoints) {
        std::memcpy(W.adHost, e->adHost.data(), sizeof(double) * (size_t)nF * nF * 36);
        std::memcpy(W.adHostF, e->adHostF.data(), sizeof(float) * (size_t)nF * nF * 36);
        std::memcpy(W.adTargetF, e->adTargetF.data(), sizeof(float) * (size_t)nF * nF * 36);
    }
    for (int i = 0; i < 4; ++i) { W.cPrior[i] = e->cPrior[i]; W.value_zero[i] = e->value_zero[i]; }
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {      // PRE_RTll_0 / PRE_tTll_0 (HessianBlocks.cpp:171-175), the same expressions as ef_upload_precalc
            double R[9];
            const gn::Pose l0 = gn::compose(e->frames[t].evalPT, gn::inverse(e->frames[h].evalPT));
            gn::rotation_matrix(l0.q, R);
            for (int i = 0; i < 9; ++i) W.l0[h * nF + t][i] = (float)R[i];
            for (int i = 0; i < 3; ++i) W.l0[h * nF + t][9 + i] = (float)l0.t[i];
        }
    for (int h = 0; h < nF; ++h) {
        const FrameH& f = e->frames[h];
        gn::pose_store(f.evalPT, W.fr[h].evalPT);
        for (int i = 0; i < 10; ++i) W.fr[h].state_zero[i] = f.state_zero[i];
        for (int i = 0; i < 6; ++i) W.fr[h].prior[i] = f.prior[i];
        W.fr[h].ab_exposure = f.ab_exposure; W.fr[h].frameID = f.frameID;
    }
    if ((int)e->HM.size() == n * n) {
        std::memcpy(W.HM, e->HM.data(), sizeof(double) * n * n); std::memcpy(W.bM, e->bM.data(), sizeof(double) * n);
        for (double v : e->HM) if (v != 0.0) { W.hm_nonzero = 1; break; }
        for (double v : e->bM) if (v != 0.0) { W.hm_nonzero = 1; break; }
    }
    if (k > 0) {
        ef_prepare_nullspace(e, n);
        std::memcpy(W.nsN, e->ns_N.data(), sizeof(double) * (size_t)n * k);
        std::memcpy(W.nsNpi, e->ns_Npi.data(), sizeof(double) * (size_t)n * k);
        // orthogonalize(&x, 0) (EnergyFunctional.cpp:615-648) subtracts 0.5 (N Npi^T + Npi N^T) x: that matrix, for the one-wave tail of the device solve
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double a = 0;
                for (int q = 0; q < k; ++q) a += e->ns_N[(size_t)i * k + q] * e->ns_Npi[(size_t)j * k + q] + e->ns_Npi[(size_t)i * k + q] * e->ns_N[(size_t)j * k + q];
                W.nsP[(size_t)i * n + j] = 0.5 * a;
            }
    }
    // (by a kernel reading the pinned block: for 68 kB the copy engine's start-up costs more than the transfer)
    static_assert(sizeof(SolveWindow) % 8 == 0, "copied as 8-byte words");
    k_ef_copy_in<<<16, 256, 0, e->stream>>>(reinterpret_cast<const unsigned long long*>(&W), reinterpret_cast<unsigned long long*>(e->win_dev), (int)(sizeof(SolveWindow) / 8),
                                          nullptr, nullptr, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->up_ev[0][half], e->stream));
    e->up_ev_used[0][half] = true;
    e->win_dirty = false;
    return 0;
}
static void ef_fill_calib(const sdvgn_ef* e, CalibDev& c) {
    c.fxl = e->C.fxl; c.fyl = e->C.fyl; c.cxl = e->C.cxl; c.cyl = e->C.cyl; c.fxli = e->C.fxli; c.fyli = e->C.fyli;
    for (int i = 0; i < 4; ++i) c.cDeltaF[i] = e->C.cDeltaF[i];
    c.pad[0] = c.pad[1] = 0;
}
// host mirror (calib value, frame states) -> the current device state set
static int ef_sync_state(sdvgn_ef* e) {
    if (!e->state_dirty) return 0;
    const int half = (e->st_flip ^= 1);       // (two pinned copies in turn, like ef_sync_window)
    if (e->up_ev_used[1][half]) HIPCHK(hipEventSynchronize(e->up_ev[1][half]));
    SolveState& S = e->sstate_host[half];
    for (int i = 0; i < 4; ++i) S.value[i] = e->value[i];
    for (int h = 0; h < e->nF; ++h) for (int i = 0; i < 10; ++i) S.state[h][i] = e->frames[h].state[i];
    ef_fill_calib(e, e->calib_host[half]);
    static_assert(sizeof(SolveState) % 8 == 0 && sizeof(CalibDev) % 8 == 0, "copied as 8-byte words");
    k_ef_copy_in<<<1, 256, 0, e->stream>>>(reinterpret_cast<const unsigned long long*>(&S), reinterpret_cast<unsigned long long*>(e->sstate_dev + e->st_cur), (int)(sizeof(SolveState) / 8),
                                         reinterpret_cast<const unsigned long long*>(e->calib_host + half), reinterpret_cast<unsigned long long*>(e->calib_dev + e->st_cur),
                                         (int)(sizeof(CalibDev) / 8));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->up_ev[1][half], e->stream));
    e->up_ev_used[1][half] = true;
    e->state_dirty = false;
    return 0;
}

// device -> pinned host by stores of the kernel itself: for a few hundred kB the copy engine's start-up costs more than the transfer
// (sdvgn_ef_optimize_finish's outputs: 256 kB at the named shape)
__global__ void __launch_bounds__(256) k_ef_copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst_pinned, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst_pinned[i] = src[i];
}

__global__ void __launch_bounds__(1024) k_ef_precalc_in(const unsigned long long* __restrict__ src_pinned, unsigned long long* __restrict__ dst, int n8) {
    for (int i = threadIdx.x; i < n8; i += 1024) dst[i] = src_pinned[i];
}

static void ef_refresh_frame_deltas(sdvgn_ef* e) {   // the host-side part of setDeltaF: FrameHessian delta / delta_prior
    for (FrameH& f : e->frames)
        for (int i = 0; i < 6; ++i) { f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
}

// setPrecalcValues + setDeltaF for the current frame states; the table goes to `dst` (default: the table the kernels read)
// stage_only: only fill the pinned staging half (e->precalc_staged); the caller's next kernel carries the copy
static int ef_upload_precalc(sdvgn_ef* e, PrecalcDev* dst = nullptr, bool stage_only = false) {
    const int nF = e->nF;
    const EFConst& C = e->C;
    const float K[9] = {C.fxl, 0, C.cxl, 0, C.fyl, C.cyl, 0, 0, 1};
    float Ki[9];
    gn::inverse3f(K, Ki);
    // the previous upload from the other half has completed: every caller synchronises the stream at least once between
    // two uploads (linearize read-back), so alternating halves is enough
    e->precalc_flip ^= 1;
    PrecalcDev* pch = e->precalc_host + (size_t)e->precalc_flip * SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES;
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            PrecalcDev& P = pch[h * nF + t];
            const FrameH& host = e->frames[h];
            const FrameH& target = e->frames[t];
            double R[9];
            const gn::Pose l0 = gn::compose(target.evalPT, gn::inverse(host.evalPT));
            gn::rotation_matrix(l0.q, R);
            for (int i = 0; i < 9; ++i) P.R0[i] = (float)R[i];
            for (int i = 0; i < 3; ++i) P.t0[i] = (float)l0.t[i];
            const gn::Pose l = gn::compose(target.PRE_worldToCam, host.PRE_camToWorld);
            gn::rotation_matrix(l.q, R);
            float Rf[9], tf[3], KR[9];
            for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
            for (int i = 0; i < 3; ++i) tf[i] = (float)l.t[i];
            m3f_mul(K, Rf, KR);
            m3f_mul(KR, Ki, P.KRKi);
            for (int i = 0; i < 3; ++i) P.Kt[i] = (K[i * 3] * tf[0] + K[i * 3 + 1] * tf[1]) + K[i * 3 + 2] * tf[2];
            double ab[2];
            gn::aff_from_to(host.ab_exposure, target.ab_exposure, host.state_scaled[6], host.state_scaled[7], target.state_scaled[6],
                            target.state_scaled[7], ab);
            P.aff0 = (float)ab[0]; P.aff1 = (float)ab[1];
            P.b0 = (float)(host.state_zero[7] * kScaleB);
            P.unused_th = 0;
            // setDeltaF: adHTdeltaF[h + t*nF]
            const float* AHf = &e->adHostF[(size_t)(h + t * nF) * 36];
            const float* ATf = &e->adTargetF[(size_t)(h + t * nF) * 36];
            float dh[6], dt[6];
            for (int i = 0; i < 6; ++i) { dh[i] = (float)(host.state[i] - host.state_zero[i]); dt[i] = (float)(target.state[i] - target.state_zero[i]); }
            for (int c = 0; c < 6; ++c) {
                float a = 0, b = 0;
                for (int q = 0; q < 6; ++q) { a += dh[q] * AHf[q * 6 + c]; b += dt[q] * ATf[q * 6 + c]; }
                P.dp[c] = a + b;
            }
            P.P0 = e->hostP0.empty() ? 0 : e->hostP0[h];
            P.np = e->hostP0.empty() ? 0 : e->hostP0[h + 1] - e->hostP0[h];
            if (e->shard_set && (h < e->h0 || h >= e->h1)) P.np = 0;   // not this rank's shard
        }
    ef_refresh_frame_deltas(e);
    e->precalc_staged = pch;
    if (!e->host_only && !stage_only) {
        // the table goes host -> device inside a one-workgroup kernel that reads the pinned staging half directly (one PCIe
        // round trip, ~3 us on the stream; a hipMemcpyAsync of the same 10 kB costs a ~8 us blit kernel plus its launch)
        static_assert(sizeof(PrecalcDev) % 8 == 0, "PrecalcDev is copied as 8-byte words");
        const int n8 = (int)(sizeof(PrecalcDev) * nF * nF / 8);
        k_ef_precalc_in<<<1, 1024, 0, e->stream>>>((const unsigned long long*)pch, (unsigned long long*)(dst ? dst : e->precalc_dev), n8);
        HIPCHK(hipGetLastError());
    }
    e->havePrecalc = true;
    return 0;
}

__global__ void k_ef_sum_energy(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double s[256];
    double a = 0;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = s[0];
}
// mode 0: backup = idepth ; 1: idepth = idepth_zero = backup + fac*step ; 2: idepth = idepth_zero = backup
__global__ void k_ef_point_step(int nP, int mode, float fac, float* __restrict__ pid, float* __restrict__ pidz, float* __restrict__ backup,
                                const float* __restrict__ step, float* __restrict__ pdeltaF) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nP) return;
    if (mode == 0) { backup[p] = pid[p]; return; }
    const float v = (mode == 1) ? backup[p] + fac * step[p] : backup[p];
    pid[p] = SDVGN_SCALE_IDEPTH * v;
    pidz[p] = SDVGN_SCALE_IDEPTH * v;
    pdeltaF[p] = v - v;   // idepth - idepth_zero
}


// PointFrameResidual::resetOOB for every non-linearised residual (start of FullSystem::optimize, :353-364)
__global__ void k_ef_reset_oob(size_t slots, EFArrays A, const uint8_t* __restrict__ mask = nullptr, int nP = 1) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    if (mask && !mask[s % nP]) return;
    const uint8_t fl = A.rflags[s];
    if (!(fl & RF_EXISTS) || (fl & RF_LINEARIZED)) return;
    A.renergy[s] = 0; A.renergy_new[s] = 0;
    A.rstate_new[s] = RS_OUTLIER;
    A.rstate[s] = RS_IN;
}

// per-block partial sums of calcLEnergyPt (EnergyFunctional.cpp:297-331); only launched when some residual is linearised
// or some point has deltaF != 0 (otherwise the sum is exactly 0)
__device__ __forceinline__ void point_stats_body(const EFConst& Cin, const EFArrays& A, const PrecalcDev* __restrict__ precalc,
                                                 const int* __restrict__ phost, double* __restrict__ partial, const int b) {
    const EFConst C = ef_const(Cin, A);
    __shared__ double sh[4];
    const int p = b * 256 + threadIdx.x;
    double e = 0;
    if (p < C.nP) {
        const int h = phost[p];
        if (precalc[h * C.nF + h].np != 0) {
            const size_t slots = (size_t)C.nF * C.nP;
            const float dd = A.pdeltaF[p];
            float acc = 0;
            for (int t = 0; t < C.nF; ++t) {
                const size_t s = (size_t)t * C.nP + p;
                const uint8_t fl = A.rflags[s];
                if (!(fl & RF_EXISTS) || !(fl & RF_ACTIVE) || !(fl & RF_LINEARIZED)) continue;
                const PrecalcDev& pc = precalc[h * C.nF + t];
                const float* Je = A.J + (size_t)((fl & RF_SEL) ? 1 : 0) * kJPlanes * slots + s;
                float dx = 0, dy = 0, cx = 0, cy = 0;
                for (int i = 0; i < 6; ++i) { dx += Je[(2 + i) * slots] * pc.dp[i]; dy += Je[(8 + i) * slots] * pc.dp[i]; }
                for (int i = 0; i < 4; ++i) { cx += Je[(14 + i) * slots] * C.cDeltaF[i]; cy += Je[(18 + i) * slots] * C.cDeltaF[i]; }
                const float jx = dx + cx + Je[22 * slots] * dd, jy = dy + cy + Je[23 * slots] * dd;
                const float r0 = A.rres_toZero[s], r1 = A.rres_toZero[slots + s];
                acc += (r0 * jx + r1 * jy) + (jx * r0 + jy * r1) + (jx * jx + jy * jy);
            }
            acc += dd * dd * A.ppriorF[p];
            e = acc;
        }
    }
    e = wave_sum_double(e);
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) partial[b] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void __launch_bounds__(256) k_ef_point_stats(EFConst Cin, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                       const int* __restrict__ phost, double* __restrict__ partial) {
    point_stats_body(Cin, A, precalc, phost, partial, (int)blockIdx.x);
}
// out[0] = sum energy partials, out[1] = sum L partials (nL may be 0), out[2], out[3] = sums of the two halves of the
// resubstitute partials (step^2, |idepth_backup|): one workgroup instead of four tiny launches
// The accept / reject test of FullSystem::optimize (FullSystemOptimize.cpp:420) taken where the sums arrive: the host hands over the parts
// it owns (the prior part of the L energy of the stepped state, the M energy, the right-hand side from the last accepted state) before the
// sums exist, the kernel finishes the comparison with the same double operations in the same order and leaves the verdict for the
// conditional k_ef_apply queued right behind it (accept_dev) and for the host (out[4]).
// en_em (device, may be NULL): the prior energy and the M energy of the stepped state as step_energy_body left them -- then En / EM of this block
// are ignored and the host need not have seen x when it queues the launch; the two values go back to the host in out[5], out[7]
__device__ __forceinline__ void sum_stats_body(const double* __restrict__ pe, int nE, const double* __restrict__ pl, int nL,
                                               const double* __restrict__ ps, int nS, double* __restrict__ out, volatile int* done_flag, int done_seq,
                                               double (*s)[256], const DecideArgs& dec = DecideArgs{0, 0, 0, nullptr, 0, nullptr, 0, nullptr}) {
    if (threadIdx.x < 256) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        // all loads of this lane in one batch (clamped indices, the out-of-range ones add 0), the sums in index order afterwards: one
        // memory round trip instead of one per stride -- this workgroup is what every apply workgroup of k_ef_stats_apply waits for
        constexpr int kB = 8;
        double ve[kB], vs2[2], vs3[2];
#pragma unroll
        for (int k = 0; k < kB; ++k) { const int i = threadIdx.x + 256 * k; ve[k] = pe[i < nE ? i : 0]; }
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = threadIdx.x + 256 * k; vs2[k] = ps[i < nS ? i : 0]; vs3[k] = ps[nS + (i < nS ? i : 0)]; }
#pragma unroll
        for (int k = 0; k < kB; ++k) if ((int)threadIdx.x + 256 * k < nE) a0 += ve[k];
        for (int i = threadIdx.x + 256 * kB; i < nE; i += 256) a0 += pe[i];
        for (int i = threadIdx.x; i < nL; i += 256) a1 += pl[i];
#pragma unroll
        for (int k = 0; k < 2; ++k) if ((int)threadIdx.x + 256 * k < nS) { a2 += vs2[k]; a3 += vs3[k]; }
        for (int i = threadIdx.x + 512; i < nS; i += 256) { a2 += ps[i]; a3 += ps[nS + i]; }
        // fixed order: 256 strided lane sums, a 64-lane tree per wave (DPP, no barrier), the four wave sums in index order
        a0 = wave_sum_double(a0); a1 = wave_sum_double(a1); a2 = wave_sum_double(a2); a3 = wave_sum_double(a3);
        if ((threadIdx.x & 63) == 63) { const int w = threadIdx.x >> 6; s[0][w] = a0; s[1][w] = a1; s[2][w] = a2; s[3][w] = a3; }
    }
    __syncthreads();
    if (threadIdx.x < 4) { const int q = threadIdx.x; s[q][0] = ((s[q][0] + s[q][1]) + s[q][2]) + s[q][3]; out[q] = s[q][0]; }
    __syncthreads();
    if (dec.on && threadIdx.x == 0) {
        const double En = dec.en_em ? dec.en_em[0] : dec.En, EM = dec.en_em ? dec.en_em[1] : dec.EM;
        const double newEnergy = s[0][0];
        const double newEnergyL = En + (double)(float)s[1][0];                      // linearize_wait's expression
        const bool accept = (newEnergy + newEnergyL) + EM < dec.rhs;
        out[5] = En; out[7] = EM;
        // first of all the verdict for the apply workgroups of the same launch (k_ef_stats_apply): one tagged word, relaxed device-scope store
        if (dec.verdict) __hip_atomic_store(dec.verdict, (dec.seq << 1) | (accept ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dec.accept_dev = accept ? 1 : 0;
        out[4] = accept ? 1.0 : 0.0;
    }
    if (done_flag) {   // single workgroup: publish after the four stores (waitflag.hpp)
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence_system(); *done_flag = done_seq; }
    }
}
__global__ void __launch_bounds__(256) k_ef_sum_stats(const double* __restrict__ pe, int nE, const double* __restrict__ pl, int nL,
                                                      const double* __restrict__ ps, int nS, double* __restrict__ out, volatile int* done_flag, int done_seq) {
    __shared__ double s[4][256];
    sum_stats_body(pe, nE, pl, nL, ps, nS, out, done_flag, done_seq, s);
}
// single-rank tail of a linearizeAll in ONE launch: workgroup 0 = the four statistics (+ flag for the host), workgroup 1 =
// FullSystem::setNewFrameEnergyTH (k_ef_select_th's body) -- they are independent and run side by side on two CUs
__global__ void __launch_bounds__(kSelLanes) k_ef_stats_select(const double* __restrict__ pe, int nE, const double* __restrict__ pl, int nL,
                                                                const double* __restrict__ ps, int nS, double* __restrict__ out, volatile int* done_flag,
                                                                int done_seq, SelArgs a, DecideArgs dec) {
    __shared__ union U { double s[4][256]; SelectSmem sel; __device__ U() {} } S;
    if (blockIdx.x == 0) sum_stats_body(pe, nE, pl, nL, ps, nS, out, done_flag, done_seq, S.s, dec);
    else select_th_body<0>(a.nF, a.nP, a.own0, a.own1, a.rflags, a.wo, nullptr, a.th_prev, a.th_out, a.log_slot, S.sel);
}

// The statistics of a trial linearizeAll, the accept test and applyRes in ONE launch (optimize loop, single rank): workgroup 0 sums, decides,
// publishes the verdict word and then the sums + flag for the host; every other workgroup is k_ef_apply for 256 slots -- its loads in
// flight while the verdict is being formed, its stores only if the verdict says accept.  Workgroup 0 is dispatched first and the whole
// grid is resident at once (2 waves of 256-lane workgroups on 256 CUs), so the poll cannot starve it.
__global__ void __launch_bounds__(256) k_ef_stats_apply(const double* __restrict__ pe, int nE, const double* __restrict__ pl, int nL,
                                                        const double* __restrict__ ps, int nS, double* __restrict__ out, volatile int* done_flag,
                                                        int done_seq, DecideArgs dec, int nF, int nP, EFArrays A, const PrecalcDev* __restrict__ precalc,
                                                        const int* __restrict__ phost) {
    if (blockIdx.x == 0) {
        __shared__ double s[4][256];
        sum_stats_body(pe, nE, pl, nL, ps, nS, out, done_flag, done_seq, s, dec);
        return;
    }
    apply_slot(nF, nP, A, precalc, phost, (size_t)(blockIdx.x - 1) * 256 + threadIdx.x, nullptr, dec.verdict, dec.seq);
}

// The loop's LAST body by count: statistics + accept test | conditional applyRes | setNewFrameEnergyTH side by side in ONE launch (no later
// launch could carry the deferred select).  The select reads the EXISTS / LINEARIZED bits of the flags and the energies the linearise
// wrote; applyRes changes neither, so the two need no order.  Workgroup 0 = sums + verdict, the last workgroup = the select, the others
// k_ef_apply for kSelLanes slots each (the whole grid is resident at once).
__global__ void __launch_bounds__(kSelLanes) k_ef_stats_apply_select(const double* __restrict__ pe, int nE, const double* __restrict__ pl, int nL,
                                                                      const double* __restrict__ ps, int nS, double* __restrict__ out, volatile int* done_flag,
                                                                      int done_seq, DecideArgs dec, int nF, int nP, EFArrays A,
                                                                      const PrecalcDev* __restrict__ precalc, const int* __restrict__ phost, SelArgs a) {
    __shared__ union U { double s[4][256]; SelectSmem sel; __device__ U() {} } S;
    if (blockIdx.x == 0) { sum_stats_body(pe, nE, pl, nL, ps, nS, out, done_flag, done_seq, S.s, dec); return; }
    if (blockIdx.x == gridDim.x - 1) { select_th_body<0>(a.nF, a.nP, a.own0, a.own1, a.rflags, a.wo, nullptr, a.th_prev, a.th_out, a.log_slot, S.sel); return; }
    apply_slot(nF, nP, A, precalc, phost, (size_t)(blockIdx.x - 1) * kSelLanes + threadIdx.x, nullptr, dec.verdict, dec.seq);
}

// The accept test of a trial step and the NEXT body's accumulate in ONE launch (optimize loop, one rank, applyRes fused into the linearise, the rejected
// case solved ahead on the side stream): the last workgroup = sums + verdict (k_ef_stats_select's first workgroup), the others k_ef_acc_fused on the accepted case's
// arguments.  The accumulate's inputs do not depend on the verdict (the linearise wrote the applied copies), its Gram tiles are scratch that only a reduce
// behind an "accept" reads; the per-point planes are the one thing a rejected step must find untouched -- point_body's first wave looks at the tagged
// verdict word before it stores (by then it is there: one round trip + a 900-term sum against two round trips + the per-point sums).  The statistics launch
// (5.4 us of the 62 us chain of an accepted body) leaves the chain.
__global__ void __launch_bounds__(256) k_ef_acc_stats(const PrecalcDev* __restrict__ precalc, EFConst Cin, EFArrays A, const int* __restrict__ phost,
                                                      float* __restrict__ top_partial, int* __restrict__ nres_partial, int top_chunks,
                                                      float* __restrict__ sc_partial, int sc_chunks, int n_sc, AccAlt alt, StatsLaunch st) {
    __shared__ AccSmem S;
    // workgroup 0 = the sums (dispatched first: the verdict is out ~1 us earlier than from the grid's last workgroup); 1..7 leave at once; the accumulate's workgroups
    // follow from id 8 on, i.e. they keep the XCDs k_ef_acc_fused gives them (id mod 8), next to the linearise workgroups whose Jacobians they read.  The whole grid is
    // resident at once (4 workgroups per CU, the sums in the accumulate's own LDS).
    if (blockIdx.x < 8) {
        if (blockIdx.x == 0) sum_stats_body(st.pe, st.nE, st.pl, st.nL, st.ps, st.nS, st.out, st.done_flag, st.done_seq, S.stats, st.dec);
        return;
    }
    acc_fused_body(S, precalc, Cin, A, phost, top_partial, nres_partial, top_chunks, sc_partial, sc_chunks, n_sc, alt, (int)blockIdx.x - 8);
}

// device buffer -> pinned host buffer + completion flag (waitflag.hpp): the read-back after an all-reduce without the copy engine
__global__ void __launch_bounds__(256) k_ef_copy_publish(const double* __restrict__ src, double* __restrict__ dst, int n, unsigned* __restrict__ ctr,
                                                         volatile int* flag, int seq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) publish_when_all_done(ctr, gridDim.x, flag, seq);
}

static int lin_groups(const sdvgn_ef* e) { return (e->C.debug_flags & 256) ? 1 : 2; }   // residual groups (64 residuals x 2 role waves) per workgroup
static int lin_chunks_for_np(const sdvgn_ef* e) {   // k_ef_linearize: 64 x groups residuals per workgroup
    int mx = 1;
    for (int h = 0; h < e->nF; ++h) mx = std::max(mx, e->hostP0[h + 1] - e->hostP0[h]);
    const int per = 64 * lin_groups(e);
    return std::min(kMaxChunks * 256 / per, (mx + per - 1) / per);
}
// launches PointFrameResidual::linearize over the window; returns the number of energy partials written (128-residual granularity)
static int ef_launch_linearize(sdvgn_ef* e) {
    const int pairs = e->nF * e->nF;
    const int chunks = lin_chunks_for_np(e);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e->time_lin) {   // measurement mode (optimize flags bit3): one HIP event pair around this launch, read back after the loop
        if (e->lin_ev_used + 2 > e->lin_events.size()) { e->lin_events.resize(e->lin_ev_used + 2, nullptr); }
        for (int k = 0; k < 2; ++k) if (!e->lin_events[e->lin_ev_used + k]) hipEventCreate(&e->lin_events[e->lin_ev_used + k]);
        ev0 = e->lin_events[e->lin_ev_used]; ev1 = e->lin_events[e->lin_ev_used + 1];
        e->lin_ev_used += 2;
        hipEventRecord(ev0, e->stream);
    }
    if ((e->C.debug_flags & 32) && e->dbg_stamps) k_ef_linearize<true><<<dim3(chunks, pairs), 256, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->energy_partial);
    else if (lin_groups(e) == 1) k_ef_linearize<false, 1><<<dim3(chunks, pairs), 128, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->energy_partial);
    else if (e->arith == 1) lin_fast::k_ef_linearize<false><<<dim3(chunks, pairs), 256, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->energy_partial);
    else k_ef_linearize<false><<<dim3(chunks, pairs), 256, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->energy_partial);
    if (ev1) hipEventRecord(ev1, e->stream);
    return chunks * pairs;
}
static int chunks_for_np(const sdvgn_ef* e) {
    int mx = 1;
    for (int h = 0; h < e->nF; ++h) mx = std::max(mx, e->hostP0[h + 1] - e->hostP0[h]);
    return std::min(kMaxChunks, (mx + 255) / 256);
}

// ---- RCCL, resolved at run time ---------------------------------------------------------------------------------
// PyTorch-ROCm ships its own librccl.so; when it is already in the process (torch.distributed "nccl") the SAME instance must be
// used, so RTLD_NOLOAD is tried first, then the normal search path (/opt/rocm/lib).
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};
static RcclApi& rccl_api() {
    static RcclApi api;
    if (api.lib || api.ok) return api;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) return api;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy;
    return api;
}
// One communicator per (process, id): every handle initialised with the same 128-byte id shares it (reference-counted) -- a process that
// optimises many sharded windows pays ncclCommInitRank once, not once per window.  The last handle to let go destroys it.
namespace {
struct SharedComm { unsigned char id[NCCL_UNIQUE_ID_BYTES]; ncclComm_t comm; int refs; int rank, world, device; };
std::mutex g_comm_mu;
std::vector<SharedComm> g_comms;
}
static void ef_release_comm(sdvgn_ef* e) {
    if (!e->rccl_comm) return;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (size_t i = 0; i < g_comms.size(); ++i)
        if (g_comms[i].comm == e->rccl_comm) {
            if (--g_comms[i].refs == 0) { rccl_api().CommDestroy(g_comms[i].comm); g_comms.erase(g_comms.begin() + (long)i); }
            break;
        }
    e->rccl_comm = nullptr;
}
static inline bool ef_sharded(const sdvgn_ef* e) { return e->allreduce != nullptr || e->rccl_comm != nullptr; }
// sum `count` doubles at buf_dev over all ranks, ordered on the library stream
static int ef_allreduce(sdvgn_ef* e, double* buf_dev, int count) {
    ++e->n_collectives;
    if (e->rccl_comm) {
        const ncclResult_t r = rccl_api().AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, e->rccl_comm, e->stream);
        return r == ncclSuccess ? SDVGN_OK : SDVGN_E_STATE;
    }
    if (e->allreduce) e->allreduce(e->allreduce_user, buf_dev, count);
    return SDVGN_OK;
}

// FullSystem::setNewFrameEnergyTH for the linearisation just launched (its state_NewEnergyWithOutlier plane is A.renergy_wo):
// the new threshold set goes to A.frameTH_w.  Single rank: one launch, nothing to wait for (the next linearise follows on the stream).
// Sharded: the candidates of all ranks arrive through stats_dev[4 ..] (see linearize_and_stats).
static void ef_owned_points(const sdvgn_ef* e, int& own0, int& own1) {   // point range hosted by this rank's key-frames
    const int h0 = e->shard_set ? std::min(e->h0, e->nF) : 0, h1 = e->shard_set ? std::min(e->h1, e->nF) : e->nF;
    own0 = e->hostP0.empty() ? 0 : e->hostP0[h0];
    own1 = e->hostP0.empty() ? 0 : e->hostP0[std::max(h0, h1)];
}
static void ef_launch_select_th(sdvgn_ef* e, bool from_reduced) {
    float* log_slot = e->th_log ? e->th_log + (e->th_log_n++ % kThLog) : nullptr;
    int own0, own1;
    ef_owned_points(e, own0, own1);
    if (from_reduced)
        k_ef_select_th<1><<<1, kSelLanes, 0, e->stream>>>(e->nF, e->nP, 0, e->nP, nullptr, nullptr, e->stats_dev + 4, e->A.frameTH_r, e->A.frameTH_w, log_slot);
    else
        k_ef_select_th<0><<<1, kSelLanes, 0, e->stream>>>(e->nF, e->nP, own0, own1, e->rflags, e->A.renergy_wo, nullptr, e->A.frameTH_r, e->A.frameTH_w, log_slot);
}
static void ef_launch_pack_th(sdvgn_ef* e) {
    int own0, own1;
    ef_owned_points(e, own0, own1);
    k_ef_pack_th_candidates<<<(e->nP + 255) / 256, 256, 0, e->stream>>>(e->nF, e->nP, own0, own1, e->rflags, e->A.renergy_wo, e->stats_dev + 4);
}

// launch what the optimize loop deferred and nothing has picked up: the select first (the re-classification reads its threshold)
static void ef_flush_pending(sdvgn_ef* e) {
    if (e->pend_sel_valid) {
        const SelArgs& a = e->pend_sel;
        k_ef_select_th<0><<<1, kSelLanes, 0, e->stream>>>(a.nF, a.nP, a.own0, a.own1, a.rflags, a.wo, nullptr, a.th_prev, a.th_out, a.log_slot);
        e->pend_sel_valid = false;
    }
    if (e->pend_rc_valid) {
        const int nthr = e->pend_rc.nP + e->pend_rc.np_last * (e->pend_rc.nF - 1);
        k_ef_reclassify<<<(nthr + 255) / 256, 256, 0, e->stream>>>(e->pend_rc);
        e->pend_rc_valid = false;
    }
}

extern "C" {

int sdvgn_ef_create(sdvgn_ef** out, int device, int w, int h, int max_points, void* stream) {
    if (!out || w < 16 || h < 16 || max_points < 1) return SDVGN_E_ARG;
    if (device < 0) {
        // host-only handle: frames / adjoints / priors / sdvgn_ef_stitch_solve_host work, nothing that launches a kernel does.
        // Used by the rank that only combines all-reduced accumulators and by the CPU (gloo) tests of the multi-GPU host logic.
        sdvgn_ef* e = new (std::nothrow) sdvgn_ef();
        if (!e) return SDVGN_E_ARG;
        e->device = -1; e->w = w; e->h = h; e->max_points = max_points; e->host_only = true;
        e->precalc_host = (PrecalcDev*)calloc(2 * SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES, sizeof(PrecalcDev));
        *out = e;
        return SDVGN_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return SDVGN_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    sdvgn_ef* e = new (std::nothrow) sdvgn_ef();
    if (!e) return SDVGN_E_ARG;
    e->device = device; e->w = w; e->h = h; e->max_points = max_points;
    if (stream == SDVGN_STREAM_OWN) {   // a stream (hardware queue) of this handle's own: windows optimised side by side, sdvgn_ef_optimize_batch
        HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    } else if (stream) e->stream = (hipStream_t)stream;
    else {
        // Windows without a caller's stream share ONE library stream per device.  A stream of its own per window handle means a
        // hardware queue per handle, and a queue that has been idle for a while is re-activated by the first launch that lands on it:
        // measured 0.5 ms for the first kernel of the first optimize call on such a handle, 0.16 ms on the next one (bench.py, four
        // windows created ahead of time) -- against 0.03 ms for the whole initial linearizeAll.  Calls on one device are issued one after
        // the other anyway; a caller that wants two windows to overlap passes two streams.
        static std::mutex mu;
        static hipStream_t shared[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        if (device < 0 || device >= 64) { delete e; return SDVGN_E_ARG; }
        if (!shared[device]) HIPCHK(hipStreamCreateWithFlags(&shared[device], hipStreamNonBlocking));
        e->stream = shared[device];
    }
    const size_t mp = max_points, slots = mp * SDVGN_MAX_FRAMES;
    e->slots_cap = slots;
    int bad = 0;
    bad |= dev_alloc(&e->pu, mp) | dev_alloc(&e->pv, mp) | dev_alloc(&e->pidz, mp) | dev_alloc(&e->pid, mp) | dev_alloc(&e->pidepth_backup, mp);
    bad |= dev_alloc(&e->ppriorF, mp) | dev_alloc(&e->pdeltaF, mp) | dev_alloc(&e->pcolor, 2 * mp) | dev_alloc(&e->pweights, 2 * mp) | dev_alloc(&e->psensor, mp);
    bad |= dev_alloc(&e->rflags, slots) | dev_alloc(&e->rstate, slots) | dev_alloc(&e->rstate_new, slots) | dev_alloc(&e->rmatcher, slots);
    bad |= dev_alloc(&e->renergy, slots) | dev_alloc(&e->renergy_new, slots) | dev_alloc(&e->renergy_wo, slots) | dev_alloc(&e->rres_toZero, 2 * slots);
    bad |= dev_alloc(&e->rstate_new2, slots) | dev_alloc(&e->renergy_new2, slots) | dev_alloc(&e->renergy_wo2, slots);
    bad |= dev_alloc(&e->J, 2 * (size_t)kJPlanes * slots) | dev_alloc(&e->JpJd, 6 * slots);
    bad |= dev_alloc(&e->rflags_alt, slots) | dev_alloc(&e->rstate_alt, slots) | dev_alloc(&e->renergy_alt, slots) | dev_alloc(&e->JpJd_alt, 6 * slots);
    bad |= dev_alloc(&e->pHddA, mp) | dev_alloc(&e->pbdA, mp) | dev_alloc(&e->pHcdA, 4 * mp) | dev_alloc(&e->pHddL, mp) | dev_alloc(&e->pbdL, mp) | dev_alloc(&e->pHcdL, 4 * mp);
    bad |= dev_alloc(&e->pHdi, mp) | dev_alloc(&e->pbdSum, mp) | dev_alloc(&e->pHcd, 4 * mp) | dev_alloc(&e->pstep, mp) | dev_alloc(&e->pnogood, mp);
    bad |= dev_alloc(&e->images, (size_t)SDVGN_MAX_FRAMES * w * h * 3) | dev_alloc(&e->img_stage, (size_t)w * h);
    bad |= dev_alloc(&e->phost_dev, mp) | dev_alloc(&e->hostP0_dev, SDVGN_MAX_FRAMES + 1);
    bad |= dev_alloc(&e->precalc_dev, SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES) | dev_alloc(&e->precalc_alt, SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES);
    bad |= dev_alloc(&e->pid_alt, mp) | dev_alloc(&e->pidz_alt, mp) | dev_alloc(&e->pdeltaF_alt, mp);
    bad |= dev_alloc(&e->marg_mask_dev, mp) | dev_alloc(&e->drop_mask_dev, mp);
    if (getenv("SDVGN_DEBUG_FLAGS") && (atoi(getenv("SDVGN_DEBUG_FLAGS")) & 32)) {
        bad |= dev_alloc(&e->dbg_stamps, kDbgStampWords);
        if (!bad) hipMemset(e->dbg_stamps, 0, sizeof(unsigned long long) * kDbgStampWords);
    }
    if (getenv("SDVGN_DEBUG_FLAGS") && (atoi(getenv("SDVGN_DEBUG_FLAGS")) & 64)) {
        bad |= SDVGN_HMALLOC((void**)&e->solve_stamps, sizeof(unsigned long long) * 16) != hipSuccess;
        if (!bad) std::memset(e->solve_stamps, 0, sizeof(unsigned long long) * 16);
    }
    bad |= dev_alloc(&e->energy_partial, (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kMaxChunks * 4);
    bad |= dev_alloc(&e->top_partial, (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kMaxChunks * kTopP);
    bad |= dev_alloc(&e->sc_partial, (size_t)SDVGN_MAX_FRAMES * kMaxChunks * kScP);
    bad |= dev_alloc(&e->nres_partial, (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kMaxChunks);
    bad |= dev_alloc(&e->imm_pc_dev, (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES);
    const size_t accmax = (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kTopE + (size_t)SDVGN_MAX_FRAMES * kScE + 1;
    bad |= dev_alloc(&e->acc_dev, accmax);
    e->stats_cap = 4 + mp;
    bad |= dev_alloc(&e->stats_dev, e->stats_cap) | dev_alloc(&e->stats_partial, 3 * (mp / 64 + 2));
    bad |= dev_alloc(&e->th_dev, 2 * SDVGN_MAX_FRAMES);
    bad |= dev_alloc(&e->win_dev, 1) | dev_alloc(&e->sstate_dev, 2) | dev_alloc(&e->calib_dev, 2) | dev_alloc(&e->rx_dev, 1) | dev_alloc(&e->sys_dev, 1);
    bad |= dev_alloc(&e->pieces_dev, SDVGN_MAX_FRAMES);
    bad |= dev_alloc(&e->en_em_dev, 4);
    if (bad) { sdvgn_ef_destroy(e); return -(int)hipErrorOutOfMemory; }
    HIPCHK(SDVGN_HMALLOC(&e->precalc_host, 2 * sizeof(PrecalcDev) * SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES));
    HIPCHK(SDVGN_HMALLOC(&e->acc_host, sizeof(double) * accmax));
    HIPCHK(SDVGN_HMALLOC(&e->stats_host, sizeof(double) * 8));
    std::memset(e->stats_host, 0, sizeof(double) * 8);   // [0..3] sums, [4] verdict, [6] (as unsigned) the sticky intra-launch wait error word
    HIPCHK(SDVGN_DMALLOC((void**)&e->accept_dev, 64));
    HIPCHK(hipMemset(e->accept_dev, 0, 64));
    HIPCHK(hipMemset(e->pnogood, 0, (size_t)e->max_points));      // (epochs, ef_next_nogood_epoch: the plane starts clean)
    HIPCHK(SDVGN_HMALLOC((void**)&e->flags_host, 64));
    HIPCHK(SDVGN_HMALLOC((void**)&e->th_log, sizeof(float) * kThLog));
    HIPCHK(SDVGN_HMALLOC((void**)&e->win_host, 2 * sizeof(SolveWindow)));
    HIPCHK(SDVGN_HMALLOC((void**)&e->sstate_host, 2 * sizeof(SolveState)));
    HIPCHK(SDVGN_HMALLOC((void**)&e->calib_host, 2 * sizeof(CalibDev)));
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&e->up_ev[a][b], hipEventDisableTiming));
    HIPCHK(SDVGN_HMALLOC((void**)&e->sol_host, sizeof(SolveOut)));
    std::memset(e->sol_host, 0, sizeof(SolveOut));
    HIPCHK(SDVGN_HMALLOC((void**)&e->sol_spec, 2 * sizeof(SolveOut)));
    std::memset(e->sol_spec, 0, 2 * sizeof(SolveOut));
    HIPCHK(SDVGN_DMALLOC((void**)&e->rx_spec, sizeof(ResubX)));
    HIPCHK(hipMemset(e->rx_spec, 0, sizeof(ResubX)));
    HIPCHK(SDVGN_DMALLOC(&e->xw_spec, sizeof(unsigned long long) * 2 * 512));
    HIPCHK(hipMemset(e->xw_spec, 0, sizeof(unsigned long long) * 2 * 512));
    {   // one side stream per device, shared like the main one (handles on the shared stream are used one after the other)
        static std::mutex mu2;
        static hipStream_t side_streams[64] = {};
        std::lock_guard<std::mutex> lk(mu2);
        if (device < 64) {
            if (!side_streams[device]) HIPCHK(hipStreamCreateWithFlags(&side_streams[device], hipStreamNonBlocking));
            e->side = side_streams[device];
        }
    }
    HIPCHK(hipMemset(e->rx_dev, 0, sizeof(ResubX)));
    HIPCHK(SDVGN_DMALLOC(&e->xw_dev, sizeof(unsigned long long) * 512));
    HIPCHK(hipMemset(e->xw_dev, 0, sizeof(unsigned long long) * 512));   // tag 0 is never current (the solves count from 1)
    for (int i = 0; i < 16; ++i) e->flags_host[i] = 0;
    HIPCHK(SDVGN_DMALLOC((void**)&e->done_ctr, 2 * sizeof(unsigned)));
    HIPCHK(hipMemset(e->done_ctr, 0, 2 * sizeof(unsigned)));
    HIPCHK(SDVGN_HMALLOC((void**)&e->imm_pc_host, sizeof(ImmPrecalc) * SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES));
    HIPCHK(hipMemsetAsync(e->rflags, 0, slots, e->stream));
    HIPCHK(hipMemsetAsync(e->stats_partial, 0, sizeof(double) * 3 * (mp / 64 + 2), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->win = win_new();
    ef_fill_arrays(e);
    *out = e;
    return SDVGN_OK;
}

void sdvgn_ef_destroy(sdvgn_ef* e) {
    if (!e) return;
    if (e->host_only) { free(e->precalc_host); delete e; return; }
    if (!e->own_acc) e->acc_dev = nullptr;
    if (!e->own_stats) e->stats_dev = nullptr;
    hipSetDevice(e->device);
    hipStreamSynchronize(e->stream);
    if (e->side) hipStreamSynchronize(e->side);          // a speculative solve may still be reading this handle's system
    ef_release_comm(e);
    void* ptrs[] = {e->pu, e->pv, e->pidz, e->pid, e->pidepth_backup, e->ppriorF, e->pdeltaF, e->pcolor, e->pweights, e->psensor, e->rflags,
                    e->rstate, e->rstate_new, e->rmatcher, e->renergy, e->renergy_new, e->renergy_wo, e->rres_toZero, e->J, e->JpJd, e->pHddA,
                    e->pbdA, e->pHcdA, e->pHddL, e->pbdL, e->pHcdL, e->pHdi, e->pbdSum, e->pHcd, e->pstep, e->pnogood, e->images, e->img_stage,
                    e->phost_dev, e->hostP0_dev, e->precalc_dev, e->energy_partial, e->top_partial, e->sc_partial, e->nres_partial, e->acc_dev,
                    e->stats_dev, e->stats_partial, e->imm_pc_dev, e->rstate_new2, e->renergy_new2, e->renergy_wo2,
                    e->pid_alt, e->pidz_alt, e->pdeltaF_alt, e->precalc_alt, e->dbg_stamps, e->marg_mask_dev, e->drop_mask_dev, e->th_dev, e->fin_dev,
                    e->win_dev, e->sstate_dev, e->calib_dev, e->rx_dev, e->xw_dev, e->sys_dev, e->pieces_dev, e->en_em_dev,
                    e->rflags_alt, e->rstate_alt, e->renergy_alt, e->JpJd_alt};
    for (void* p : ptrs) if (p) SDVGN_DFREE(p);
    if (e->precalc_host) SDVGN_HFREE(e->precalc_host);
    if (e->acc_host) SDVGN_HFREE(e->acc_host);
    if (e->stats_host) SDVGN_HFREE(e->stats_host);
    if (e->accept_dev) SDVGN_DFREE(e->accept_dev);
    if (e->flags_host) SDVGN_HFREE(e->flags_host);
    if (e->th_log) SDVGN_HFREE(e->th_log);
    if (e->win_host) SDVGN_HFREE(e->win_host);
    if (e->sstate_host) SDVGN_HFREE(e->sstate_host);
    if (e->calib_host) SDVGN_HFREE(e->calib_host);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (e->up_ev[a][b]) hipEventDestroy(e->up_ev[a][b]);
    if (e->sol_host) SDVGN_HFREE(e->sol_host);
    if (e->sol_spec) SDVGN_HFREE(e->sol_spec);
    if (e->rx_spec) SDVGN_DFREE(e->rx_spec);
    if (e->xw_spec) SDVGN_DFREE(e->xw_spec);
    for (hipEvent_t ev : e->lin_events) if (ev) hipEventDestroy(ev);
    if (e->solve_stamps) SDVGN_HFREE(e->solve_stamps);
    if (e->done_ctr) SDVGN_DFREE(e->done_ctr);
    if (e->imm_pc_host) SDVGN_HFREE(e->imm_pc_host);
    if (e->imm_stage) SDVGN_HFREE(e->imm_stage);
    if (e->fin_host) SDVGN_HFREE(e->fin_host);
    if (e->jstage_dev) SDVGN_DFREE(e->jstage_dev);
    if (e->own_coll && e->coll[0]) SDVGN_DFREE(e->coll[0]);
    if (e->apply_bak.fl) { SDVGN_DFREE(e->apply_bak.fl); SDVGN_DFREE(e->apply_bak.st); SDVGN_DFREE(e->apply_bak.en); SDVGN_DFREE(e->apply_bak.JpJd); }
    win_delete(e->win);
    if (e->own_stream) hipStreamDestroy(e->stream);
    delete e;
}

void* sdvgn_ef_stream(sdvgn_ef* e) { return e ? (void*)e->stream : nullptr; }
const float* sdvgn_ef_frame_image_dev(sdvgn_ef* e, int idx) {
    if (!e || e->host_only || idx < 0 || idx >= e->nF || !e->images) return nullptr;
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return nullptr;   // uploads / pyramid kernels done
    return e->images + (size_t)e->img_slot[idx] * e->C.w * e->C.h * 3;
}

int sdvgn_ef_set_calib(sdvgn_ef* e, const double vs[4], const double vmz[4]) {
    if (!e || !vs || !vmz) return SDVGN_E_ARG;
    for (int i = 0; i < 4; ++i) { e->value_scaled[i] = vs[i]; e->value_minus_value_zero[i] = vmz[i]; }
    // CalibHessian::setValueScaled: value = SCALE_*_INVERSE * value_scaled (HessianBlocks.h:318-330)
    e->value[0] = (1.0f / SDVGN_SCALE_F) * vs[0]; e->value[1] = (1.0f / SDVGN_SCALE_F) * vs[1];
    e->value[2] = (1.0f / SDVGN_SCALE_C) * vs[2]; e->value[3] = (1.0f / SDVGN_SCALE_C) * vs[3];
    for (int i = 0; i < 4; ++i) e->value_zero[i] = e->value[i] - vmz[i];
    ef_update_const(e);
    e->havePrecalc = false;
    e->win_dirty = e->state_dirty = true; e->sys_valid = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_frames(sdvgn_ef* e, int nF, const double* evalPT7, const double* state10, const double* state_zero10,
                        const int* frameID, const float* ab_exposure, const float* frameEnergyTH) {
    if (!e || nF < 1 || nF > SDVGN_MAX_FRAMES || !evalPT7 || !state10 || !state_zero10 || !frameID || !ab_exposure || !frameEnergyTH)
        return SDVGN_E_ARG;
    e->nF = nF;
    if (e->stats_host) *reinterpret_cast<volatile unsigned*>(e->stats_host + 6) = 0;   // a new window: the sticky wait-error word starts clean
    e->frames.resize(nF);
    for (int i = 0; i < nF; ++i) {
        FrameH& f = e->frames[i];
        gn::pose_load(f.evalPT, evalPT7 + 7 * i);
        for (int k = 0; k < 10; ++k) f.state_zero[k] = state_zero10[10 * i + k];
        frame_set_state(f, state10 + 10 * i);
        f.frameID = frameID[i]; f.ab_exposure = ab_exposure[i]; f.frameEnergyTH = frameEnergyTH[i];
        for (int k = 0; k < 6; ++k) f.prior[k] = 0;   // FrameHessian::getPrior, HessianBlocks.h:220-250
        if (f.frameID == 0) { for (int k = 0; k < 3; ++k) f.prior[k] = kInitialTransPrior; for (int k = 3; k < 6; ++k) f.prior[k] = kInitialRotPrior; }
    }
    const int n = CPARS + 6 * nF;
    // the marginalisation prior belongs to a window layout: a new frame set starts without one (documented in sdvgn.h; the caller
    // re-installs the prior it carried over with sdvgn_ef_set_marg_prior)
    e->HM.assign((size_t)n * n, 0); e->bM.assign(n, 0);
    // the point / residual tables are laid out for the previous frame count (slot = target*nP + p, hostP0[nF+1]): invalidate them,
    // so that nothing can run on a stale layout before sdvgn_ef_set_points / _set_residuals are called again
    e->nP = 0; e->nR = -1;
    e->hostP0.clear(); e->phost.clear(); e->r_slot.clear();
    // a reload through the whole-plane setters: frame k's image is slot k again, point ids / frame uids of an edited window are forgotten
    for (int t = 0; t < SDVGN_MAX_FRAMES; ++t) e->img_slot[t] = (uint8_t)t;
    e->A.img_slots = 0x76543210u;
    e->table_mode = false;
    if (e->win) win_forget(e->win);
    e->win_dirty = e->state_dirty = true; e->sys_valid = false;
    if (!e->host_only) {
        float th[2 * SDVGN_MAX_FRAMES] = {0};
        for (int i = 0; i < nF; ++i) th[i] = th[SDVGN_MAX_FRAMES + i] = frameEnergyTH[i];
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipMemcpyAsync(e->th_dev, th, sizeof(th), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->new_cur = 0;
        ef_select_new_set(e, 0, 0);
    }
    ef_update_const(e);
    e->havePrecalc = e->haveAdjoints = false;
    return SDVGN_OK;
}

// FrameHessian::frameEnergyTH of every frame as the last linearizeAll left it (setNewFrameEnergyTH moves the newest frame's)
int sdvgn_ef_get_frame_energy_th(sdvgn_ef* e, float* th) {
    if (!e || !th || e->nF < 1) return SDVGN_E_ARG;
    if (e->host_only) { for (int i = 0; i < e->nF; ++i) th[i] = e->frames[i].frameEnergyTH; return SDVGN_OK; }
    EF_DEVICE(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(th, e->th_dev + (size_t)e->new_cur * SDVGN_MAX_FRAMES, sizeof(float) * e->nF, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nF; ++i) e->frames[i].frameEnergyTH = th[i];
    return SDVGN_OK;
}

int sdvgn_ef_set_frame_states(sdvgn_ef* e, const double* state10) {
    if (!e || !state10) return SDVGN_E_ARG;
    for (int i = 0; i < e->nF; ++i) frame_set_state(e->frames[i], state10 + 10 * i);
    e->havePrecalc = false;
    e->state_dirty = true; e->sys_valid = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_host_range(sdvgn_ef* e, int h0, int h1) {
    if (!e || h0 < 0 || h1 < h0 || h1 > SDVGN_MAX_FRAMES) return SDVGN_E_ARG;
    e->h0 = h0; e->h1 = h1;
    e->shard_set = !(h0 == 0 && h1 == SDVGN_MAX_FRAMES);   // [0, SDVGN_MAX_FRAMES) = "everything", whatever nF becomes
    e->havePrecalc = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_frame_image(sdvgn_ef* e, int idx, const float* dI) {
    if (!e || !dI || idx < 0 || idx >= SDVGN_MAX_FRAMES) return SDVGN_E_ARG;
    EF_DEVICE(e);
    const size_t n = (size_t)e->w * e->h * 3;
    HIPCHK(hipMemcpyAsync(e->images + n * e->img_slot[idx], dI, sizeof(float) * n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SDVGN_OK;
}

int sdvgn_ef_set_frame_image_raw(sdvgn_ef* e, int idx, const float* image) {
    if (!e || !image || idx < 0 || idx >= SDVGN_MAX_FRAMES) return SDVGN_E_ARG;
    EF_DEVICE(e);
    const size_t n = (size_t)e->w * e->h;
    HIPCHK(hipMemcpyAsync(e->img_stage, image, sizeof(float) * n, hipMemcpyHostToDevice, e->stream));
    const int qw = (e->w + 1) >> 1, qh = (e->h + 1) >> 1;
    k_pyr_level<<<dim3((qw + 255) / 256, qh), 256, 0, e->stream>>>(e->img_stage, e->images + 3 * n * e->img_slot[idx], nullptr, e->w, e->h, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return SDVGN_OK;
}

int sdvgn_ef_set_points(sdvgn_ef* e, int nP, const int* host, const float* u, const float* v, const float* idepth, const float* idepth_zero,
                        const float* color8, const float* weights8, const unsigned char* hasDepthPrior, const unsigned char* isFromSensor) {
    if (!e || nP < 0 || nP > e->max_points || e->nF < 1) return SDVGN_E_ARG;
    if (nP > 0 && (!host || !u || !v || !idepth || !idepth_zero || !color8 || !weights8 || !hasDepthPrior || !isFromSensor)) return SDVGN_E_ARG;
    // a reload of the point table ends the resident window (ids, slot-addressed residual tables): without this a set_points / set_residuals
    // sequence that skips set_frames left table_mode and the old id tables in place, and the next optimize_finish copied nF * nP bytes into
    // a caller buffer of nR.  An open edit session must be committed first.
    if (e->win && ef_win_is_open(e->win)) return SDVGN_E_STATE;
    e->table_mode = false;
    if (e->win) win_forget(e->win);
    EF_DEVICE(e);
    e->phost.assign(host, host + nP);
    e->hostP0.assign(e->nF + 1, 0);
    for (int i = 0; i < nP; ++i) {
        if (host[i] < 0 || host[i] >= e->nF || (i > 0 && host[i] < host[i - 1])) return SDVGN_E_ARG;  // grouped by host, ascending
        e->hostP0[host[i] + 1]++;
    }
    for (int h = 0; h < e->nF; ++h) {
        // the per-pair kernels cover kMaxChunks * 256 points of one host frame (k_ef_linearize: 2 * kMaxChunks chunks of 128)
        if (e->hostP0[h + 1] > kMaxChunks * 256) { e->hostP0.clear(); e->phost.clear(); e->nP = 0; return SDVGN_E_ARG; }
        e->hostP0[h + 1] += e->hostP0[h];
    }
    e->nP = nP;
    std::vector<float> prior(nP), delta(nP), ids(nP), idz(nP);
    bool any_delta = false;
    for (int i = 0; i < nP; ++i) {
        prior[i] = hasDepthPrior[i] ? kIdepthFixPrior * SDVGN_SCALE_IDEPTH * SDVGN_SCALE_IDEPTH : 0;   // EFPoint::takeData
        delta[i] = idepth[i] - idepth_zero[i];
        if (delta[i] != 0.0f) any_delta = true;
        ids[i] = SDVGN_SCALE_IDEPTH * idepth[i]; idz[i] = SDVGN_SCALE_IDEPTH * idepth_zero[i];
    }
    const hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(e->pu, u, 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pv, v, 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pid, ids.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pidz, idz.data(), 4 * nP, hipMemcpyHostToDevice, s));
    // the trial copies start identical (points outside this rank's shard are never rewritten and must read the same from both)
    HIPCHK(hipMemcpyAsync(e->pid_alt, ids.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pidz_alt, idz.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pdeltaF_alt, delta.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pcolor, color8, 32 * (size_t)nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pweights, weights8, 32 * (size_t)nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->ppriorF, prior.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->pdeltaF, delta.data(), 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->psensor, isFromSensor, nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->phost_dev, host, 4 * nP, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->hostP0_dev, e->hostP0.data(), 4 * (e->nF + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(e->rflags, 0, (size_t)e->nF * nP, s));
    HIPCHK(hipMemsetAsync(e->pstep, 0, 4 * (size_t)nP, s));
    HIPCHK(hipStreamSynchronize(s));
    ef_update_const(e);
    e->havePrecalc = false;
    e->nR = 0;
    e->deltaF_nonzero = any_delta;
    return SDVGN_OK;
}

int sdvgn_ef_set_residuals(sdvgn_ef* e, int nR, const int* point, const int* target, const int* state_state, const unsigned char* hasMatcher,
                           const double* matcher, const unsigned char* isLinearized, const unsigned char* isActive) {
    if (!e || nR < 0 || e->nP < 1) return SDVGN_E_ARG;
    if (nR > 0 && (!point || !target || !state_state || !hasMatcher || !matcher || !isLinearized || !isActive)) return SDVGN_E_ARG;
    if (e->win && ef_win_is_open(e->win)) return SDVGN_E_STATE;
    e->table_mode = false;                  // (a caller-side residual list again: see sdvgn_ef_set_points)
    if (e->win) win_forget(e->win);
    EF_DEVICE(e);
    const size_t slots = (size_t)e->nF * e->nP;
    std::vector<uint8_t> flags(slots, 0);
    std::vector<int8_t> st(slots, 0);
    std::vector<float2> m(slots, make_float2(0, 0));
    e->r_slot.resize(nR);
    bool any_lin = false;
    for (int i = 0; i < nR; ++i) {
        if (point[i] < 0 || point[i] >= e->nP || target[i] < 0 || target[i] >= e->nF || target[i] == e->phost[point[i]]) return SDVGN_E_ARG;
        const size_t s = (size_t)target[i] * e->nP + point[i];
        if (flags[s] & RF_EXISTS) return SDVGN_E_ARG;
        flags[s] = RF_EXISTS | (hasMatcher[i] ? RF_MATCHER : 0) | (isLinearized[i] ? RF_LINEARIZED : 0) | (isActive[i] ? RF_ACTIVE : 0);
        st[s] = (int8_t)state_state[i];
        m[s] = make_float2((float)matcher[2 * i], (float)matcher[2 * i + 1]);   // `matcher.cast<float>()`, Residuals.cpp:196
        if (isLinearized[i]) any_lin = true;
        e->r_slot[i] = (int)s;
    }
    const hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(e->rflags, flags.data(), slots, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->rstate, st.data(), slots, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->rflags_alt, flags.data(), slots, hipMemcpyHostToDevice, s));      // (both copies of the planes applyRes writes)
    HIPCHK(hipMemcpyAsync(e->rstate_alt, st.data(), slots, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(e->renergy_alt, 0, 4 * slots, s));
    HIPCHK(hipMemsetAsync(e->JpJd_alt, 0, sizeof(float) * 6 * slots, s));
    e->applied_synced = true;
    HIPCHK(hipMemcpyAsync(e->rmatcher, m.data(), sizeof(float2) * slots, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(e->rstate_new, RS_OUTLIER, slots, s));
    HIPCHK(hipMemsetAsync(e->renergy, 0, 4 * slots, s));
    HIPCHK(hipMemsetAsync(e->renergy_new, 0, 4 * slots, s));
    HIPCHK(hipMemsetAsync(e->renergy_wo, 0, 4 * slots, s));
    HIPCHK(hipMemsetAsync(e->rstate_new2, RS_OUTLIER, slots, s));
    HIPCHK(hipMemsetAsync(e->renergy_new2, 0, 4 * slots, s));
    HIPCHK(hipMemsetAsync(e->renergy_wo2, 0, 4 * slots, s));
    e->new_cur = 0;
    ef_select_new_set(e, 0, 0);
    HIPCHK(hipMemsetAsync(e->rres_toZero, 0, 8 * slots, s));
    HIPCHK(hipMemsetAsync(e->J, 0, sizeof(float) * 2 * kJPlanes * slots, s));
    HIPCHK(hipMemsetAsync(e->JpJd, 0, sizeof(float) * 6 * slots, s));
    HIPCHK(hipStreamSynchronize(s));
    e->nR = nR;
    e->has_linearized = any_lin;
    return SDVGN_OK;
}

// EFResidual::takeDataF for Jacobians linearised on the host (EnergyFunctionalStructs.cpp:15-25): the rows go to the buffer the
// EnergyFunctional side owns (RF_SEL as sdvgn_ef_set_residuals left it), JpJdF = Jpdxi[0] * Jpdd[0] + Jpdxi[1] * Jpdd[1] in float like the reference
// (rows scattered on the device from one staged upload: the first version read all 48 Jacobian planes back, patched them on the host and wrote them
// again -- 25 MB each way per call at the named size, for every solveSystemF of the drop-in's form A)
__global__ void __launch_bounds__(256) k_ef_scatter_jacobians(int nR, size_t slots, const int* __restrict__ r_slot, const float* __restrict__ J24, const float* __restrict__ r2z,
                                                              const uint8_t* __restrict__ rflags, float* __restrict__ J, float* __restrict__ JpJd, float* __restrict__ rres_toZero) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;     // 32 lanes per residual: 24 Jacobian planes | 6 JpJd | 2 res_toZero
    if (i >= nR) return;
    const size_t s = (size_t)r_slot[i];
    const float* j = J24 + (size_t)i * 24;
    if (k < kJPlanes) {
        const int buf = (rflags[s] & RF_SEL) ? 1 : 0;
        J[(size_t)buf * kJPlanes * slots + (size_t)k * slots + s] = j[k];
    } else if (k < kJPlanes + 6) {
        const int q = k - kJPlanes;
        JpJd[(size_t)q * slots + s] = j[2 + q] * j[22] + j[8 + q] * j[23];        // takeDataF: Jpdxi[0] * Jpdd[0] + Jpdxi[1] * Jpdd[1], in float
    } else if (r2z) {
        const int q = k - kJPlanes - 6;
        rres_toZero[(size_t)q * slots + s] = r2z[2 * i + q];
    }
}
int sdvgn_ef_set_residual_jacobians(sdvgn_ef* e, int nR, const float* J24, const float* res_toZero2) {
    if (!e || !J24 || nR < 0 || nR != e->nR || e->nP < 1) return SDVGN_E_ARG;
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    const size_t slots = (size_t)e->nF * e->nP;
    const size_t need = (size_t)nR * (4 + 96 + 8);
    if (need > e->jstage_bytes) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->jstage_dev) SDVGN_DFREE(e->jstage_dev);
        e->jstage_dev = nullptr; e->jstage_bytes = 0;
        const size_t cap = std::max(need, (size_t)e->slots_cap * (4 + 96 + 8) / 2);
        HIPCHK(SDVGN_DMALLOC(&e->jstage_dev, cap));
        e->jstage_bytes = cap;
    }
    int* slot_d = (int*)e->jstage_dev;
    float* j_d = (float*)((char*)e->jstage_dev + 4 * (size_t)nR);
    float* z_d = j_d + 24 * (size_t)nR;
    const hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(slot_d, e->r_slot.data(), 4 * (size_t)nR, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(j_d, J24, 96 * (size_t)nR, hipMemcpyHostToDevice, s));
    if (res_toZero2) HIPCHK(hipMemcpyAsync(z_d, res_toZero2, 8 * (size_t)nR, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(e->JpJd, 0, sizeof(float) * 6 * slots, s));               // (slots without a listed residual read 0, as before)
    HIPCHK(hipMemsetAsync(e->rres_toZero, 0, sizeof(float) * 2 * slots, s));
    if (nR > 0) k_ef_scatter_jacobians<<<(nR + 7) / 8, 256, 0, s>>>(nR, slots, slot_d, j_d, res_toZero2 ? z_d : nullptr, e->rflags, e->J, e->JpJd, e->rres_toZero);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));      // the caller's arrays may go away
    e->sys_valid = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_marg_prior(sdvgn_ef* e, const double* HM, const double* bM) {
    if (!e || !HM || !bM || e->nF < 1) return SDVGN_E_ARG;
    const int n = CPARS + 6 * e->nF;
    e->HM.assign(HM, HM + (size_t)n * n); e->bM.assign(bM, bM + n);
    e->win_dirty = true; e->sys_valid = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_nullspaces(sdvgn_ef* e, int k, const double* v) {
    if (!e || k < 0 || (k > 0 && !v) || e->nF < 1) return SDVGN_E_ARG;
    const int n = CPARS + 6 * e->nF;
    e->nullspaces.clear();
    e->ns_dirty = true; e->win_dirty = true;
    for (int j = 0; j < k; ++j) e->nullspaces.emplace_back(v + (size_t)j * n, v + (size_t)(j + 1) * n);
    return SDVGN_OK;
}

// FullSystem::getNullspaces (FullSystemOptimize.cpp:548-588) from the per-frame null-space columns FrameHessian::setStateZero computes
// by central differences around the linearisation point (HessianBlocks.cpp:57-76): 6 pose directions + 1 scale direction, the vectors
// solveSystemF's orthogonalize(&x, 0) projects out (lastNullspaces_pose + lastNullspaces_scale, EnergyFunctional.cpp:727-735).
int sdvgn_ef_compute_nullspaces(sdvgn_ef* e) {
    if (!e || e->nF < 1) return SDVGN_E_STATE;
    const int nF = e->nF, n = CPARS + 6 * nF;
    e->nullspaces.assign(7, std::vector<double>(n, 0.0));
    const double sInv[6] = {1.0f / kScaleXiTrans, 1.0f / kScaleXiTrans, 1.0f / kScaleXiTrans, 1.0f / kScaleXiRot, 1.0f / kScaleXiRot, 1.0f / kScaleXiRot};
    for (int h = 0; h < nF; ++h) {
        const gn::Pose T = e->frames[h].evalPT, Ti = gn::inverse(T);
        for (int i = 0; i < 6; ++i) {
            double ep[6] = {0, 0, 0, 0, 0, 0}, em[6] = {0, 0, 0, 0, 0, 0}, lp[6], lm[6];
            ep[i] = 1e-3; em[i] = -1e-3;
            gn::log_se3(gn::compose(gn::compose(T, gn::exp_se3(ep)), Ti), lp);
            gn::log_se3(gn::compose(gn::compose(T, gn::exp_se3(em)), Ti), lm);
            for (int r = 0; r < 6; ++r) e->nullspaces[i][CPARS + 6 * h + r] = (lp[r] - lm[r]) / (2e-3) * sInv[r];
        }
        gn::Pose P = T, M = T;
        for (int r = 0; r < 3; ++r) { P.t[r] *= 1.00001; M.t[r] /= 1.00001; }
        double lp[6], lm[6];
        gn::log_se3(gn::compose(P, Ti), lp);
        gn::log_se3(gn::compose(M, Ti), lm);
        for (int r = 0; r < 6; ++r) e->nullspaces[6][CPARS + 6 * h + r] = (lp[r] - lm[r]) / (2e-3) * sInv[r];
    }
    e->ns_dirty = true; e->win_dirty = true;
    return SDVGN_OK;
}

int sdvgn_ef_get_nullspaces(sdvgn_ef* e, double* out, int cap_vectors) {
    if (!e) return SDVGN_E_ARG;
    const int k = (int)e->nullspaces.size(), n = CPARS + 6 * e->nF;
    if (out) for (int j = 0; j < k && j < cap_vectors; ++j) std::memcpy(out + (size_t)j * n, e->nullspaces[j].data(), sizeof(double) * n);
    return k;
}

int sdvgn_ef_set_adjoints(sdvgn_ef* e) {  // EnergyFunctional::setAdjointsF
    if (!e || e->nF < 1) return SDVGN_E_STATE;
    const int nF = e->nF;
    e->adHost.assign((size_t)nF * nF * 36, 0); e->adTarget.assign((size_t)nF * nF * 36, 0);
    e->adHostF.assign((size_t)nF * nF * 36, 0); e->adTargetF.assign((size_t)nF * nF * 36, 0);
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            const gn::Pose hostToTarget = gn::compose(e->frames[t].evalPT, gn::inverse(e->frames[h].evalPT));
            double Adj[36];
            gn::adjoint(hostToTarget, Adj);
            double* AH = &e->adHost[(size_t)(h + t * nF) * 36];
            double* AT = &e->adTarget[(size_t)(h + t * nF) * 36];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) { AH[r * 6 + c] = -Adj[c * 6 + r]; AT[r * 6 + c] = (r == c) ? 1.0 : 0.0; }
            for (int r = 0; r < 6; ++r) {
                const float sc = r < 3 ? kScaleXiTrans : kScaleXiRot;
                for (int c = 0; c < 6; ++c) { AH[r * 6 + c] *= sc; AT[r * 6 + c] *= sc; }
            }
            for (int i = 0; i < 36; ++i) { e->adHostF[(size_t)(h + t * nF) * 36 + i] = (float)AH[i]; e->adTargetF[(size_t)(h + t * nF) * 36 + i] = (float)AT[i]; }
        }
    for (int i = 0; i < 4; ++i) e->cPrior[i] = kInitialCalibHessian;
    e->haveAdjoints = true;
    e->havePrecalc = false;
    e->win_dirty = true; e->sys_valid = false;
    return SDVGN_OK;
}

int sdvgn_ef_set_precalc(sdvgn_ef* e) {
    if (!e || e->nF < 1 || !e->haveAdjoints) return SDVGN_E_STATE;
    if (!e->host_only) HIPCHK(hipSetDevice(e->device));
    return ef_upload_precalc(e);
}

int sdvgn_ef_make_resident(sdvgn_ef* e) {
    if (!e || e->host_only || e->nF < 1 || !e->haveAdjoints) return SDVGN_E_STATE;
    EF_DEVICE(e);
    int rc;
    if (!e->havePrecalc && (rc = ef_upload_precalc(e))) return rc;
    if ((rc = ef_sync_window(e)) || (rc = ef_sync_state(e))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return SDVGN_OK;
}

int sdvgn_ef_linearize_all(sdvgn_ef* e, double* energy_out) {
    if (!e || e->host_only || !e->havePrecalc || e->nR < 0 || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    const int n_partials = ef_launch_linearize(e);
    double* edst = e->stats_dev;
    if (energy_out) k_ef_sum_energy<<<1, 256, 0, e->stream>>>(e->energy_partial, n_partials, edst);   // NULL: leave the per-workgroup partials
    if (ef_sharded(e)) {   // setNewFrameEnergyTH needs the candidates of every rank: one all-reduce of [4 unused | nP candidates]
        ef_launch_pack_th(e);
        { const int rca = ef_allreduce(e, e->stats_dev, 4 + e->nP); if (rca) return rca; }
        ef_launch_select_th(e, true);
    } else ef_launch_select_th(e, false);
    HIPCHK(hipGetLastError());
    if (energy_out) {
        HIPCHK(hipMemcpyAsync(e->acc_host, edst, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        *energy_out = e->acc_host[0];
    }
    return SDVGN_OK;
}

int sdvgn_ef_apply_res(sdvgn_ef* e) {
    if (!e || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    const size_t slots = (size_t)e->nF * e->nP;
    // (a threshold select the optimize loop has pending stays pending: applyRes does not read the thresholds)
    k_ef_apply<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->nF, e->nP, e->A, e->precalc_dev, e->phost_dev, nullptr);
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

// The accumulate kernels of one solveSystemF: [top Gram | per-point sums], [Schur Gram]; their per-workgroup partial tiles are
// reduced either by k_ef_acc_reduce (with_reduce: packed buffer in acc_dev, the form the all-reduce of a sharded window needs) or by
// left as partial tiles.
struct AccGeom { int pairs, chunks, sc_chunks, sc_ppb, ntop, nsc; };
static AccGeom ef_acc_geom(const sdvgn_ef* e) {
    AccGeom g;
    const int nF = e->nF;
    g.pairs = nF * nF; g.chunks = chunks_for_np(e);
    int mx = 1;
    for (int h = 0; h < nF; ++h) mx = std::max(mx, e->hostP0[h + 1] - e->hostP0[h]);
    // one 64-point tile per Schur-Gram workgroup (a workgroup is a chain of dependent round trips: flags + JpJdF -> LDS -> MFMA;
    // 64 / 128 / 192 points per workgroup measured 99.9 / 101.0 / 105.3 us per loop body in round 1); the same chunking on every path,
    // so that a sharded run reproduces the single-GPU run bit for bit
    g.sc_chunks = std::min(kMaxChunks, (mx + 63) / 64);
    g.sc_ppb = ((mx + g.sc_chunks - 1) / g.sc_chunks + 63) / 64 * 64;
    g.ntop = g.pairs * kTopE; g.nsc = nF * kScE;
    return g;
}
static int ef_flush_stats(sdvgn_ef* e) {      // the pending initial statistics as a launch of their own
    if (!e->pend_stats_valid) return SDVGN_OK;
    e->pend_stats_valid = false;
    const StatsLaunch& st = e->pend_stats;
    const SelArgs nosel{};
    k_ef_stats_select<<<1, kSelLanes, 0, e->stream>>>(st.pe, st.nE, st.pl, st.nL, st.ps, st.nS, st.out, st.done_flag, st.done_seq, nosel, st.dec);
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}
// st: the statistics + accept test of the trial step as workgroup 0 of the accumulate's launch (k_ef_acc_stats; alt->verdict_word set by the caller)
static int ef_accumulate(sdvgn_ef* e, bool with_reduce, const AccAlt* alt = nullptr, const StatsLaunch* st = nullptr) {
    const AccGeom g = ef_acc_geom(e);
    const int nF = e->nF, n_top = g.chunks * g.pairs, n_pt = (e->nP + 63) / 64;
    if (st && !(alt && alt->verdict_word && alt->skip_on_reject && g.sc_ppb == 64)) return SDVGN_E_STATE;
    if (e->pend_stats_valid) {       // the initial linearizeAll's statistics: with this launch when it can carry them, alone and first otherwise
        if (!st && !alt && g.sc_ppb == 64) { st = &e->pend_stats; e->pend_stats_valid = false; }
        else { const int rcf = ef_flush_stats(e); if (rcf) return rcf; }
    }
    if (g.sc_ppb == 64) {
        const int n_sc = nF * g.sc_chunks;
        const AccAlt none{};
        if (st) k_ef_acc_stats<<<8 + n_sc + n_top, 256, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->phost_dev, e->top_partial, e->nres_partial, g.chunks, e->sc_partial,
                                                                        g.sc_chunks, n_sc, alt ? *alt : none, *st);
        else
        k_ef_acc_fused<<<n_sc + n_top, 256, 0, e->stream>>>(e->precalc_dev, e->C, e->A, e->phost_dev, e->top_partial, e->nres_partial, g.chunks, e->sc_partial, g.sc_chunks, n_sc,
                                                            alt ? *alt : none);
    } else {
        if (alt) return SDVGN_E_STATE;   // (the speculative launch exists for the fused kernel only; callers check ef_acc_geom first)
        k_ef_acc_stage1<<<n_top + n_pt, 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->phost_dev, e->top_partial, e->nres_partial, g.chunks, n_top);
        k_ef_sc_gram<<<dim3(g.sc_chunks, nF), 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->sc_partial, g.sc_ppb);
    }
    if (with_reduce)
        k_ef_acc_reduce<<<acc_reduce_grid(g.pairs, nF), 256, 0, e->stream>>>(e->top_partial, g.pairs, g.chunks, e->sc_partial, nF, g.sc_chunks, e->nres_partial, e->acc_dev,
                                                                              (alt && alt->skip_on_reject) ? alt->verdict : nullptr);
    e->acc_in_host = false;
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

int sdvgn_ef_accumulate(sdvgn_ef* e) {
    if (!e || e->host_only || !e->havePrecalc || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    return ef_accumulate(e, /*with_reduce=*/true);
}

int sdvgn_ef_set_external_buffers(sdvgn_ef* e, double* acc_dev, int acc_capacity, double* stats_dev, int stats_capacity) {
    if (!e || e->host_only || !acc_dev || !stats_dev) return SDVGN_E_ARG;
    const size_t accmax = (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kTopE + (size_t)SDVGN_MAX_FRAMES * kScE + 1;
    if ((size_t)acc_capacity < accmax || (size_t)stats_capacity < 4 + (size_t)e->max_points) return SDVGN_E_ARG;
    EF_DEVICE(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_acc) SDVGN_DFREE(e->acc_dev);
    if (e->own_stats) SDVGN_DFREE(e->stats_dev);
    e->acc_dev = acc_dev; e->stats_dev = stats_dev;
    e->own_acc = e->own_stats = false;
    return SDVGN_OK;
}

// One collective per loop body: `buf_dev` (device, >= 2 x sdvgn_ef_collective_stride doubles; NULL: the library allocates) becomes two
// message buffers [packed accumulators | 4 statistics | max_points quantile candidates]; sdvgn_ef_optimize then all-reduces ONE message per
// loop body (+ one per call) instead of the accumulators and the statistics separately.
static size_t coll_stride_for(const sdvgn_ef* e) {
    return (size_t)SDVGN_MAX_FRAMES * SDVGN_MAX_FRAMES * kTopE + (size_t)SDVGN_MAX_FRAMES * kScE + 1 + 4 + (size_t)e->max_points;
}
int sdvgn_ef_collective_stride(sdvgn_ef* e) { return e && !e->host_only ? (int)coll_stride_for(e) : SDVGN_E_ARG; }
int sdvgn_ef_set_collective_buffer(sdvgn_ef* e, double* buf_dev, int capacity) {
    if (!e || e->host_only) return SDVGN_E_ARG;
    EF_DEVICE(e);
    const size_t stride = coll_stride_for(e);
    if (buf_dev && (size_t)capacity < 2 * stride) return SDVGN_E_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_coll && e->coll[0]) SDVGN_DFREE(e->coll[0]);
    e->own_coll = false;
    if (!buf_dev) { HIPCHK(SDVGN_DMALLOC((void**)&buf_dev, sizeof(double) * 2 * stride)); e->own_coll = true; }
    HIPCHK(hipMemsetAsync(buf_dev, 0, sizeof(double) * 2 * stride, e->stream));
    e->coll[0] = buf_dev; e->coll[1] = buf_dev + stride; e->coll_stride = stride; e->coll_cur = 0;
    if (!e->apply_bak.fl) {
        const size_t slots = e->slots_cap;
        HIPCHK(SDVGN_DMALLOC((void**)&e->apply_bak.fl, slots)); HIPCHK(SDVGN_DMALLOC((void**)&e->apply_bak.st, slots));
        HIPCHK(SDVGN_DMALLOC((void**)&e->apply_bak.en, sizeof(float) * slots)); HIPCHK(SDVGN_DMALLOC((void**)&e->apply_bak.JpJd, sizeof(float) * 6 * slots));
    }
    return SDVGN_OK;
}
unsigned long long sdvgn_ef_collective_count(sdvgn_ef* e) { return e ? e->n_collectives : 0; }

int sdvgn_ef_set_allreduce(sdvgn_ef* e, void (*fn)(void*, double*, int), void* user) {
    if (!e) return SDVGN_E_ARG;
    e->allreduce = fn; e->allreduce_user = user;
    return SDVGN_OK;
}

int sdvgn_rccl_unique_id(unsigned char* out128) {
    if (!out128) return SDVGN_E_ARG;
    RcclApi& api = rccl_api();
    if (!api.ok) return SDVGN_E_STATE;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return SDVGN_E_STATE;
    std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SDVGN_OK;
}

int sdvgn_ef_init_rccl(sdvgn_ef* e, const unsigned char* id128, int rank, int world) {
    if (e && !e->host_only && !id128) {   // id128 == NULL: drop the communicator, back to the callback / single-GPU behaviour
        EF_DEVICE(e);
        if (e->rccl_comm) { hipStreamSynchronize(e->stream); ef_release_comm(e); }
        return SDVGN_OK;
    }
    if (!e || e->host_only || world < 1 || rank < 0 || rank >= world) return SDVGN_E_ARG;
    RcclApi& api = rccl_api();
    if (!api.ok) return SDVGN_E_STATE;
    EF_DEVICE(e);
    if (e->rccl_comm) { hipStreamSynchronize(e->stream); ef_release_comm(e); }
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (SharedComm& c : g_comms)
        if (!std::memcmp(c.id, id128, NCCL_UNIQUE_ID_BYTES)) {
            // the communicator of this id exists in the process: share it -- but only as the SAME rank of the same clique on the same device
            // (a second rank of one id inside one process would need a communicator of its own; not supported: say so instead of returning
            // the first handle's and hanging or mis-summing later)
            if (c.rank != rank || c.world != world || c.device != e->device) return SDVGN_E_ARG;
            ++c.refs; e->rccl_comm = c.comm; return SDVGN_OK;   // not collective
        }
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    if (api.CommInitRank(&comm, world, id, rank) != ncclSuccess) return SDVGN_E_STATE;   // collective: every rank calls it
    SharedComm sc;
    std::memcpy(sc.id, id128, NCCL_UNIQUE_ID_BYTES); sc.comm = comm; sc.refs = 1; sc.rank = rank; sc.world = world; sc.device = e->device;
    g_comms.push_back(sc);
    e->rccl_comm = comm;
    return SDVGN_OK;
}

// ranks of the RCCL communicator this handle issues its collectives on (ncclCommCount), 0 if it has none (single GPU, or the callback path)
int sdvgn_ef_rccl_ranks(sdvgn_ef* e) {
    if (!e) return SDVGN_E_ARG;
    if (!e->rccl_comm || !rccl_api().CommCount) return 0;
    int n = 0;
    return rccl_api().CommCount(e->rccl_comm, &n) == ncclSuccess ? n : SDVGN_E_STATE;
}

int sdvgn_rccl_comm_alive(const unsigned char* id128) {
    if (!id128) return 0;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (const SharedComm& c : g_comms)
        if (!std::memcmp(c.id, id128, NCCL_UNIQUE_ID_BYTES)) return 1;
    return 0;
}

int sdvgn_ef_accumulator_count(sdvgn_ef* e) { return e ? (int)acc_count(e) : SDVGN_E_ARG; }

int sdvgn_ef_accumulators_dev(sdvgn_ef* e, double** buf, int* count) {
    if (!e || !buf || !count) return SDVGN_E_ARG;
    *buf = e->acc_dev; *count = (int)acc_count(e);
    return SDVGN_OK;
}

// host part of solveSystemF on an accumulator buffer (own or all-reduced): stitch, HM/bM, damped preconditioned LDLT,
// null-space projection.  Pure host code -- also the entry point of the CPU (gloo) test of the multi-GPU logic.
// Host LDL^T with Eigen's pivot order (largest remaining ORIGINAL diagonal first, see tracker_track_kernel.inc) in right-looking
// form: every inner loop is an element-wise row update (no floating-point reduction), which the compiler may vectorise without
// -ffast-math; the left-looking gn::ldlt_solve_inplace spends its time in scalar dot products at n = 52.  Only the upper triangle
// A[r][c], c >= r of the trailing block is kept current (half the updates of a full-square right-looking step); the multipliers
// of step k stay in row k (A[k][c] = L[c][k], c > k), i.e. L^T sits in the strict upper triangle.  A (n x n, row-major, upper
// triangle read) and b are overwritten; b returns x.  Same solution as the left-looking form up to rounding order.
SDVGN_HOST_FMA static void ldlt_solve_rl(int n, double* A, double* b) {
#pragma clang fp contract(fast)
    constexpr int MAXN = CPARS + 6 * SDVGN_MAX_FRAMES;
    int perm[MAXN];
    double d0[MAXN];
    for (int i = 0; i < n; ++i) d0[i] = std::fabs(A[(size_t)i * n + i]);
    for (int k = 0; k < n; ++k) {
        int p = k; double big = d0[k];
        for (int i = k + 1; i < n; ++i) if (d0[i] > big) { big = d0[i]; p = i; }
        perm[k] = p;
        if (p != k) {   // symmetric swap k <-> p on upper storage
            std::swap(d0[k], d0[p]);
            for (int i = 0; i < k; ++i) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + p]);          // finished multipliers: columns k,p of rows < k
            std::swap(A[(size_t)k * n + k], A[(size_t)p * n + p]);
            for (int i = k + 1; i < p; ++i) std::swap(A[(size_t)k * n + i], A[(size_t)i * n + p]);      // (k,i) <-> (i,p)
            for (int i = p + 1; i < n; ++i) std::swap(A[(size_t)k * n + i], A[(size_t)p * n + i]);      // (k,i) <-> (p,i)
            std::swap(b[k], b[p]);
        }
        const double d = A[(size_t)k * n + k];
        if (std::fabs(d) > 0.0) {
            double* rowk = A + (size_t)k * n;
            for (int r = k + 1; r < n; ++r) {
                const double l = rowk[r] / d;                 // L[r][k]
                double* rr = A + (size_t)r * n;
                for (int c = r; c < n; ++c) rr[c] -= l * rowk[c];
                // rowk[r] must stay the un-scaled value until every row has used it: scale afterwards
            }
            for (int r = k + 1; r < n; ++r) rowk[r] = rowk[r] / d;
        }
    }
    // L^T in the strict upper triangle: forward substitution L y = b  ->  y[r] -= L[r][k] y[k] = U[k][r] y[k]
    for (int k = 0; k < n; ++k) { const double yk = b[k]; const double* rowk = A + (size_t)k * n; for (int r = k + 1; r < n; ++r) b[r] -= rowk[r] * yk; }
    for (int i = 0; i < n; ++i) { const double d = A[(size_t)i * n + i]; b[i] = (std::fabs(d) > 5.562684646268003e-309) ? b[i] / d : 0.0; }
    for (int k = n - 1; k >= 0; --k) { double s = b[k]; const double* rowk = A + (size_t)k * n; for (int r = k + 1; r < n; ++r) s -= rowk[r] * b[r]; b[k] = s; }
    for (int k = n - 1; k >= 0; --k) if (perm[k] != k) std::swap(b[k], b[perm[k]]);
}

static void ef_swap_point_copies(sdvgn_ef* e) {
    std::swap(e->pid, e->pid_alt); std::swap(e->pidz, e->pidz_alt); std::swap(e->pdeltaF, e->pdeltaF_alt);
    e->A.pid = e->pid; e->A.pidz = e->pidz; e->A.pdeltaF = e->pdeltaF;
}

// host version of the small solve (stitch, HM/bM, damped preconditioned LDLT, null-space projection) on a caller-supplied packed
// accumulator buffer: the entry point of host-only handles (CPU / gloo tests of the multi-GPU logic) and the reference the device
// solve (backend_solve.inc) is tested against
static int ef_stitch_solve_host(sdvgn_ef* e, const double* acc, int iteration, double lambda, double* x_out) {
    const int nF = e->nF, n = CPARS + 6 * nF, pairs = nF * nF;
    stitch_top(e, acc);
    std::vector<double> d(n), bM_top(n);
    for (int i = 0; i < CPARS; ++i) d[i] = (double)e->C.cDeltaF[i];
    for (int h = 0; h < nF; ++h) for (int i = 0; i < 6; ++i) d[CPARS + 6 * h + i] = e->frames[h].delta[i];
    for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < n; ++j) a += e->HM[(size_t)i * n + j] * d[j]; bM_top[i] = e->bM[i] + a; }
    e->resInA = (int)acc[acc_count(e) - 1];
    stitch_sc(e, acc + (size_t)pairs * kTopE);
    // HFinal = HA + HM - Hsc ; bFinal = bA + bM_top - bsc   (EnergyFunctional.cpp:668-699)
    e->HFinal.resize((size_t)n * n); e->bFinal.resize(n);
    for (size_t i = 0; i < (size_t)n * n; ++i) e->HFinal[i] = e->HA[i] + e->HM[i] - e->Hsc[i];
    for (int i = 0; i < n; ++i) e->bFinal[i] = e->bA[i] + bM_top[i] - e->bsc[i];
    std::vector<double> Hs((size_t)n * n), xs(n), sv(n);
    for (int i = 0; i < n; ++i) sv[i] = 1.0 / std::sqrt(e->HFinal[(size_t)i * n + i] * (1 + lambda) + 10);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            const double hij = (i == j) ? e->HFinal[(size_t)i * n + j] * (1 + lambda) : e->HFinal[(size_t)i * n + j];
            Hs[(size_t)i * n + j] = sv[i] * hij * sv[j];
        }
        xs[i] = sv[i] * e->bFinal[i];
    }
    ldlt_solve_rl(n, Hs.data(), xs.data());
    e->lastX.resize(n);
    for (int i = 0; i < n; ++i) e->lastX[i] = sv[i] * xs[i];
    if (iteration >= 2) orthogonalize_x(e, e->lastX);   // SOLVER_ORTHOGONALIZE_X_LATER
    if (x_out) std::memcpy(x_out, e->lastX.data(), sizeof(double) * n);
    e->sys_on_device = false;
    return SDVGN_OK;
}

int sdvgn_ef_stitch_solve_host(sdvgn_ef* e, const double* acc, int iteration, double lambda, double* x_out) {
    if (!e || !acc || !e->haveAdjoints || !e->havePrecalc) return SDVGN_E_STATE;
    ef_refresh_frame_deltas(e);
    return ef_stitch_solve_host(e, acc, iteration, lambda, x_out);
}

static void calib_set_value(sdvgn_ef* e, const double* v);

// ---- the device-resident solve ------------------------------------------------------------------------------------------------------
static void ef_fill_solve_io(sdvgn_ef* e, SolveIO& io, int iteration, double lambda, bool do_step, float stepsize, bool reuse) {
    io.acc = e->acc_dev;
    io.pieces = e->pieces_dev;
    io.W = e->win_dev;
    io.cur = e->sstate_dev + e->st_cur; io.trial = e->sstate_dev + (1 - e->st_cur);
    io.calib_cur = e->calib_dev + e->st_cur; io.calib_trial = e->calib_dev + (1 - e->st_cur);
    io.pc_cur = e->precalc_dev; io.pc_trial = e->precalc_alt;
    io.rx = e->rx_dev; io.sys = e->sys_dev; io.out = e->sol_host;
    io.done_flag = e->flags_host + 3; io.done_seq = ++e->seq_solve;
    io.xw = e->xw_dev;
    io.err_word = e->stats_host ? (unsigned*)(e->stats_host + 6) : nullptr;
    io.lambda = lambda; io.iteration = iteration; io.do_step = do_step ? 1 : 0; io.reuse = reuse ? 1 : 0; io.stepsize = stepsize;
    io.stamps = e->solve_stamps;
    io.wait_xw = nullptr; io.wait_seq = 0; io.spec = 0;
    io.tri_ready = (reuse || !e->xw_dev) ? nullptr : e->xw_dev + 498;      // (a solve that stitches anew says when its system is in memory)
    io.en_em_trial = (do_step && e->en_em_dev) ? e->en_em_dev + 2 * (1 - e->st_cur) : nullptr;
}
// solveSystemF on the device, all launches asynchronous: accumulate + reduce (unless the stitched system of the previous body is
// re-used), per-host stitch, then the one-workgroup tail: LDL^T + null-space projection (+ the calib / frame part of doStepFromBackup
// and the precalc table of the stepped state when do_step), then resubstituteF (+ the point part of doStepFromBackup when
// step_fac >= 0).  x reaches the host through pinned memory; ef_wait_solve fetches it.
static int ef_drain_spec(sdvgn_ef* e);
static int ef_launch_solve(sdvgn_ef* e, int iteration, double lambda, bool do_step, float step_fac, bool reuse, bool accumulated) {
    int rc;
    if ((rc = ef_sync_window(e)) || (rc = ef_sync_state(e))) return rc;
    if (!reuse && (rc = ef_drain_spec(e))) return rc;       // this solve rewrites the system a speculative solve may still be reading
    const int nF = e->nF;
    SolveIO io;
    ef_fill_solve_io(e, io, iteration, lambda, do_step, do_step ? step_fac : 0.0f, reuse);
    if (!reuse) {
        if (!accumulated) {
            if ((rc = ef_accumulate(e, /*with_reduce=*/true))) return rc;
            // sharded window: the packed buffer of every rank is summed (ONE all-reduce per GN iteration), then every rank stitches and solves
            if (ef_sharded(e) && (rc = ef_allreduce(e, e->acc_dev, (int)acc_count(e)))) return rc;
        }
        // the pending threshold select rides in the factorisation's launch (below); a handle that runs beside others keeps it in this one
        const int has_sel = (e->own_stream && e->pend_sel_valid) ? 1 : 0;
        k_ef_stitch<<<kStitchParts * nF + 1 + has_sel, kSolveLanes, 0, e->stream>>>(io, e->pend_sel, has_sel);
        if (has_sel) e->pend_sel_valid = false;
    }
    if (e->pend_sel_valid && e->own_stream) ef_flush_pending(e);    // (own stream + system re-used: no launch to ride in)
    const int has_rc = e->pend_rc_valid ? 1 : 0;
    const int nblk = (e->nP + 63) / 64;
    const int n_recl = has_rc ? kReclBlocks : 0;      // (beside the factorisation four workgroups are plenty: they have ~18 us)
    const int head = 1 + n_recl, rest = (nblk + 1) / 2 + (do_step ? (io.en_em_trial ? 2 : 1) : 0);   // + step (+ energy) workgroups
    const SelArgs sel = e->pend_sel;
    if (e->own_stream) {   // a window that runs beside others: no spinning workgroups (see k_ef_tail_resub)
        k_ef_tail_resub<<<head, kTailLanes, 0, e->stream>>>(io, e->pend_rc, n_recl, e->C, e->A, e->precalc_dev, e->phost_dev, e->pidepth_backup,
                                                             e->stats_partial + (e->nP / 64 + 2), step_fac, e->pid_alt, e->pidz_alt, e->pdeltaF_alt, nblk, 0, 0,
                                                             sel, -1, nullptr);
        k_ef_tail_resub<<<rest, kTailLanes, 0, e->stream>>>(io, e->pend_rc, n_recl, e->C, e->A, e->precalc_dev, e->phost_dev, e->pidepth_backup,
                                                             e->stats_partial + (e->nP / 64 + 2), step_fac, e->pid_alt, e->pidz_alt, e->pdeltaF_alt, nblk, head, 1,
                                                             sel, -1, nullptr);
    } else {
        // the pending select as the LAST workgroup of this launch; a re-classification of the same launch that reads the thresholds it writes
        // takes them as tagged words
        const int has_sel = e->pend_sel_valid ? 1 : 0;
        ReclArgs rcl = e->pend_rc;
        unsigned long long* thw = has_sel ? e->xw_dev + 500 : nullptr;
        if (has_sel && has_rc && rcl.th == sel.th_out) { rcl.thw = thw; rcl.thseq = (unsigned)io.done_seq; }
        k_ef_tail_resub<<<head + rest + has_sel, kTailLanes, 0, e->stream>>>(
            io, rcl, n_recl, e->C, e->A, e->precalc_dev, e->phost_dev, e->pidepth_backup, e->stats_partial + (e->nP / 64 + 2), step_fac, e->pid_alt, e->pidz_alt,
            e->pdeltaF_alt, nblk, 0, 0, sel, has_sel ? head + rest : -1, thw);
        e->pend_sel_valid = false;
    }
    e->pend_rc_valid = false;
    HIPCHK(hipGetLastError());
    e->sys_on_device = true; e->sys_fetched = false; e->sys_valid = true;
    return SDVGN_OK;
}
static inline bool ef_wait_error(const sdvgn_ef* e) {   // a workgroup gave up an intra-launch wait (EFArrays::err): the window is not trustworthy
    return e->stats_host && *reinterpret_cast<volatile const unsigned*>(e->stats_host + 6) != 0;
}
// an internal consistency check failed: the call returns SDVGN_E_STATE -- and says which one on stderr (these cannot happen in a correct build;
// a bare error code from a 300-line loop would be all a bug report had)
static int ef_state_failure(const sdvgn_ef* e, const char* what, double a = 0, double b = 0) {
    fprintf(stderr, "[sdvgn] internal check failed: %s (%.17g vs %.17g; error word %u)\n", what, a, b,
            e->stats_host ? *reinterpret_cast<volatile const unsigned*>(e->stats_host + 6) : 0u);
    // the hand-off words as they are now (the launches in flight are drained first)
    hipDeviceSynchronize();
    unsigned long long tri = 0, thw[2] = {0, 0}, xs[2] = {0, 0}; unsigned ver = 0;
    if (e->xw_dev) hipMemcpy(&tri, e->xw_dev + 498, 8, hipMemcpyDeviceToHost);
    if (e->accept_dev) hipMemcpy(&ver, e->accept_dev + 4, 4, hipMemcpyDeviceToHost);
    for (int q = 0; q < 2 && e->xw_spec; ++q) { hipMemcpy(&thw[q], e->xw_spec + q * 512 + 500, 8, hipMemcpyDeviceToHost); hipMemcpy(&xs[q], e->xw_spec + q * 512, 8, hipMemcpyDeviceToHost); }
    fprintf(stderr, "[sdvgn]   seq_solve %d seq_spec %d (tag %#x) seq_verdict %u | spec_last buf %d seq %#x | flags solve %d spec %d %d stats %d (seq_stats %d)\n"
                    "[sdvgn]   device words: tri_ready tag %u | verdict %u (seq %u) | spec x[0] tags %#x %#x | spec th[0] tags %#x %#x\n",
            e->seq_solve, e->seq_spec, 0x40000000u | (unsigned)e->seq_spec, e->seq_verdict, e->spec_last_buf, (unsigned)e->spec_last_seq,
            e->flags_host[3], e->flags_host[5], e->flags_host[6], e->flags_host[2], e->seq_stats,
            (unsigned)(tri >> 32), ver, ver >> 1, (unsigned)(xs[0] >> 32), (unsigned)(xs[1] >> 32), (unsigned)(thw[0] >> 32), (unsigned)(thw[1] >> 32));
    return SDVGN_E_STATE;
}
// ---- the rejected case, solved ahead ------------------------------------------------------------------------------------------------
// After a rejected step the reference restores the state and solves the SAME normal equations again with 100 x the damping
// (FullSystemOptimize.cpp:446-458: lambda *= 1e2, next iteration).  Nothing of that solve depends on the trial: the stitched system of the
// body that is running (SolveSys::tri), the next lambda and the next iteration number are all known the moment this body's solve has
// been queued.  So one workgroup on the side stream runs the factorisation for that case (k_ef_tail_resub's workgroup 0 in its re-use form)
// while the main stream linearises the trial; a rejection then finds x / xc / xAd ready and the next body is resubstitute + step + linearise.
// Bit for bit the solve the body would have run itself (the re-use form is tested against it: test_reuse_after_reject_is_bit_identical).
// No HIP events between the streams (an event record / wait pair per body cost more than the factorisation it hides: 61 -> 78 us per body,
// profiles/r04_notes.txt): the side launch polls the main solve's tagged x words before it reads the system, and publishes its own results
// as tagged words the next body's workgroups poll.
constexpr unsigned kSpecTag = 0x40000000u;
// The ONLY wait of the side-stream launch is for something the main stream was handed EARLIER (the solve whose system this one re-uses), and
// the host makes sure this launch is through before it queues anything that polls its words (ef_launch_spec_rest).  An earlier form of this
// round also selected the trial's thresholds on the side stream, waiting there for the body's accept test: 1 to 3 of 10 passes of the GPU
// suite ended with that wait given up -- the side stream is not always served while the main stream runs (profiles/r04_fault_hunt.txt).
static int ef_launch_spec_solve(sdvgn_ef* e, int iteration_next, double lambda_next, bool main_solve_in_flight) {
    const int buf = (e->seq_spec + 1) & 1;
    SolveIO io;
    std::memset(&io, 0, sizeof(io));
    io.W = e->win_dev; io.sys = e->sys_dev; io.pieces = e->pieces_dev; io.acc = e->acc_dev;
    io.rx = e->rx_spec; io.out = e->sol_spec + buf;
    io.done_flag = e->flags_host + 5 + buf; io.done_seq = (int)(kSpecTag | (unsigned)(++e->seq_spec));
    io.xw = e->xw_spec + (size_t)buf * 512;
    io.err_word = nullptr;                             // (its one wait gives up quietly: SolveOut::status 3, see solve_tail and ef_spec_valid)
    io.lambda = lambda_next; io.iteration = iteration_next; io.do_step = 0; io.reuse = 1; io.stepsize = 0.0f;
    io.stamps = nullptr;
    io.spec = 1;
    // the system is written by the solve the main stream has just been handed (unless that body itself started from a speculative solution:
    // then the system has not changed since an earlier body)
    io.wait_xw = main_solve_in_flight ? e->xw_dev + 498 : nullptr; io.wait_seq = (unsigned)e->seq_solve;
    const ReclArgs no_rc{};
    const SelArgs no_sel{};
    k_ef_tail_lookahead<<<1, kTailLanes, 0, e->side>>>(io, no_rc, 0, e->C, e->A, e->precalc_dev, e->phost_dev, e->pidepth_backup, e->stats_partial, -1.0f,
                                                   e->pid_alt, e->pidz_alt, e->pdeltaF_alt, 0, 0, 0, no_sel, -1, nullptr);
    HIPCHK(hipGetLastError());
    e->spec_last_buf = buf; e->spec_last_seq = io.done_seq;
    return SDVGN_OK;
}
// before the main stream gets a launch that REWRITES the system (a solve that stitches anew): the speculative solve launched last must be
// through -- it publishes a host flag when it is; normally it finished a whole linearise ago and this is one read of pinned memory
static int ef_drain_spec(sdvgn_ef* e) {
    if (e->spec_last_buf < 0) return SDVGN_OK;
    HIPCHK(wait_flag(e->flags_host + 5 + e->spec_last_buf, e->spec_last_seq, e->side));
    e->spec_last_buf = -1;
    return SDVGN_OK;
}
// the body after a rejected step when its solution was computed ahead: resubstituteF + doStepFromBackup (+ the pending threshold select and
// re-classification) -- k_ef_tail_resub without its workgroup 0; the workgroups poll the speculative solve's tagged words (normally long there)
static int ef_launch_spec_rest(sdvgn_ef* e, int iteration, double lambda, float step_fac) {
    int rc;
    if ((rc = ef_sync_window(e)) || (rc = ef_sync_state(e))) return rc;
    SolveIO io;
    ef_fill_solve_io(e, io, iteration, lambda, /*do_step=*/true, step_fac, /*reuse=*/true);
    io.rx = e->rx_spec;
    io.xw = e->xw_spec + (size_t)e->spec_last_buf * 512;
    io.done_seq = e->spec_last_seq;                     // the tag the polls wait for (and, being unique, the tag of this launch's threshold words)
    // (the speculative solve is through and valid: ef_spec_valid was asked before this body was entered -- nothing queued below waits for the side
    // stream on the device)
    const int has_rc = e->pend_rc_valid ? 1 : 0, has_sel = e->pend_sel_valid ? 1 : 0;
    const int nblk = (e->nP + 63) / 64;
    const int rest = (nblk + 1) / 2 + (io.en_em_trial ? 2 : 1);
    const SelArgs sel = e->pend_sel;
    ReclArgs rcl = e->pend_rc;
    unsigned long long* thw = has_sel ? e->xw_dev + 500 : nullptr;
    if (has_sel && has_rc && rcl.th == sel.th_out) { rcl.thw = thw; rcl.thseq = (unsigned)io.done_seq; }
    // the re-classification in ONE pass (one slot per lane): beside the factorisation four workgroups walking ~8 slots per lane were hidden,
    // here they would be the launch's duration (measured: 19 us for resubstitute + step with them, profiles/r04_notes.txt)
    const int n_recl = has_rc ? (rcl.nP + rcl.np_last * (rcl.nF - 1) + kTailLanes - 1) / kTailLanes : 0;
    const int lead = 1 + n_recl;                              // block indices [1, lead): the re-classification; block 0 (the factorisation) is not launched
    k_ef_resub_after_reject<<<lead - 1 + rest + has_sel, kTailLanes, 0, e->stream>>>(
        io, rcl, n_recl, e->C, e->A, e->precalc_dev, e->phost_dev, e->pidepth_backup, e->stats_partial + (e->nP / 64 + 2), step_fac, e->pid_alt, e->pidz_alt,
        e->pdeltaF_alt, nblk, /*first_block=*/1, /*no_wait=*/0, sel, has_sel ? lead + rest : -1, thw);
    e->pend_sel_valid = false; e->pend_rc_valid = false;
    HIPCHK(hipGetLastError());
    e->sys_on_device = true; e->sys_fetched = false; e->sys_valid = true;
    return SDVGN_OK;
}
// after a rejected step: is the solution computed ahead there?  Waits for the side-stream launch's completion flag (normally published a whole
// linearise ago: one read of pinned memory) and looks at its status -- 3 = it never saw the system it was to re-use (streams that are not served
// side by side, e.g. under a serialising profiler): the loop then solves the rejected case itself
static int ef_spec_valid(sdvgn_ef* e, bool* valid) {
    *valid = false;
    if (e->spec_last_buf < 0) return SDVGN_OK;
    HIPCHK(wait_flag(e->flags_host + 5 + e->spec_last_buf, e->spec_last_seq, e->side));
    *valid = e->sol_spec[e->spec_last_buf].status != 3;
    return SDVGN_OK;
}
static int ef_wait_spec_solve(sdvgn_ef* e, int buf, int seq, double* x_out) {
    HIPCHK(wait_flag(e->flags_host + 5 + buf, seq, e->side));
    if (ef_wait_error(e)) return ef_state_failure(e, "a workgroup gave up an intra-launch wait (solve ahead)");
    const int n = CPARS + 6 * e->nF;
    const SolveOut& o = e->sol_spec[buf];
    e->lastX.assign(o.x, o.x + n);
    if (x_out) std::memcpy(x_out, e->lastX.data(), sizeof(double) * n);
    e->solve_status = o.status;
    return SDVGN_OK;
}

static int ef_wait_solve(sdvgn_ef* e, double* x_out) {
    HIPCHK(wait_flag(e->flags_host + 3, e->seq_solve, e->stream));
    if (ef_wait_error(e)) return ef_state_failure(e, "a workgroup gave up an intra-launch wait (solve)");
    const int n = CPARS + 6 * e->nF;
    e->lastX.assign(e->sol_host->x, e->sol_host->x + n);
    e->resInA = e->sol_host->resInA;
    if (x_out) std::memcpy(x_out, e->lastX.data(), sizeof(double) * n);
    // a non-positive / non-finite pivot is not an error of the call: the device falls back to Eigen's pivoted LDL^T (status 2), which like
    // the reference's ldlt().solve() returns a finite x on an indefinite system (sdvgn_ef_get_solve_status)
    e->solve_status = e->sol_host->status;
    return SDVGN_OK;
}

int sdvgn_ef_finish_solve(sdvgn_ef* e, int iteration, double lambda, double* x_out) {
    if (!e || e->host_only || !e->havePrecalc || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    int rc = ef_launch_solve(e, iteration, lambda, /*do_step=*/false, -1.0f, /*reuse=*/false, /*accumulated=*/true);
    if (rc) return rc;
    return ef_wait_solve(e, x_out);
}

int sdvgn_ef_solve_system(sdvgn_ef* e, int iteration, double lambda, double* x_out) {
    if (!e || e->host_only || !e->havePrecalc || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    int rc = ef_launch_solve(e, iteration, lambda, /*do_step=*/false, -1.0f, /*reuse=*/false, /*accumulated=*/false);
    if (rc) return rc;
    return ef_wait_solve(e, x_out);
}

int sdvgn_ef_point_step(sdvgn_ef* e, int mode, float stepfacD) {
    if (!e || mode < 0 || mode > 2 || e->nP < 1) return SDVGN_E_ARG;
    EF_DEVICE(e);
    k_ef_point_step<<<(e->nP + 255) / 256, 256, 0, e->stream>>>(e->nP, mode, stepfacD, e->pid, e->pidz, e->pidepth_backup, e->pstep, e->pdeltaF);
    HIPCHK(hipGetLastError());
    if (mode != 0) e->deltaF_nonzero = false;
    return SDVGN_OK;
}


// ---- FullSystem::optimize loop body pieces (FullSystemOptimize.cpp:165-321, 344-458) ------------------------
static void calib_set_value(sdvgn_ef* e, const double* v) {   // CalibHessian::setValue
    for (int i = 0; i < 4; ++i) e->value[i] = v[i];
    e->value_scaled[0] = SDVGN_SCALE_F * v[0]; e->value_scaled[1] = SDVGN_SCALE_F * v[1];
    e->value_scaled[2] = SDVGN_SCALE_C * v[2]; e->value_scaled[3] = SDVGN_SCALE_C * v[3];
    for (int i = 0; i < 4; ++i) e->value_minus_value_zero[i] = e->value[i] - e->value_zero[i];
    ef_update_const(e);
}
static double calc_M_energy(sdvgn_ef* e) {   // EnergyFunctional::calcMEnergyF
    const int nF = e->nF, n = CPARS + 6 * nF;
    double d[CPARS + 6 * SDVGN_MAX_FRAMES];
    for (int i = 0; i < CPARS; ++i) d[i] = (double)e->C.cDeltaF[i];
    for (int h = 0; h < nF; ++h) for (int i = 0; i < 6; ++i) d[CPARS + 6 * h + i] = e->frames[h].delta[i];
    double s = 0;
    for (int i = 0; i < n; ++i) { double a = 2 * e->bM[i]; for (int j = 0; j < n; ++j) a += e->HM[(size_t)i * n + j] * d[j]; s += d[i] * a; }
    return s;
}
// linearizeAll + the point statistics + setNewFrameEnergyTH, in two halves: the launches (asynchronous) and the wait for the four
// sums {energy, L-energy point part, sum step^2, sum |idepth_backup|}
// defer_select (optimize loop, single rank): only the statistics are launched behind the linearise; setNewFrameEnergyTH is recorded in
// e->pend_sel and rides as one more workgroup in the k_ef_stitch launch of the next loop body (ef_launch_solve) -- its result is needed by
// the next linearise (accepted step) or by the re-classification in that body's k_ef_tail_resub (rejected step), not before -- instead of
// sitting between the statistics and the host's decision on the stream (one workgroup of serial passes, ~7 us).
static int linearize_launch_kernels(sdvgn_ef* e) {   // first half: the linearise itself (+ the point statistics when they are not trivially 0)
    if (!e->havePrecalc) return SDVGN_E_STATE;
    ef_flush_pending(e);   // a pending select reads, a pending re-classification feeds, planes this linearise is about to overwrite / read
    e->lin_partials = ef_launch_linearize(e);
    e->lin_nL = 0;
    if (e->deltaF_nonzero || e->has_linearized) {
        e->lin_nL = (e->nP + 255) / 256;
        k_ef_point_stats<<<e->lin_nL, 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->phost_dev, e->stats_partial);
    }
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}
static double host_prior_energy(const sdvgn_ef* e) {   // calcLEnergyF_MT: frame + calib priors, of the host mirror's state
    double En = 0;
    for (const FrameH& f : e->frames) for (int i = 0; i < 6; ++i) En += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
    { float a = 0; for (int i = 0; i < 4; ++i) a += e->C.cDeltaF[i] * (float)e->cPrior[i] * e->C.cDeltaF[i]; En += a; }
    return En;
}
// final_body: the loop's last body by count -- nothing follows that could carry the deferred select (the next body's k_ef_stitch does
// otherwise), so it runs beside the statistics in this launch instead of as a launch of its own after the host has seen the verdict
// with_apply: applyRes of this linearisation in the same launch, unconditionally (the call's initial linearizeAll + applyRes: statistics
// workgroup + apply workgroups side by side, nothing to wait for) -- single rank, shared stream only
// fused: applyRes of this linearisation was computed by the linearise itself (EFArrays::rflags_w): sums, accept test (+ the select in the last body) only
// merged_out (fused, not the final body): nothing is launched -- the arguments of the statistics workgroup go to *merged_out and the caller makes it workgroup 0 of
// the next body's accumulate (k_ef_acc_stats); the bookkeeping (sequence number of the host's flag, the pending select) is this function's either way
static int linearize_launch_stats(sdvgn_ef* e, bool defer_select, const DecideArgs* dec = nullptr, bool final_body = false, bool with_apply = false,
                                  bool fused = false, StatsLaunch* merged_out = nullptr) {   // second half: the sums (+ threshold select)
    const int n_partials = e->lin_partials, nL = e->lin_nL;
    const int nS = (e->nP + 63) / 64;
    const double* ps = e->stats_partial + (e->nP / 64 + 2);
    if (ef_sharded(e)) {
        k_ef_sum_stats<<<1, 256, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_dev, nullptr, 0);
        // ranks hold disjoint host-frame shards: ONE all-reduce carries the four sums and the candidates of setNewFrameEnergyTH's quantile
        ef_launch_pack_th(e);
        { const int rca = ef_allreduce(e, e->stats_dev, 4 + e->nP); if (rca) return rca; }
        k_ef_copy_publish<<<1, 256, 0, e->stream>>>(e->stats_dev, e->stats_host, 4, e->done_ctr, e->flags_host + 2, ++e->seq_stats);
        ef_launch_select_th(e, true);
    } else {
        // statistics (the host waits for their flag; pinned memory, no copy engine) and setNewFrameEnergyTH (only the NEXT linearise
        // needs it) side by side in one launch -- or, deferred, the select as a workgroup of the next body's k_ef_stitch
        SelArgs a{};
        a.nF = e->nF; a.nP = e->nP;
        ef_owned_points(e, a.own0, a.own1);
        a.rflags = e->rflags; a.wo = e->A.renergy_wo; a.th_prev = e->A.frameTH_r; a.th_out = e->A.frameTH_w;
        a.log_slot = e->th_log ? e->th_log + (e->th_log_n++ % kThLog) : nullptr;
        const DecideArgs none{0, 0, 0, nullptr, 0, nullptr, 0, nullptr};
        if (fused) {
            const bool sel_now = final_body || !defer_select;
            if (merged_out && sel_now) return SDVGN_E_STATE;
            if (merged_out) *merged_out = StatsLaunch{e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host, e->flags_host + 2, ++e->seq_stats, dec ? *dec : none};
            else {
                // (the loop's last body: the select as a launch of its own BEHIND the sums -- the host waits for the sums' flag, not for the 8 us of serial passes of
                // the select, which then run while the call returns; as the second workgroup of this launch they set its duration and the host's wait)
                k_ef_stats_select<<<1, kSelLanes, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host,
                                                                 e->flags_host + 2, ++e->seq_stats, a, dec ? *dec : none);
                if (sel_now) k_ef_select_th<0><<<1, kSelLanes, 0, e->stream>>>(a.nF, a.nP, a.own0, a.own1, a.rflags, a.wo, nullptr, a.th_prev, a.th_out, a.log_slot);
            }
            if (sel_now) defer_select = false;
        } else
        if (dec && dec->verdict && final_body && !e->own_stream) {
            const size_t slots = (size_t)e->nF * e->nP;
            k_ef_stats_apply_select<<<2 + (unsigned)((slots + kSelLanes - 1) / kSelLanes), kSelLanes, 0, e->stream>>>(
                e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host, e->flags_host + 2, ++e->seq_stats, *dec, e->nF, e->nP, e->A,
                e->precalc_dev, e->phost_dev, a);
            defer_select = false;   // taken in this launch
        } else
        if (dec && dec->verdict && (e->own_stream || final_body)) {
            // a window that runs beside others (sdvgn_ef_optimize_batch): the same two steps as two launches -- the apply workgroups would
            // otherwise sit on the CUs polling the verdict word while other windows' kernels wait for a place
            const size_t slots = (size_t)e->nF * e->nP;
            DecideArgs d2 = *dec; d2.verdict = nullptr;
            k_ef_stats_select<<<final_body ? 2 : 1, kSelLanes, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host, e->flags_host + 2, ++e->seq_stats, a, d2);
            k_ef_apply<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->nF, e->nP, e->A, e->precalc_dev, e->phost_dev, e->accept_dev);
            if (final_body) defer_select = false;   // taken in this launch
        } else if (with_apply && !dec) {
            const size_t slots = (size_t)e->nF * e->nP;
            k_ef_stats_apply<<<1 + (unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host,
                                                                                       e->flags_host + 2, ++e->seq_stats, none, e->nF, e->nP, e->A, e->precalc_dev,
                                                                                       e->phost_dev);
        } else if (dec && dec->verdict) {   // statistics + accept test + conditional applyRes of the trial set in one launch
            const size_t slots = (size_t)e->nF * e->nP;
            k_ef_stats_apply<<<1 + (unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host,
                                                                                       e->flags_host + 2, ++e->seq_stats, *dec, e->nF, e->nP, e->A, e->precalc_dev,
                                                                                       e->phost_dev);
        } else
        k_ef_stats_select<<<defer_select ? 1 : 2, kSelLanes, 0, e->stream>>>(e->energy_partial, n_partials, e->stats_partial, nL, ps, nS, e->stats_host,
                                                                              e->flags_host + 2, ++e->seq_stats, a, dec ? *dec : none);
        if (defer_select) { e->pend_sel = a; e->pend_sel_valid = true; }
    }
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}
static int linearize_launch(sdvgn_ef* e, bool defer_select = false) {
    const int rc = linearize_launch_kernels(e);
    return rc ? rc : linearize_launch_stats(e, defer_select);
}
static int linearize_wait(sdvgn_ef* e, double* energy, double* EL, double* sumID, double* sumNID) {
    HIPCHK(wait_flag(e->flags_host + 2, e->seq_stats, e->stream));
    if (ef_wait_error(e)) return ef_state_failure(e, "a workgroup gave up an intra-launch wait (statistics)");
    *energy = e->stats_host[0];
    const double En = host_prior_energy(e);   // calcLEnergyF_MT: frame + calib priors on the host, point part from the device
    *EL = En + (double)(float)e->stats_host[1];
    if (sumID) *sumID = e->stats_host[2];
    if (sumNID) *sumNID = e->stats_host[3];
    return SDVGN_OK;
}
static int linearize_and_stats(sdvgn_ef* e, double* energy, double* EL, double* sumID, double* sumNID, bool defer_select = false) {
    const int rc = linearize_launch(e, defer_select);
    return rc ? rc : linearize_wait(e, energy, EL, sumID, sumNID);
}

// ---- sharded window, ONE collective per loop body (BASELINE.json north_star: "a single RCCL all-reduce ... before the tiny
// Cholesky").  The accept test needs the global energy of the trial linearisation, the next solve the global accumulators of the accepted
// state: one message carries both if the accumulate is SPECULATIVE -- the trial is applied (what it overwrites is backed up), this rank's
// shard is accumulated as if the step had been accepted, and [accumulators | 4 statistics | quantile candidates] go out as one all-reduce.
// accept: the message buffer becomes the current accumulators (nothing else to do); reject: the apply is taken back (k_ef_apply_revert), the
// per-point planes are recomputed for the kept state (an accumulate without its reduce: they are local), and the kept message buffer -- whose
// system differs from the next one only in lambda -- serves the next solve.
static inline bool ef_one_collective(const sdvgn_ef* e) { return ef_sharded(e) && e->coll[0] != nullptr; }
static void ef_use_coll(sdvgn_ef* e, int p) { e->acc_dev = e->coll[p]; e->stats_dev = e->coll[p] + acc_count(e); }
static int ef_sharded_message(sdvgn_ef* e, int p, bool speculative) {
    ef_use_coll(e, p);
    const int nS = (e->nP + 63) / 64;
    const double* ps = e->stats_partial + (e->nP / 64 + 2);
    k_ef_sum_stats<<<1, 256, 0, e->stream>>>(e->energy_partial, e->lin_partials, e->stats_partial, e->lin_nL, ps, nS, e->stats_dev, nullptr, 0);
    ef_launch_pack_th(e);
    const size_t slots = (size_t)e->nF * e->nP;
    if (speculative) k_ef_apply_backup<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->nF, e->nP, e->A, e->precalc_dev, e->phost_dev, e->apply_bak);
    else k_ef_apply<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->nF, e->nP, e->A, e->precalc_dev, e->phost_dev, nullptr);
    int rc = ef_accumulate(e, /*with_reduce=*/true);
    if (rc) return rc;
    if ((rc = ef_allreduce(e, e->coll[p], (int)acc_count(e) + 4 + e->nP))) return rc;
    k_ef_copy_publish<<<1, 256, 0, e->stream>>>(e->stats_dev, e->stats_host, 4, e->done_ctr, e->flags_host + 2, ++e->seq_stats);
    ef_launch_select_th(e, true);
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

// FullSystem::optimize, the loop (FullSystemOptimize.cpp:344-458).  The window's state lives on the device: per loop body the host
// only launches -- accumulate + reduce, stitch (+ the previous linearisation's threshold select), LDL^T tail (+ the re-classification
// after a rejected step), resubstitute + step + precalc table, linearize -- mirrors the step from x (pinned memory) while the GPU
// linearises, hands its parts of the accept / reject comparison to the last launch (k_ef_stats_apply: statistics, the test itself and the
// conditional applyRes), waits ONCE for the sums + verdict and does its own bookkeeping of the accepted or restored state.
int sdvgn_ef_optimize(sdvgn_ef* e, int mnumOptIts, int flags, double* trace, int trace_stride, int trace_cap) {
    if (!e || !e->haveAdjoints || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    const int nF = e->nF, n = CPARS + 6 * nF;
    if (nF < 2) return 0;
    const bool fixed_its = (flags & 1) != 0;   // benchmark mode: exactly mnumOptIts loop bodies, no early break
    const bool relinearize_on_reject = (flags & 2) != 0;   // run the reference's redundant re-linearisation literally (A/B timing, tests)
    const bool reuse_after_reject = (flags & 4) != 0;      // opt-in: the solve after a rejected step reuses the stitched system (see header)
    const bool no_spec_solve = (flags & 16) != 0;          // A/B and tests: do not solve the rejected case ahead on the side stream
    e->pend_sel_valid = e->pend_rc_valid = e->pend_stats_valid = false;           // nothing of an earlier (failed) call is carried over
    { const int rce = ef_next_nogood_epoch(e, e->stream); if (rce) return rce; }      // sdvgn_ef_get_point_nogood: "during THIS call"
    e->time_lin = (flags & 8) != 0;                          // measurement: event pair around every k_ef_linearize launch
    e->lin_ev_used = 0; e->lin_ms.clear();
    struct TimeLinGuard { sdvgn_ef* e; ~TimeLinGuard() { e->time_lin = false; } } time_lin_guard{e};
    bool prev_rejected_clean = false;
    if (!fixed_its && nF < 3) mnumOptIts = 100;
    if (!fixed_its && nF < 4) mnumOptIts = 75;
    int rc;
    static const bool opt_timing = getenv("SDVGN_OPT_TIMING") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    auto us_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
    if (!e->havePrecalc && (rc = ef_upload_precalc(e))) return rc;
    if ((rc = ef_sync_window(e)) || (rc = ef_sync_state(e))) return rc;
    const double tt_sync = us_since(tt0);
    struct CalibGuard { sdvgn_ef* e; ~CalibGuard() { e->A.calib = nullptr; } } calib_guard{e};   // outside the loop the kernels take EFConst by value
    e->A.calib = e->calib_dev + e->st_cur;
    // resetOOB of every active residual (FullSystemOptimize.cpp:349-353) is folded into the call's first linearise + apply (EFArrays::reset_oob)
    struct ResetGuard { sdvgn_ef* e; ~ResetGuard() { e->A.reset_oob = 0; } } reset_guard{e};
    e->A.reset_oob = 1;
    double lastEnergy, lastEnergyL, lastEnergyM;
    e->th_log_n = 0;
    ef_select_new_set(e, e->new_cur, e->new_cur);
    ef_refresh_frame_deltas(e);
    const double tt_pre = us_since(tt0);
    const bool defer = !ef_sharded(e) && !relinearize_on_reject;   // the literal variant keeps the reference's order of launches
    // linearizeAll + applyRes: the apply does not depend on the sums, and neither does the first loop body's accumulate / solve / linearise --
    // the host's parts of the energies (priors, M energy: functions of the host mirror, which the first body's step moves) are taken now,
    // the device's sums are fetched when the first accept test needs them (take_initial_energies), by then long there
    const bool onecoll = ef_one_collective(e) && !relinearize_on_reject;
    struct CollGuard { sdvgn_ef* e; double* a; double* s; ~CollGuard() { e->acc_dev = a; e->stats_dev = s; } } coll_guard{e, e->acc_dev, e->stats_dev};
    // applyRes fused into the linearise (sdvgn_ef::rflags_alt).  The batched launch sequence always runs that way (backend_lockstep.inc: no
    // workgroups that poll a verdict, one pass over the residual planes less).  For ONE window it is a wash -- the apply workgroups run beside the
    // statistics workgroup anyway, the linearise grows by the 30 bytes per residual applyRes writes (in-loop 16.1 -> 17.0 us, statistics launch
    // 5.7 -> 5.4 us, headline 15.0 k it/s either way; profiles/r04_notes.txt) -- so the single-window loop keeps applyRes as workgroups of the
    // statistics launch unless SDVGN_FUSED_APPLY is set (tests run both: bit-identical)
    // Round 6: fused is the default -- with it the accumulate of the next body does not read anything the accept test decides, and the test itself becomes a workgroup
    // of that accumulate's launch (k_ef_acc_stats; SDVGN_DEBUG_FLAGS bit 9 keeps the statistics a launch of their own).  SDVGN_FUSED_APPLY=0: the old form.
    const char* fused_env = getenv("SDVGN_FUSED_APPLY");
    const bool fused = defer && !e->own_stream && !ef_sharded(e) && e->rflags_alt && !e->deltaF_nonzero && !(fused_env && fused_env[0] == '0');
    const bool merge_stats = fused && !(e->C.debug_flags & 512);
    struct ApplyTargetGuard { sdvgn_ef* e; ~ApplyTargetGuard() { ef_set_apply_target(e, false); } } apply_target_guard{e};
    if (onecoll) {
        // the call's one extra collective: initial linearizeAll + applyRes + accumulate, their sums and accumulators in one message
        if ((rc = linearize_launch_kernels(e)) || (rc = ef_sharded_message(e, e->coll_cur, /*speculative=*/false))) return rc;
    } else
    if (fused) {                     // linearise + applyRes in one kernel, then the sums; the applied copies become the current ones at once
        if ((rc = ef_sync_applied(e))) return rc;
        ef_set_apply_target(e, true);
        rc = linearize_launch_kernels(e);
        ef_set_apply_target(e, false);
        // (merge_stats: the sums are not launched here -- they ride in the first body's accumulate, ef_accumulate / take_initial_energies)
        if (rc || (rc = linearize_launch_stats(e, defer, nullptr, false, false, /*fused=*/true, (merge_stats && mnumOptIts > 0) ? &e->pend_stats : nullptr))) return rc;
        e->pend_stats_valid = merge_stats && mnumOptIts > 0;
        ef_flip_applied(e);
    } else
    if (defer && !e->own_stream) {   // linearise, then its statistics and applyRes in ONE launch (they do not depend on each other)
        if ((rc = linearize_launch_kernels(e)) || (rc = linearize_launch_stats(e, defer, nullptr, false, /*with_apply=*/true))) return rc;
    } else
    if ((rc = linearize_launch(e, defer)) || (rc = sdvgn_ef_apply_res(e))) return rc;
    e->A.reset_oob = 0;
    const double En_initial = host_prior_energy(e);
    lastEnergyM = calc_M_energy(e);
    bool initial_pending = true;
    auto take_initial_energies = [&]() -> int {
        if (!initial_pending) return SDVGN_OK;
        initial_pending = false;
        { const int rcf = ef_flush_stats(e); if (rcf) return rcf; }      // (normally launched long ago, inside the first body's accumulate)
        HIPCHK(wait_flag(e->flags_host + 2, e->seq_stats, e->stream));
        lastEnergy = e->stats_host[0];
        lastEnergyL = En_initial + (double)(float)e->stats_host[1];       // linearize_wait's expression, with the mirror as it was at the launch
        return SDVGN_OK;
    };
    const double tt_lin = us_since(tt0);
    if (opt_timing) fprintf(stderr, "[sdvgn] optimize pre-loop: uploads %.1f | reset_oob launch %.1f | linearize + stats + apply + wait %.1f | M energy %.1f us\n",
                            tt_sync, tt_pre - tt_sync, tt_lin - tt_pre, us_since(tt0) - tt_lin);
    double lambda = 1e-1;
    const float stepsize = 1, thOpt = 1.2f;
    std::vector<double> x(n);
    int it = 0;
    e->iter_us.clear();
    e->n_accepted = 0;
    e->n_spec_launched = e->n_spec_used = 0;
    e->n_merged_tests = e->n_pre_acc = 0;
    std::vector<int> th_idx;   // th_log slot of every trial linearisation (trace only)
    bool host_restore_pending = false;
    bool pre_accumulated = false;   // the accumulate of the coming body was queued behind the previous body's accept test (AccAlt)
    // the rejected case solved ahead (ef_launch_spec_solve): single rank, shared stream, the product's default loop
    const bool spec_enabled = defer && !e->own_stream && !no_spec_solve && !reuse_after_reject && e->side != nullptr;
    bool spec_pending = false;      // this body's rejected-case solve has been queued on the side stream
    bool spec_use = false;          // this body takes its solution from the solve its predecessor queued
    // an error exit from inside the loop leaves launches queued whose host-side bookkeeping (set flips, copy swaps) did not happen: the host
    // mirror goes back to the backed-up state and the handle refuses further work until the window is loaded again (sdvgn_ef_set_frames)
    struct LoopGuard {
        sdvgn_ef* e; bool* restore_pending; bool ok = false;
        ~LoopGuard() {
            if (ok) return;
            if (*restore_pending) {
                calib_set_value(e, e->value_backup);
                for (FrameH& f : e->frames) frame_set_state(f, f.state_backup);
            }
            hipStreamSynchronize(e->stream);
            e->sys_valid = false; e->haveAdjoints = false; e->havePrecalc = false;   // E_STATE from every compute entry point until reloaded
        }
    } loop_guard{e, &host_restore_pending};
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
        const auto t_iter = std::chrono::steady_clock::now();
        g_pt.start();
        if (!host_restore_pending) {                                                      // backupState (after a rejected step the backup still IS the state)
            for (int i = 0; i < 4; ++i) e->value_backup[i] = e->value[i];
            for (FrameH& f : e->frames) for (int i = 0; i < 10; ++i) f.state_backup[i] = f.state[i];
        }
        const bool zero_differs = e->deltaF_nonzero;   // idepth != idepth_zero before this trial (only possible right after a load)
        // solveSystemF + doStepFromBackup, all on the device: the trial state (frame states, calib, precalc table, idepths) goes to the
        // second copies; resubstitute also backs up the idepths and applies the point step
        const bool reuse = reuse_after_reject && prev_rejected_clean && e->sys_valid;
        if (onecoll) ef_use_coll(e, e->coll_cur);                                        // the reduced accumulators of the current state
        const bool from_spec = spec_use;                                                  // the predecessor was rejected and this very solve ran ahead
        spec_use = false;
        const int use_buf = e->spec_last_buf, use_seq = e->spec_last_seq;
        if (from_spec) { ++e->n_spec_used; if ((rc = ef_launch_spec_rest(e, iteration, lambda, stepsize))) return rc; }
        else if ((rc = ef_launch_solve(e, iteration, lambda, /*do_step=*/true, stepsize, reuse, /*accumulated=*/pre_accumulated || onecoll))) return rc;
        pre_accumulated = false;
        ef_swap_point_copies(e);                                                          // the stepped idepths are the ones every later launch reads
        e->deltaF_nonzero = false;
        std::swap(e->precalc_dev, e->precalc_alt);
        const int st_trial = 1 - e->st_cur;
        e->A.calib = e->calib_dev + st_trial;
        ef_select_new_set(e, 1 - e->new_cur, e->new_cur);                                 // trial linearisation goes to the other set
        th_idx.push_back(e->th_log_n % kThLog);
        spec_pending = false;
        // ... and this body's own rejected case goes to the side stream now: same system (it is in SolveSys::tri), 100 x the damping, next iteration
        // (queued AHEAD of the trial linearise, so that its one workgroup has its CU before the linearise fills the chip; behind it: measured, no difference)
        if (spec_enabled && !zero_differs && e->sys_valid && iteration + 1 < mnumOptIts) {
            if ((rc = ef_launch_spec_solve(e, iteration + 1, lambda * 1e2, /*main_solve_in_flight=*/!from_spec))) return rc;
            spec_pending = true;
            ++e->n_spec_launched;
        }
        // device-side accept test: the statistics launch waits for the host's parts of the comparison (after the mirror below), the
        // linearise does not
        const bool dev_decide = defer && !zero_differs;
        bool stats_deferred = false;
        DecideArgs dec_deferred{};
        if (!dev_decide && (rc = take_initial_energies())) return rc;          // (that path launches its statistics right behind the linearise)
        const bool spec = onecoll && !zero_differs;
        const bool fused_body = fused && dev_decide;       // the trial linearise leaves its applyRes in the second copies
        if (spec) {
            // trial linearise, then its sums + the speculative apply / accumulate and the body's ONE collective (ef_sharded_message)
            if ((rc = linearize_launch_kernels(e)) || (rc = ef_sharded_message(e, 1 - e->coll_cur, /*speculative=*/true))) return rc;
        } else {
            if (fused_body) ef_set_apply_target(e, true);
            rc = dev_decide ? linearize_launch_kernels(e) : linearize_launch(e, defer);
            ef_set_apply_target(e, false);
            if (rc) return rc;
        }
        if (dev_decide) {
            // the sums, the accept test and -- in the same launch -- applyRes of the trial linearisation, queued right behind the linearise: what
            // the test needs from the stepped state (prior energy, M energy) was formed on the device beside the step (step_energy_body), what it
            // needs from the last accepted state (the right-hand side) the host holds, so the host has not seen x yet and need not
            if ((rc = take_initial_energies())) return rc;                     // first body: the call's initial sums, before this body's overwrite them
            DecideArgs dec;
            dec.En = dec.EM = 0; dec.en_em = e->en_em_dev + 2 * st_trial;
            dec.rhs = lastEnergy + lastEnergyL + lastEnergyM; dec.accept_dev = e->accept_dev; dec.on = 1;
            dec.verdict = (unsigned*)(e->accept_dev + 4); dec.seq = (unsigned)(++e->seq_verdict) & 0x7fffffffu;
            // (merge_stats: the launch waits until the host knows whether it queues the next body's accumulate -- `may_break` below needs the solve's sums --
            // and then rides in that launch; the linearise it follows runs for 16 us from here)
            stats_deferred = merge_stats && fused_body && spec_pending && !reuse_after_reject && iteration + 1 < mnumOptIts && ef_acc_geom(e).sc_ppb == 64;
            dec_deferred = dec;
            if (!stats_deferred && (rc = linearize_launch_stats(e, defer, &dec, /*final_body=*/iteration + 1 == mnumOptIts, false, fused_body))) return rc;
        }
        g_pt.stop(PT_ACCUM);
        // x is in pinned memory as soon as the solve kernel is through: mirror doStepFromBackup on the host (FrameHessian::setState,
        // CalibHessian::setValue -- the same double-precision operations the device performed) while the GPU linearises
        if ((rc = from_spec ? ef_wait_spec_solve(e, use_buf, use_seq, x.data()) : ef_wait_solve(e, x.data()))) return rc;
        g_pt.stop(PT_D2H);
        {
            double v[4];
            for (int i = 0; i < 4; ++i) v[i] = e->value_backup[i] + stepsize * (-x[i]);
            calib_set_value(e, v);
            for (int h = 0; h < nF; ++h) {
                FrameH& f = e->frames[h];
                double st[10];
                for (int i = 0; i < 6; ++i) st[i] = f.state_backup[i] + (double)stepsize * (-x[CPARS + 6 * h + i]);
                for (int i = 6; i < 10; ++i) st[i] = f.state_backup[i];
                frame_set_state(f, st);
            }
            ef_refresh_frame_deltas(e);
            host_restore_pending = false;     // the mirror holds the new trial state
        }
        float sumT = from_spec ? e->sol_spec[use_buf].sumT : e->sol_host->sumT, sumR = from_spec ? e->sol_spec[use_buf].sumR : e->sol_host->sumR;
        double newEnergyM = calc_M_energy(e);                                  // (host mirror: the value itself in the host-decided paths, the cross-check below otherwise)
        const double En_host = host_prior_energy(e);
        if ((rc = take_initial_energies())) return rc;
        if (dev_decide) {
            // the next body's accumulate, queued before the verdict is known (AccAlt): its arguments are the accepted case -- the state
            // this body's launches run on --, the kept copies go along for the rejected one
            pre_accumulated = false;
            // (not when this body may be the call's last one -- the step is small enough for `canbreak`, whose second half the host only
            // learns with the sums: the per-point planes must hold what the LAST executed solveSystemF left, like the reference's EFPoints)
            const bool may_break = !fixed_its && iteration >= 1 && sqrtf(sumR / nF) < 0.00005 * thOpt;
            if (stats_deferred && may_break) {       // no accumulate to ride in: the statistics as a launch of their own after all
                if ((rc = linearize_launch_stats(e, defer, &dec_deferred, /*final_body=*/false, false, fused_body))) return rc;
                stats_deferred = false;
            }
            if (!reuse_after_reject && !may_break && iteration + 1 < mnumOptIts && ef_acc_geom(e).sc_ppb == 64) {
                // (when the rejected case has been solved ahead, a rejection leaves this accumulate without a reader: it returns at once)
                // (fused applyRes: the accepted case reads the copies the trial linearise wrote, the rejected one the kept copies)
                if (fused_body) ef_flip_applied(e);
                const AccAlt alt{e->accept_dev, e->pid_alt, e->pidz_alt, e->pdeltaF_alt, e->calib_dev + e->st_cur, e->precalc_alt, spec_pending ? 1 : 0,
                                 fused_body ? e->rflags_alt : nullptr, fused_body ? e->rstate_alt : nullptr, fused_body ? e->renergy_alt : nullptr,
                                 fused_body ? e->JpJd_alt : nullptr, stats_deferred ? dec_deferred.verdict : nullptr, dec_deferred.seq};
                if (stats_deferred) {
                    StatsLaunch st;
                    if ((rc = linearize_launch_stats(e, defer, &dec_deferred, /*final_body=*/false, false, fused_body, &st))) return rc;
                    rc = ef_accumulate(e, /*with_reduce=*/true, &alt, &st);
                    stats_deferred = false;
                    ++e->n_merged_tests;
                } else
                rc = ef_accumulate(e, /*with_reduce=*/true, &alt);
                if (fused_body) ef_flip_applied(e);        // (back: the verdict is not known yet)
                if (rc) return rc;
                pre_accumulated = true;
                ++e->n_pre_acc;
            }
            if (stats_deferred && (rc = linearize_launch_stats(e, defer, &dec_deferred, /*final_body=*/false, false, fused_body))) return rc;   // (cannot happen: same conditions)
        }
        g_pt.stop(PT_STEP);
        double newEnergy, newEnergyL, sID, sNID;
        if ((rc = linearize_wait(e, &newEnergy, &newEnergyL, &sID, &sNID))) return rc;
        if (dev_decide) {
            // the device formed the state's parts of the energies itself (stats_host[5], [7]); the host's mirror must give the same doubles -- the
            // same IEEE operations in the same order on the same numbers (cannot differ; if it ever does, the call fails instead of drifting)
            // (bit patterns, not values: a non-finite energy -- an overflowing prior term, 0 * inf in HM d -- is not a mismatch; the reference just
            // rejects such a step, FullSystemOptimize.cpp:420: NaN < rhs is false, and so does the accept test below)
            auto same_bits = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (std::isnan(a) && std::isnan(b)); };
            if (!same_bits(e->stats_host[5], En_host)) return ef_state_failure(e, "prior energy of the stepped state: device vs host mirror", e->stats_host[5], En_host);
            if (!same_bits(e->stats_host[7], newEnergyM)) return ef_state_failure(e, "M energy of the stepped state: device vs host mirror", e->stats_host[7], newEnergyM);
            newEnergyL = e->stats_host[5] + (double)(float)e->stats_host[1];
            newEnergyM = e->stats_host[7];
        }
        g_pt.stop(PT_LIN);
        sumR /= nF; sumT /= nF;
        const float sumNID = (float)sNID / (float)e->nP;
        const bool canbreak = sqrtf(sumR) < 0.00005 * thOpt && sqrtf(sumT) * sumNID < 0.00005 * thOpt;
        const bool accept_host = newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM;
        const bool accept = dev_decide ? (e->stats_host[4] != 0.0) : accept_host;
        if (dev_decide && accept != accept_host) return ef_state_failure(e, "accept test: device vs host", accept, accept_host);   // the same IEEE operations on the same numbers: cannot happen
        if (trace && iteration < trace_cap) {
            double* tr = trace + (size_t)iteration * trace_stride;
            tr[0] = iteration; tr[1] = lambda; tr[2] = accept; tr[3] = newEnergy; tr[4] = newEnergyL; tr[5] = newEnergyM; tr[6] = canbreak;
            for (int i = 0; i < n && 7 + i < trace_stride; ++i) tr[7 + i] = x[i];
        }
        it = iteration + 1;
        if (accept) {
            ++e->n_accepted;
            e->new_cur = 1 - e->new_cur;                                                  // the trial sets become the current ones
            e->st_cur = st_trial;
            ef_select_new_set(e, e->new_cur, e->new_cur);
            if (fused_body) ef_flip_applied(e);                               // applyRes: the copies the trial linearise wrote are the residuals' state now
            if (spec) e->coll_cur = 1 - e->coll_cur;                          // the speculative message IS the accepted state's: applied, accumulated, reduced
            else if (!dev_decide && (rc = sdvgn_ef_apply_res(e))) return rc;  // (device-side test: the conditional apply is already queued)
            if (onecoll && !spec) {   // (idepth_zero differed before this trial: the body ran the two-collective way; rebuild the current message)
                ef_use_coll(e, e->coll_cur);
                if ((rc = ef_accumulate(e, true)) || (rc = ef_allreduce(e, e->acc_dev, (int)acc_count(e)))) return rc;
            }
            lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
            lambda *= 0.25;
            prev_rejected_clean = false;
        } else {
            // loadSateBackup: the idepths, the precalc table, the calib floats and the frame states of the backed-up state are still in
            // the copies the trial did not write -- switch back (no launch); the host mirror is recomputed from the backup
            // (the host mirror of the restored state is only needed by a literal re-linearisation or when the loop ends: the next body
            // steps from the backup, which still is the state -- restoring it here would sit between the verdict and the next launches)
            auto restore_host_mirror = [&]() {
                calib_set_value(e, e->value_backup);
                for (FrameH& f : e->frames) frame_set_state(f, f.state_backup);
                ef_refresh_frame_deltas(e);
            };
            if (relinearize_on_reject || zero_differs) restore_host_mirror(); else host_restore_pending = true;
            if (spec) {   // take the speculative applyRes back before anything else touches the residual planes
                const size_t slots_r = (size_t)nF * e->nP;
                k_ef_apply_revert<<<(unsigned)((slots_r + 255) / 256), 256, 0, e->stream>>>(nF, e->nP, e->A, e->apply_bak);
            }
            ef_swap_point_copies(e);
            std::swap(e->precalc_dev, e->precalc_alt);
            e->A.calib = e->calib_dev + e->st_cur;
            // the reference also sets idepth_zero = idepth_backup here (FullSystemOptimize.cpp:276-277); the swapped-back copies
            // already satisfy that unless the window was loaded with idepth != idepth_zero and its very first step is rejected
            if (zero_differs && (rc = sdvgn_ef_point_step(e, 2, 0.f))) return rc;
            // The reference re-linearises here (`lastEnergy = linearizeAll(false)`, FullSystemOptimize.cpp:446-449).  With the state
            // restored exactly, that is a bit-for-bit recomputation of the previous accepted linearisation: its energies are the
            // lastEnergy* values still held, its state_New* planes are the current set (untouched by the trial), and the J it would
            // write into the not-owned buffer is overwritten by the next linearise before anything reads it.  So: switch back.
            // Two things do depend on that re-linearisation and are reproduced: (1) it classifies IN / OUTLIER under the threshold the
            // TRIAL's linearizeAll just set (setNewFrameEnergyTH, FullSystemOptimize.cpp:122) -- k_ef_reclassify repeats that decision
            // on the kept set; (2) its own setNewFrameEnergyTH recomputes the quantile of the same energies, i.e. restores the
            // threshold of the last accepted linearisation -- which is the kept set's threshold array.
            // Exception: if the restore just moved idepth_zero (zero_differs), the linearisation point of the centre projection and
            // deltaF changed, so the kept energies do not describe the restored state -- re-linearise like the reference.
            const int trial = 1 - e->new_cur;
            if (relinearize_on_reject || zero_differs) {
                ef_flush_pending(e);                       // the trial's threshold first: this re-linearisation classifies with it
                ef_select_new_set(e, e->new_cur, e->new_cur, /*th_read=*/trial);
                if ((rc = linearize_and_stats(e, &lastEnergy, &lastEnergyL, nullptr, nullptr))) return rc;
                lastEnergyM = calc_M_energy(e);
            } else {
                ef_select_new_set(e, e->new_cur, e->new_cur);
                ReclArgs ra;
                ra.nF = nF; ra.nP = e->nP; ra.P0_last = e->hostP0[nF - 1]; ra.np_last = e->hostP0[nF] - ra.P0_last;
                ra.rflags = e->rflags; ra.wo = e->A.renergy_wo; ra.rstate_new = e->A.rstate_new; ra.renergy_new = e->A.renergy_new;
                ra.phost = e->phost_dev; ra.precalc = e->precalc_dev; ra.th = e->th_dev + (size_t)trial * SDVGN_MAX_FRAMES;
                ra.err = e->A.err;
                if (defer) {
                    // nothing is launched here: the trial's select rides in the next body's k_ef_stitch, this re-classification in its
                    // k_ef_tail_resub -- both through before that body's linearise (which reads the kept set's state_NewEnergy for
                    // residuals that leave the image) -- or ef_flush_pending launches them when the loop ends
                    e->pend_rc = ra; e->pend_rc_valid = true;
                } else {
                    const int nthr = e->nP + ra.np_last * (nF - 1);
                    k_ef_reclassify<<<(nthr + 255) / 256, 256, 0, e->stream>>>(ra);
                    HIPCHK(hipGetLastError());
                }
            }
            ef_select_new_set(e, e->new_cur, e->new_cur);
            if (onecoll && !spec) {
                // (idepth_zero differed before this trial, so the body ran the two-collective way; the restore above moved idepth_zero, i.e.
                // deltaF and with it bdSum / the linearised terms of the CURRENT state: the kept message no longer describes it -- rebuild it,
                // like the accept branch does, instead of solving the next body on stale accumulators)
                ef_use_coll(e, e->coll_cur);
                if ((rc = ef_accumulate(e, /*with_reduce=*/true)) || (rc = ef_allreduce(e, e->acc_dev, (int)acc_count(e)))) return rc;
            }
            if (spec) {   // the per-point planes (Hdd / bd / Hcd sums, HdiF, bdSum: inputs of the next resubstitute) back to the kept state: they
                // are local to the rank, so an accumulate WITHOUT its reduce -- the kept message buffer stays as it is
                ef_use_coll(e, e->coll_cur);
                if ((rc = ef_accumulate(e, /*with_reduce=*/false))) return rc;
            }
            lambda *= 1e2;
            // the restored state is the one this body's system was built on, bit for bit (unless idepth_zero just changed, above)
            prev_rejected_clean = !zero_differs;
            spec_use = spec_pending && prev_rejected_clean;     // the next body's solution is (being) computed on the side stream
            if (spec_use) {
                bool valid = false;
                if ((rc = ef_spec_valid(e, &valid))) return rc;
                if (!valid) { spec_use = false; pre_accumulated = false; }   // (the accumulate queued ahead returned at once on this rejection: the body accumulates itself)
            }
        }
        g_pt.stop(PT_APPLY);
        e->iter_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_iter).count());
        if (!fixed_its && canbreak && iteration >= 1) break;
    }
    if ((rc = take_initial_energies())) return rc;                             // (a call of zero bodies)
    if ((rc = ef_drain_spec(e))) return rc;                                    // (normally one read of pinned memory: the last side-stream solve is through)
    ef_flush_pending(e);
    HIPCHK(hipGetLastError());
    if (ef_wait_error(e)) return ef_state_failure(e, "a workgroup gave up an intra-launch wait (end of the loop)");
    loop_guard.ok = true;
    if (host_restore_pending) {   // the loop ended on a rejected step: loadSateBackup for the host mirror
        calib_set_value(e, e->value_backup);
        for (FrameH& f : e->frames) frame_set_state(f, f.state_backup);
        ef_refresh_frame_deltas(e);
    }
    if (g_pt.on) sdvgn_debug_phase_report(it);
    if (e->time_lin) {
        HIPCHK(hipStreamSynchronize(e->stream));
        for (size_t k = 0; k + 1 < e->lin_ev_used; k += 2) { float ms = 0; hipEventElapsedTime(&ms, e->lin_events[k], e->lin_events[k + 1]); e->lin_ms.push_back(ms); }
    }
    if (trace && 7 + n < trace_stride) {   // frameEnergyTH of the newest frame as each trial linearizeAll left it
        HIPCHK(hipStreamSynchronize(e->stream));
        for (int i = 0; i < it && i < trace_cap; ++i) trace[(size_t)i * trace_stride + 7 + n] = e->th_log[th_idx[i]];
    }
    return it;
}

// Tail of FullSystem::optimize (FullSystemOptimize.cpp:460-470): setEvalPT on the newest frame (its optimised pose becomes the
// linearisation point, state = zero except the affine part), setAdjointsF, setPrecalcValues, linearizeAll(true) = linearize +
// applyRes(true) per active residual + the isNew bookkeeping of the points (:34-47) + setNewFrameEnergyTH + the toRemove list
// (:136-155: every residual that is not active afterwards is dropped; its slot ceases to exist here as well).
// B independent windows optimised side by side: one host thread per handle, every handle on its own HIP stream, so that the launch chains
// of different windows -- each a sequence of short, latency-bound kernels that fill a fraction of the chip -- overlap on the device.
// The threads are kept (a pool grown on demand): a call costs two condition-variable hand-offs per extra window.
namespace {
struct BatchPool {
    std::mutex call_mu;   // one batch call at a time (a second caller waits: the job list below belongs to the call in flight)
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> workers;
    struct Job { sdvgn_ef* e; int its, flags, rc; };
    std::vector<Job> jobs;
    size_t next = 0, done = 0;
    unsigned long long epoch = 0;
    bool quit = false;
    void worker() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_go.wait(lk, [&] { return quit || (epoch != seen && next < jobs.size()) ; });
            if (quit) return;
            while (next < jobs.size()) {
                const size_t k = next++;
                Job j = jobs[k];
                lk.unlock();
                const int rc = sdvgn_ef_optimize(j.e, j.its, j.flags, nullptr, 0, 0);
                lk.lock();
                jobs[k].rc = rc;
                if (++done == jobs.size()) cv_done.notify_all();
            }
            seen = epoch;
        }
    }
    ~BatchPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (std::thread& t : workers) t.join();
    }
};
static BatchPool& batch_pool() { static BatchPool* p = new BatchPool(); return *p; }   // (leaked on purpose: no join at library unload)
}  // namespace

#include "backend_lockstep.inc"

static bool lock_batch_ok(sdvgn_ef* const* handles, int B, int flags) {
    for (int b = 0; b < B; ++b) if (!lock_eligible(handles[b], flags) || handles[b]->device != handles[0]->device || handles[b]->arith != handles[0]->arith) return false;
    return true;
}
int sdvgn_ef_optimize_lockstep(sdvgn_ef* const* handles, int B, int mnumOptIts, int flags, int* its_out, double* trace, int trace_stride, int trace_cap) {
    if (!handles || B < 1 || B > 256 || mnumOptIts < 0) return SDVGN_E_ARG;
    for (int b = 0; b < B; ++b) {
        if (!handles[b]) return SDVGN_E_ARG;
        for (int c = 0; c < b; ++c) if (handles[c] == handles[b]) return SDVGN_E_ARG;
    }
    if (trace && (trace_stride < 7 || trace_cap < 1)) return SDVGN_E_ARG;
    if (!lock_batch_ok(handles, B, flags)) return SDVGN_E_ARG;
    return ef_optimize_lockstep(handles, B, mnumOptIts, flags, its_out, trace, trace_stride, trace ? trace_cap : 0);
}

int sdvgn_ef_optimize_batch(sdvgn_ef* const* handles, int B, int mnumOptIts, int flags, int* its_out) {
    if (!handles || B < 1 || B > 256) return SDVGN_E_ARG;
    for (int b = 0; b < B; ++b) {
        if (!handles[b]) return SDVGN_E_ARG;
        for (int c = 0; c < b; ++c) if (handles[c] == handles[b]) return SDVGN_E_ARG;   // a handle is single-threaded
    }
    // the default: ONE launch sequence for all B windows (backend_lockstep.inc); windows it does not take (and SDVGN_BATCH_THREADS=1, the A/B
    // switch of the benchmark) run as B sdvgn_ef_optimize calls on B host threads, below
    const bool force_threads = getenv("SDVGN_BATCH_THREADS") != nullptr;
    if (B > 1 && !force_threads && mnumOptIts >= 0 && lock_batch_ok(handles, B, flags))
        return ef_optimize_lockstep(handles, B, mnumOptIts, flags, its_out, nullptr, 0, 0);
    // sharded windows issue collectives on their communicator: several of them side by side would enqueue ncclAllReduce calls from several
    // threads / streams in an order the other ranks do not share -- one at a time (plain sdvgn_ef_optimize), never as a batch
    for (int b = 0; b < B && B > 1; ++b) if (ef_sharded(handles[b])) return SDVGN_E_ARG;
    if (B == 1) { const int rc = sdvgn_ef_optimize(handles[0], mnumOptIts, flags, nullptr, 0, 0); if (its_out) its_out[0] = rc; return rc < 0 ? rc : SDVGN_OK; }
    BatchPool& P = batch_pool();
    std::lock_guard<std::mutex> one_call(P.call_mu);
    std::unique_lock<std::mutex> lk(P.mu);
    while ((int)P.workers.size() < B - 1) P.workers.emplace_back([&P] { P.worker(); });
    P.jobs.clear();
    for (int b = 0; b < B; ++b) P.jobs.push_back(BatchPool::Job{handles[b], mnumOptIts, flags, 0});
    P.next = 0; P.done = 0; ++P.epoch;
    P.cv_go.notify_all();
    while (P.next < P.jobs.size()) {                    // the calling thread takes its share
        const size_t k = P.next++;
        BatchPool::Job j = P.jobs[k];
        lk.unlock();
        const int rc = sdvgn_ef_optimize(j.e, j.its, j.flags, nullptr, 0, 0);
        lk.lock();
        P.jobs[k].rc = rc;
        ++P.done;
    }
    P.cv_done.wait(lk, [&] { return P.done == P.jobs.size(); });
    int worst = SDVGN_OK;
    for (int b = 0; b < B; ++b) { if (its_out) its_out[b] = P.jobs[b].rc; if (P.jobs[b].rc < 0) worst = P.jobs[b].rc; }
    return worst;
}

int sdvgn_ef_optimize_finish(sdvgn_ef* e, double* lastEnergy_out, float* relbs_max, int* ngood_inc, unsigned char* removed) {
    if (!e || e->host_only || e->nP < 1 || e->nR < 0 || e->nF < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    const int nF = e->nF;
    static const bool fin_timing = getenv("SDVGN_OPT_TIMING") != nullptr;
    using fclk = std::chrono::steady_clock;
    fclk::time_point ft[6];
    ft[0] = fclk::now();
    FrameH& nf = e->frames[nF - 1];
    const double newStateZero[10] = {0, 0, 0, 0, 0, 0, nf.state[6], nf.state[7], 0, 0};
    nf.evalPT = nf.PRE_worldToCam;
    frame_set_state(nf, newStateZero);
    for (int i = 0; i < 10; ++i) nf.state_zero[i] = newStateZero[i];
    e->win_dirty = e->state_dirty = true; e->sys_valid = false;   // evalPT / state / state_zero of the newest frame moved
    int rc;
    if ((rc = sdvgn_ef_set_adjoints(e))) return rc;
    if ((rc = ef_upload_precalc(e))) return rc;
    ef_select_new_set(e, e->new_cur, e->new_cur);
    double energy = 0, EL = 0;
    ft[1] = fclk::now();
    // everything is queued before anything is waited for: linearise + its sums, applyRes, the per-point bookkeeping, ONE copy of the outputs
    if ((rc = linearize_launch(e))) return rc;
    if ((rc = sdvgn_ef_apply_res(e))) return rc;
    const size_t slots = (size_t)nF * e->nP, need = slots + 8 * (size_t)e->nP;
    if (need > e->fin_bytes) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->fin_dev) SDVGN_DFREE(e->fin_dev);
        if (e->fin_host) SDVGN_HFREE(e->fin_host);
        e->fin_dev = nullptr; e->fin_host = nullptr; e->fin_bytes = 0;
        const size_t cap = (std::max(need, (size_t)e->slots_cap + 8 * (size_t)e->max_points) + 15) / 16 * 16;
        HIPCHK(SDVGN_DMALLOC(&e->fin_dev, cap));
        HIPCHK(SDVGN_HMALLOC(&e->fin_host, cap));
        e->fin_bytes = cap;
    }
    float* relbs_dev = (float*)e->fin_dev;
    int* ngood_dev = (int*)e->fin_dev + e->nP;
    uint8_t* removed_dev = (uint8_t*)e->fin_dev + 8 * (size_t)e->nP;
    k_ef_finish_points<<<(e->nP + 255) / 256, 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->phost_dev, relbs_dev, ngood_dev, removed_dev);
    HIPCHK(hipGetLastError());
    // the three outputs lie back to back on the device: ONE copy into pinned memory, one wait
    static const bool fin_engine_copy = getenv("SDVGN_FINISH_ENGINE_COPY") != nullptr;   // (measurement: the copy as a hipMemcpyAsync)
    if (fin_engine_copy) HIPCHK(hipMemcpyAsync(e->fin_host, e->fin_dev, need, hipMemcpyDeviceToHost, e->stream));
    else {
        const int n16 = (int)((need + 15) / 16);          // (both blocks are allocated in multiples of 16 bytes beyond `need`: see `cap`)
        k_ef_copy_out<<<std::min(64, (n16 + 255) / 256), 256, 0, e->stream>>>((const uint4*)e->fin_dev, (uint4*)e->fin_host, n16);
        HIPCHK(hipGetLastError());
    }
    ft[2] = fclk::now();
    if ((rc = linearize_wait(e, &energy, &EL, nullptr, nullptr))) return rc;
    ft[3] = fclk::now();
    HIPCHK(hipStreamSynchronize(e->stream));
    ft[4] = fclk::now();
    const float* rb = (const float*)e->fin_host;
    const int* ng = (const int*)e->fin_host + e->nP;
    const uint8_t* rm = (const uint8_t*)e->fin_host + 8 * (size_t)e->nP;
    if (lastEnergy_out) *lastEnergy_out = energy;
    if (relbs_max) std::memcpy(relbs_max, rb, 4 * (size_t)e->nP);
    if (ngood_inc) std::memcpy(ngood_inc, ng, 4 * (size_t)e->nP);
    if (removed) {
        if (e->table_mode) std::memcpy(removed, rm, slots);    // by slot (target * nP + point index): the window is edited in place, there is no residual list
        else for (int i = 0; i < e->nR; ++i) removed[i] = rm[(size_t)e->r_slot[i]];
    }
    if (fin_timing) {
        ft[5] = fclk::now();
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(ft[b] - ft[a]).count(); };
        fprintf(stderr, "[sdvgn] optimize_finish: adjoints + precalc %.1f | launches %.1f | wait linearise %.1f | wait copy %.1f | outputs %.1f us\n",
                us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5));
    }
    return SDVGN_OK;
}

// ---- marginalisation: EFResidual::fixLinearizationF, EnergyFunctional::marginalizePointsF / dropPointsF / marginalizeFrame ----------
int sdvgn_ef_reset_oob(sdvgn_ef* e, const unsigned char* mask) {
    if (!e || e->nP < 1) return SDVGN_E_ARG;
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    const size_t slots = (size_t)e->nF * e->nP;
    if (mask) HIPCHK(hipMemcpyAsync(e->marg_mask_dev, mask, e->nP, hipMemcpyHostToDevice, e->stream));
    k_ef_reset_oob<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(slots, e->A, mask ? e->marg_mask_dev : nullptr, e->nP);
    HIPCHK(hipGetLastError());
    if (mask) HIPCHK(hipStreamSynchronize(e->stream));
    return SDVGN_OK;
}

int sdvgn_ef_fix_linearization(sdvgn_ef* e, const unsigned char* mask) {
    if (!e || !mask || e->nP < 1) return SDVGN_E_ARG;
    if (!e->havePrecalc || !e->haveAdjoints) return SDVGN_E_STATE;
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    HIPCHK(hipMemcpyAsync(e->marg_mask_dev, mask, e->nP, hipMemcpyHostToDevice, e->stream));
    const size_t slots = (size_t)e->nF * e->nP;
    k_ef_fix_linearization<<<(unsigned)((slots + 255) / 256), 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->phost_dev, e->marg_mask_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));   // `mask` may be freed by the caller
    e->has_linearized = true;
    return SDVGN_OK;
}

int sdvgn_ef_marginalize_points(sdvgn_ef* e, const unsigned char* marg, const unsigned char* drop) {
    if (!e || !marg || e->nP < 1) return SDVGN_E_ARG;
    if (!e->havePrecalc || !e->haveAdjoints) return SDVGN_E_STATE;
    if (ef_sharded(e)) return SDVGN_E_STATE;   // single-GPU entry point (the key-frame cycle around optimize), not part of cfg4
    EF_DEVICE(e);
    e->applied_synced = false;   // (writes the first copies of the planes applyRes owns: see sdvgn_ef::applied_synced)
    const int nF = e->nF, n = CPARS + 6 * nF, pairs = nF * nF, chunks = chunks_for_np(e);
    HIPCHK(hipMemcpyAsync(e->marg_mask_dev, marg, e->nP, hipMemcpyHostToDevice, e->stream));
    if (drop) HIPCHK(hipMemcpyAsync(e->drop_mask_dev, drop, e->nP, hipMemcpyHostToDevice, e->stream));
    int mx = 1;
    for (int h = 0; h < nF; ++h) mx = std::max(mx, e->hostP0[h + 1] - e->hostP0[h]);
    const int sc_chunks = std::min(kMaxChunks, (mx + 127) / 128);
    const int sc_ppb = ((mx + sc_chunks - 1) / sc_chunks + 63) / 64 * 64;
    const int n_top = chunks * pairs, n_pt = (e->nP + 63) / 64;
    // addPoint<2> + AccumulatedSCHessian::addPoint(p, false) over the flagged points (priorF *= setting_idepthFixPriorMargFac inside)
    k_ef_marg_stage1<<<n_top + n_pt, 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->phost_dev, e->top_partial, e->nres_partial, chunks, n_top,
                                                          e->marg_mask_dev, e->ppriorF);
    k_ef_marg_sc_gram<<<dim3(sc_chunks, nF), 256, 0, e->stream>>>(e->C, e->A, e->precalc_dev, e->sc_partial, sc_ppb, e->marg_mask_dev);
    k_ef_acc_reduce<<<acc_reduce_grid(pairs, nF), 256, 0, e->stream>>>(e->top_partial, pairs, chunks, e->sc_partial, nF, sc_chunks, e->nres_partial, e->acc_dev);
    k_ef_remove_points<<<(unsigned)(((size_t)nF * e->nP + 255) / 256), 256, 0, e->stream>>>(nF, e->nP, e->rflags, e->marg_mask_dev, drop ? e->drop_mask_dev : nullptr);
    HIPCHK(hipGetLastError());
    const int na = (int)acc_count(e);
    HIPCHK(hipMemcpyAsync(e->acc_host, e->acc_dev, sizeof(double) * na, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->acc_in_host = false;
    // stitchDouble(M, Mb, usePrior = false), SC stitch, HM += margWeightFac * (M - Msc), bM likewise (:545-566); the null-space
    // branches are off under setting_solverMode = SOLVER_ORTHOGONALIZE_X_LATER
    stitch_top(e, e->acc_host, /*use_prior=*/false);
    stitch_sc(e, e->acc_host + (size_t)pairs * kTopE);
    e->resInM += (int)e->acc_host[na - 1];
    if ((int)e->HM.size() != n * n) { e->HM.assign((size_t)n * n, 0); e->bM.assign(n, 0); }
    const double wfac = (double)(0.5f * 0.5f);   // setting_margWeightFac, settings.cpp:71
    for (size_t i = 0; i < (size_t)n * n; ++i) e->HM[i] += wfac * (e->HA[i] - e->Hsc[i]);
    for (int i = 0; i < n; ++i) e->bM[i] += wfac * (e->bA[i] - e->bsc[i]);
    e->win_dirty = true; e->sys_valid = false;   // the device-side solve reads HM / bM from the window block
    return SDVGN_OK;
}

int sdvgn_ef_get_marg_prior(sdvgn_ef* e, double* HM, double* bM) {
    if (!e || !HM || !bM || e->nF < 1) return SDVGN_E_ARG;
    const int n = CPARS + 6 * e->nF;
    if ((int)e->HM.size() != n * n) { std::memset(HM, 0, sizeof(double) * n * n); std::memset(bM, 0, sizeof(double) * n); return SDVGN_OK; }
    std::memcpy(HM, e->HM.data(), sizeof(double) * n * n);
    std::memcpy(bM, e->bM.data(), sizeof(double) * n);
    return SDVGN_OK;
}

static void inverse6(const double* A, double* Ainv) {   // Gauss-Jordan with partial pivoting (Mat66::inverse, EnergyFunctional.cpp:479)
    double a[6][12];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j); }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 12; ++j) std::swap(a[c][j], a[piv][j]);
        const double d = a[c][c];
        for (int j = 0; j < 12; ++j) a[c][j] /= d;
        for (int r = 0; r < 6; ++r) if (r != c) { const double f = a[r][c]; if (f != 0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = a[i][6 + j];
}

// EnergyFunctional::marginalizeFrame (:434-512), the algebra on HM / bM only: frame idx moves to the end, its prior is added, the
// Schur complement on the preconditioned last 6x6 block is taken.  Pure host function of the handle's HM, bM and frame prior; the
// caller rebuilds the window without that frame and installs the outputs with sdvgn_ef_set_marg_prior.
static void marginalize_frame_algebra(int nF, const std::vector<double>& HM, const std::vector<double>& bM, const double* prior6, const double* delta_prior6, int idx,
                                      double* HM_out, double* bM_out) {
    const int odim = CPARS + 6 * nF, ndim = odim - 6;
    std::vector<double> H = HM, b = bM;
    if ((int)H.size() != odim * odim) { H.assign((size_t)odim * odim, 0); b.assign(odim, 0); }
    std::vector<int> perm;
    for (int i = 0; i < odim; ++i) if (i < CPARS + 6 * idx || i >= CPARS + 6 * idx + 6) perm.push_back(i);
    for (int i = 0; i < 6; ++i) perm.push_back(CPARS + 6 * idx + i);
    std::vector<double> Hp((size_t)odim * odim), bp(odim);
    for (int i = 0; i < odim; ++i) { bp[i] = b[perm[i]]; for (int j = 0; j < odim; ++j) Hp[(size_t)i * odim + j] = H[(size_t)perm[i] * odim + perm[j]]; }
    for (int i = 0; i < 6; ++i) { Hp[(size_t)(ndim + i) * odim + ndim + i] += prior6[i]; bp[ndim + i] += prior6[i] * delta_prior6[i]; }
    std::vector<double> SVec(odim), SVecI(odim), Hs((size_t)odim * odim), bs(odim);
    for (int i = 0; i < odim; ++i) { SVec[i] = std::sqrt(std::fabs(Hp[(size_t)i * odim + i]) + 10); SVecI[i] = 1.0 / SVec[i]; }
    for (int i = 0; i < odim; ++i) { bs[i] = SVecI[i] * bp[i]; for (int j = 0; j < odim; ++j) Hs[(size_t)i * odim + j] = SVecI[i] * Hp[(size_t)i * odim + j] * SVecI[j]; }
    double hp[36], hpi[36];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) hp[i * 6 + j] = Hs[(size_t)(ndim + i) * odim + ndim + j];
    inverse6(hp, hpi);   // the reference's two `0.5f*(hpi+hpi)` statements (:478,480) are identities
    std::vector<double> bli((size_t)ndim * 6);
    for (int r = 0; r < ndim; ++r) for (int c = 0; c < 6; ++c) { double a = 0; for (int k = 0; k < 6; ++k) a += Hs[(size_t)(ndim + k) * odim + r] * hpi[k * 6 + c]; bli[(size_t)r * 6 + c] = a; }
    for (int r = 0; r < ndim; ++r) {
        for (int c = 0; c < ndim; ++c) { double a = 0; for (int k = 0; k < 6; ++k) a += bli[(size_t)r * 6 + k] * Hs[(size_t)(ndim + k) * odim + c]; Hs[(size_t)r * odim + c] -= a; }
        double a = 0; for (int k = 0; k < 6; ++k) a += bli[(size_t)r * 6 + k] * bs[ndim + k];
        bs[r] -= a;
    }
    for (int i = 0; i < odim; ++i) { bs[i] = SVec[i] * bs[i]; for (int j = 0; j < odim; ++j) Hs[(size_t)i * odim + j] = SVec[i] * Hs[(size_t)i * odim + j] * SVec[j]; }
    for (int r = 0; r < ndim; ++r) { bM_out[r] = bs[r]; for (int c = 0; c < ndim; ++c) HM_out[(size_t)r * ndim + c] = 0.5 * (Hs[(size_t)r * odim + c] + Hs[(size_t)c * odim + r]); }
}
int sdvgn_ef_marginalize_frame(sdvgn_ef* e, int idx, double* HM_out, double* bM_out) {
    if (!e || !HM_out || !bM_out || idx < 0 || idx >= e->nF || e->nF < 2) return SDVGN_E_ARG;
    marginalize_frame_algebra(e->nF, e->HM, e->bM, e->frames[idx].prior, e->frames[idx].delta_prior, idx, HM_out, bM_out);
    return SDVGN_OK;
}

#include "backend_window.inc"

int sdvgn_ef_get_res_toZero(sdvgn_ef* e, float* res_toZero2, unsigned char* isLinearized) {
    if (!e || !res_toZero2 || !isLinearized || e->host_only) return SDVGN_E_ARG;
    EF_DEVICE(e);
    const size_t slots = (size_t)e->nF * e->nP;
    std::vector<float> r2z(2 * slots);
    std::vector<uint8_t> fl(slots);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(r2z.data(), e->rres_toZero, sizeof(float) * 2 * slots, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fl.data(), e->rflags, slots, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nR; ++i) {
        const size_t s = (size_t)e->r_slot[i];
        res_toZero2[2 * i] = r2z[s]; res_toZero2[2 * i + 1] = r2z[slots + s];
        isLinearized[i] = (fl[s] & RF_LINEARIZED) ? 1 : 0;
    }
    return SDVGN_OK;
}

int sdvgn_debug_read_stamps(sdvgn_ef* e, unsigned long long* out, int cap_words) {
    if (!e || !out || !e->dbg_stamps) return 0;
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return SDVGN_E_NODEVICE;
    const int n = std::min<int>(cap_words, (int)kDbgStampWords);
    if (hipMemcpy(out, e->dbg_stamps, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) != hipSuccess) return SDVGN_E_NODEVICE;
    return n;
}

int sdvgn_ef_get_solve_status(sdvgn_ef* e) { return e ? e->solve_status : SDVGN_E_ARG; }
int sdvgn_ef_set_arith(sdvgn_ef* e, int mode) {
    if (!e || mode < 0 || mode > 1) return SDVGN_E_ARG;
    e->arith = mode;
    return SDVGN_OK;
}

int sdvgn_debug_solve_stamps(sdvgn_ef* e, unsigned long long* out16) {
    if (!e || !out16 || !e->solve_stamps) return 0;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return SDVGN_E_NODEVICE;
    std::memcpy(out16, e->solve_stamps, sizeof(unsigned long long) * 16);
    return 16;
}

int sdvgn_debug_phase_report(int per) {
    if (!g_pt.on || per < 1) return 0;
    fprintf(stderr, "[sdvgn profile] %d iterations\n", per);
    for (int k = 0; k < PT_N; ++k) if (g_pt.cnt[k]) fprintf(stderr, "  %-24s %8.1f us/iter  (%ld calls, %.1f us each)\n", kPtNames[k], g_pt.acc[k] / per, g_pt.cnt[k], g_pt.acc[k] / g_pt.cnt[k]);
    for (int k = 0; k < PT_N; ++k) { g_pt.acc[k] = 0; g_pt.cnt[k] = 0; }
    return 1;
}

int sdvgn_ef_get_linearize_times(sdvgn_ef* e, float* ms, int cap) {
    if (!e) return SDVGN_E_ARG;
    const int n = (int)e->lin_ms.size();
    if (ms) for (int i = 0; i < n && i < cap; ++i) ms[i] = e->lin_ms[i];
    return n;
}

int sdvgn_debug_launch_linearize(sdvgn_ef* e, int reps) {   // k_ef_linearize alone, `reps` launches back to back (no statistics, no threshold select)
    if (!e || e->host_only || !e->havePrecalc || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    for (int i = 0; i < reps; ++i) ef_launch_linearize(e);
    HIPCHK(hipGetLastError());
    return SDVGN_OK;
}

// Diagnostics: k_ef_linearize in different company, `reps` times each: 0 alone; 1 behind accumulate + reduce; 2 behind stitch + tail +
// resubstitute (no step: the state is not touched); 3 behind a one-wave kernel that just waits `spin_us` (the chip idles like it does
// during the small solve); 4 behind all of 1 and 2.  For kernel traces (tools/exp_linearize_context.py).
__global__ void k_debug_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int sdvgn_debug_launch_pattern(sdvgn_ef* e, int pattern, int reps, int spin_us) {
    if (!e || e->host_only || !e->havePrecalc || e->nP < 1 || !e->haveAdjoints) return SDVGN_E_STATE;
    EF_DEVICE(e);
    int rc;
    if ((rc = ef_accumulate(e, /*with_reduce=*/true))) return rc;
    for (int i = 0; i < reps; ++i) {
        if (pattern == 1 || pattern == 4) { if ((rc = ef_accumulate(e, true))) return rc; }
        if (pattern == 2 || pattern == 4) { if ((rc = ef_launch_solve(e, 0, 0.1, /*do_step=*/false, -1.0f, /*reuse=*/false, /*accumulated=*/true))) return rc; }
        if (pattern == 3) k_debug_spin<<<1, 64, 0, e->stream>>>((long long)spin_us * 100);
        ef_launch_linearize(e);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return SDVGN_OK;
}

int sdvgn_debug_loop_counters(sdvgn_ef* e, int* out4) {
    if (!e || !out4) return SDVGN_E_ARG;
    out4[0] = e->n_spec_launched; out4[1] = e->n_spec_used; out4[2] = e->n_merged_tests; out4[3] = e->n_pre_acc;
    return SDVGN_OK;
}
int sdvgn_ef_get_accepted_steps(sdvgn_ef* e) { return e ? e->n_accepted : SDVGN_E_ARG; }
int sdvgn_ef_get_look_ahead(sdvgn_ef* e, int* launched, int* used) {
    if (!e) return SDVGN_E_ARG;
    if (launched) *launched = e->n_spec_launched;
    if (used) *used = e->n_spec_used;
    return SDVGN_OK;
}

int sdvgn_ef_get_iteration_times(sdvgn_ef* e, double* us, int cap) {
    if (!e) return SDVGN_E_ARG;
    const int n = (int)e->iter_us.size();
    if (us) for (int i = 0; i < n && i < cap; ++i) us[i] = e->iter_us[i];
    return n;
}

int sdvgn_ef_get_state(sdvgn_ef* e, double* value_scaled4, double* state10, float* idepth) {
    if (!e) return SDVGN_E_ARG;
    EF_DEVICE(e);
    if (value_scaled4) for (int i = 0; i < 4; ++i) value_scaled4[i] = e->value_scaled[i];
    if (state10) for (int h = 0; h < e->nF; ++h) for (int i = 0; i < 10; ++i) state10[10 * h + i] = e->frames[h].state[i];
    if (idepth) {
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipMemcpy(idepth, e->pid, sizeof(float) * e->nP, hipMemcpyDeviceToHost));
    }
    return SDVGN_OK;
}

// ------------------------------------- read-back hooks ------------------------------------------------------
int sdvgn_ef_dim(sdvgn_ef* e) { return e ? CPARS + 6 * e->nF : SDVGN_E_ARG; }

int sdvgn_ef_get_system(sdvgn_ef* e, double* HA, double* bA, double* Hsc, double* bsc, double* HFinal, double* bFinal) {
    if (!e || (e->HFinal.empty() && !e->sys_on_device)) return SDVGN_E_STATE;
    const size_t n = CPARS + 6 * e->nF;
    if (e->sys_on_device && !e->sys_fetched) {   // the last solve ran on the device: fetch its system and sum the per-host shares once
        EF_DEVICE(e);
        std::vector<SolveSys> tmp(1);
        std::vector<SolvePieces> pc(e->nF);
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipMemcpy(tmp.data(), e->sys_dev, sizeof(SolveSys), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pc.data(), e->pieces_dev, sizeof(SolvePieces) * e->nF, hipMemcpyDeviceToHost));
        e->HA.assign(n * n, 0); e->bA.assign(n, 0); e->Hsc.assign(n * n, 0); e->bsc.assign(n, 0);
        e->HFinal.assign(n * n, 0); e->bFinal.assign(n, 0);
        auto tri = [](size_t i, size_t j) { return i * (i + 1) / 2 + j; };    // packed lower triangle, the rhs as row n
        for (size_t i = 0; i < n; ++i) {
            for (size_t j = 0; j <= i; ++j) {
                double ha = 0, hs = 0;
                for (int h = 0; h < e->nF; ++h) { ha += pc[h].CA[tri(i, j)]; hs += pc[h].CS[tri(i, j)]; }
                if (i == j) ha += i < CPARS ? e->cPrior[i] : e->frames[(i - CPARS) / 6].prior[(i - CPARS) % 6];
                e->HA[i * n + j] = e->HA[j * n + i] = ha;
                e->Hsc[i * n + j] = e->Hsc[j * n + i] = hs;
                e->HFinal[i * n + j] = e->HFinal[j * n + i] = tmp[0].tri[tri(i, j)];
            }
            double ba = 0, bs = 0;
            for (int h = 0; h < e->nF; ++h) { ba += pc[h].CA[tri(n, i)]; bs += pc[h].CS[tri(n, i)]; }
            e->bA[i] = ba + tmp[0].bprior[i];
            e->bsc[i] = bs;
            e->bFinal[i] = tmp[0].tri[tri(n, i)];
        }
        e->sys_fetched = true;
    }
    if (HA) std::memcpy(HA, e->HA.data(), 8 * n * n);
    if (bA) std::memcpy(bA, e->bA.data(), 8 * n);
    if (Hsc) std::memcpy(Hsc, e->Hsc.data(), 8 * n * n);
    if (bsc) std::memcpy(bsc, e->bsc.data(), 8 * n);
    if (HFinal) std::memcpy(HFinal, e->HFinal.data(), 8 * n * n);
    if (bFinal) std::memcpy(bFinal, e->bFinal.data(), 8 * n);
    return SDVGN_OK;
}

int sdvgn_ef_get_residual_J(sdvgn_ef* e, int which, float* out24) {
    if (!e || !out24 || e->nR < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    const size_t slots = (size_t)e->nF * e->nP;
    std::vector<float> J(2 * (size_t)kJPlanes * slots);
    std::vector<uint8_t> fl(slots);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(J.data(), e->J, sizeof(float) * J.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fl.data(), e->rflags, slots, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nR; ++i) {
        const size_t s = e->r_slot[i];
        const int sel = (fl[s] & RF_SEL) ? 1 : 0;
        const int buf = which ? sel : 1 - sel;
        const float* src = J.data() + (size_t)buf * kJPlanes * slots + s;
        for (int k = 0; k < kJPlanes; ++k) out24[(size_t)i * 24 + k] = src[(size_t)k * slots];
    }
    return SDVGN_OK;
}

int sdvgn_ef_get_residual_state(sdvgn_ef* e, int* state_state, int* state_new, float* energy_new, float* energy_wo, unsigned char* isActive) {
    if (!e || e->nR < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    const size_t slots = (size_t)e->nF * e->nP;
    std::vector<int8_t> st(slots), sn(slots);
    std::vector<float> en(slots), ew(slots);
    std::vector<uint8_t> fl(slots);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(st.data(), e->rstate, slots, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sn.data(), e->A.rstate_new, slots, hipMemcpyDeviceToHost));      // the current set (see new_cur)
    HIPCHK(hipMemcpy(en.data(), e->A.renergy_new, 4 * slots, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ew.data(), e->A.renergy_wo, 4 * slots, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fl.data(), e->rflags, slots, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nR; ++i) {
        const size_t s = e->r_slot[i];
        if (state_state) state_state[i] = st[s];
        if (state_new) state_new[i] = sn[s] & RS_MASK;   // (bit RS_WJLOW rides in the same plane)
        if (energy_new) energy_new[i] = en[s];
        if (energy_wo) energy_wo[i] = ew[s];
        if (isActive) isActive[i] = (fl[s] & RF_ACTIVE) ? 1 : 0;
    }
    return SDVGN_OK;
}

int sdvgn_ef_clear_error(sdvgn_ef* e) {
    if (!e) return SDVGN_E_ARG;
    if (!e->stats_host) return 0;
    if (!e->host_only) { EF_DEVICE(e); HIPCHK(hipStreamSynchronize(e->stream)); }      // nothing in flight may still raise it
    volatile unsigned* w = reinterpret_cast<volatile unsigned*>(e->stats_host + 6);
    const unsigned was = *w;
    *w = 0;
    e->sys_valid = false;              // (whatever system the failed call left is not re-used)
    e->pend_sel_valid = e->pend_rc_valid = e->pend_stats_valid = false;
    return (int)was;
}

int sdvgn_ef_get_point_nogood(sdvgn_ef* e, unsigned char* out) {
    if (!e || !out || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->pnogood, (size_t)e->nP, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nP; ++i) out[i] = (e->nogood_epoch != 0 && out[i] == e->nogood_epoch) ? 1 : 0;
    return e->nP;
}

int sdvgn_ef_get_points(sdvgn_ef* e, float* out9) {
    if (!e || !out9 || e->nP < 1) return SDVGN_E_STATE;
    EF_DEVICE(e);
    const size_t nP = e->nP;
    std::vector<float> a(nP), b(nP), c(4 * nP), d(nP), f(nP), g(nP);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(a.data(), e->pHddA, 4 * nP, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), e->pbdA, 4 * nP, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c.data(), e->pHcdA, 16 * nP, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(d.data(), e->pHdi, 4 * nP, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(f.data(), e->pbdSum, 4 * nP, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(g.data(), e->pstep, 4 * nP, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nP; ++i) {
        float* o = out9 + 9 * i;
        o[0] = a[i]; o[1] = b[i];
        for (int k = 0; k < 4; ++k) o[2 + k] = c[(size_t)k * nP + i];
        o[6] = d[i]; o[7] = f[i]; o[8] = g[i];
    }
    return SDVGN_OK;
}

// FullSystem::optimizeImmaturePoint (FullSystemOptPoint.cpp:18-185) for n immature points against the window currently loaded
// (frames, images, calibration and frame states as set by set_frames / set_frame_states).
int sdvgn_ef_optimize_immature(sdvgn_ef* e, int n, const int* host, const float* u, const float* v, const float* idepth_min,
                               const float* idepth_max, const float* energyTH, const float* color8, const float* weights8,
                               const unsigned char* isFromSensor, int minObs, int* result, float* idepth, int* res_state) {
    if (!e || e->host_only || e->nF < 2 || n < 0) return e && !e->host_only && n >= 0 ? SDVGN_E_STATE : SDVGN_E_ARG;
    if (n == 0) return SDVGN_OK;
    if (!host || !u || !v || !idepth_min || !idepth_max || !energyTH || !color8 || !weights8 || !isFromSensor || !result || !idepth || !res_state)
        return SDVGN_E_ARG;
    const int nF = e->nF;
    for (int i = 0; i < n; ++i) if (host[i] < 0 || host[i] >= nF) return SDVGN_E_ARG;
    EF_DEVICE(e);
    // FrameFramePrecalc::set (HessianBlocks.cpp:169-195), the three fields linearizeResidual reads
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t) {
            ImmPrecalc& P = e->imm_pc_host[h * nF + t];
            const FrameH& hf = e->frames[h];
            const FrameH& tf = e->frames[t];
            const gn::Pose l = gn::compose(tf.PRE_worldToCam, hf.PRE_camToWorld);
            double R[9];
            gn::rotation_matrix(l.q, R);
            for (int k = 0; k < 9; ++k) P.R[k] = (float)R[k];
            for (int k = 0; k < 3; ++k) P.t[k] = (float)l.t[k];
            double ab[2];
            gn::aff_from_to(hf.ab_exposure, tf.ab_exposure, hf.state_scaled[6], hf.state_scaled[7], tf.state_scaled[6], tf.state_scaled[7], ab);
            P.aff[0] = (float)ab[0]; P.aff[1] = (float)ab[1];
        }
    HIPCHK(hipMemcpyAsync(e->imm_pc_dev, e->imm_pc_host, sizeof(ImmPrecalc) * nF * nF, hipMemcpyHostToDevice, e->stream));
    // candidates in / results out through pinned memory the kernel accesses directly
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    const size_t bytes = np * (4 * 6 + 32 + 32 + 4 + 4 + 4 + 4 * (size_t)nF);
    if (bytes > e->imm_stage_bytes) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->imm_stage) SDVGN_HFREE(e->imm_stage);
        e->imm_stage = nullptr; e->imm_stage_bytes = 0;
        HIPCHK(SDVGN_HMALLOC(&e->imm_stage, bytes * 2));
        e->imm_stage_bytes = bytes * 2;
    }
    float* base = (float*)e->imm_stage;
    int* s_host = (int*)base;
    float* s_u = base + np, *s_v = base + 2 * np, *s_min = base + 3 * np, *s_max = base + 4 * np, *s_eth = base + 5 * np;
    float* s_col = base + 6 * np, *s_wts = base + 14 * np;
    int* s_res = (int*)(base + 22 * np);
    float* s_id = base + 23 * np;
    int* s_rs = (int*)(base + 24 * np);
    unsigned char* s_sens = (unsigned char*)(base + 24 * np + (size_t)nF * np);
    std::memcpy(s_host, host, 4 * (size_t)n); std::memcpy(s_u, u, 4 * (size_t)n); std::memcpy(s_v, v, 4 * (size_t)n);
    std::memcpy(s_min, idepth_min, 4 * (size_t)n); std::memcpy(s_max, idepth_max, 4 * (size_t)n); std::memcpy(s_eth, energyTH, 4 * (size_t)n);
    std::memcpy(s_col, color8, 32 * (size_t)n); std::memcpy(s_wts, weights8, 32 * (size_t)n); std::memcpy(s_sens, isFromSensor, (size_t)n);
    k_ef_optimize_immature<<<(n + 63) / 64, 64, 0, e->stream>>>(e->C, e->images, e->A.img_slots, e->imm_pc_dev, n, minObs, s_host, s_u, s_v, s_min, s_max, s_eth,
                                                              (const float4*)s_col, (const float4*)s_wts, s_sens, s_res, s_id, s_rs);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    std::memcpy(result, s_res, 4 * (size_t)n); std::memcpy(idepth, s_id, 4 * (size_t)n); std::memcpy(res_state, s_rs, 4 * (size_t)n * nF);
    return SDVGN_OK;
}

int sdvgn_ef_get_top_acc(sdvgn_ef* e, double* out, int* resInA) {
    if (!e) return SDVGN_E_STATE;
    if (!out) { if (resInA) *resInA = e->resInA; return SDVGN_OK; }   // (only EnergyFunctional::resInA of the last solve)
    EF_DEVICE(e);
    const int nF = e->nF;
    std::vector<double> g((size_t)nF * nF * kTopE);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->acc_in_host) std::memcpy(g.data(), e->acc_host, 8 * g.size());
    else HIPCHK(hipMemcpy(g.data(), e->acc_dev, 8 * g.size(), hipMemcpyDeviceToHost));
    for (int h = 0; h < nF; ++h)
        for (int t = 0; t < nF; ++t)
            for (int r = 0; r < 11; ++r)
                for (int c = 0; c < 11; ++c) out[(size_t)(h + nF * t) * 121 + r * 11 + c] = g[(size_t)(h * nF + t) * kTopE + r * 11 + c];
    if (resInA) *resInA = e->resInA;
    return SDVGN_OK;
}

}  // extern "C"
