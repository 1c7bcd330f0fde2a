/* tools/kf_host_loop.c -- bench infrastructure: the key-frame cycle of tools/exp_keyframe_update.py driven from a C host loop through
 * include/sdvgn.h alone (the consumer of the ABI is a C++ host loop; a Python loop adds ~10 us per call to what it measures).
 * Built by __graft_entry__.build() into tools/libkfloop.so (gcc, links libsdvgn.so).  Not part of the product.
 *
 * The world: F frames (F = window + 1), P points hosted by each, everything the host loop would hold in its own objects handed over as flat
 * arrays once.  A step = one FullSystem::makeKeyFrame's worth of graph edits + the optimisation it triggers (see exp_keyframe_update.py):
 *   removePoint x P (the oldest frame's points), marginalizeFrame, insertFrame (raw image), insertPoint x P, insertResidual x 2 (F-2) P,
 *   makeIDX, setAdjointsF, setPrecalcValues, optimize(its, exactly), optimize's tail (linearizeAll(true)). */
#define _POSIX_C_SOURCE 199309L
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sdvgn.h"

typedef struct {
    int F, P, w, h;
    const double* evalPT7;      /* [F][7]  */
    const double* state10;      /* [F][10] */
    const double* state_zero10; /* [F][10] */
    const int* frameID;         /* [F] */
    const float* frameTH;       /* [F] */
    const float* const* image;  /* [F] raw images (pinned host memory) */
    const float* u; const float* v; const float* idepth; const float* idepth_zero;      /* [F][P] */
    const float* color8; const float* weights8;                                        /* [F][P][8] */
    const unsigned char* prior; const unsigned char* sensor;                            /* [F][P] */
    const double* HM1; const double* bM1;   /* the prior that is left when the oldest frame goes: (4+6(F-2))^2, 4+6(F-2) */
} kf_world;

typedef struct {               /* per step, in the order the step sends its residuals: the (F-2) P surviving points -> the new frame, then the new points -> every surviving frame */
    const int* r_target; const unsigned char* r_hasMatcher; const double* r_matcher;
} kf_step;

typedef struct {
    int win[16];               /* world frame at every window index */
    int* ids;                  /* [F][P] library point ids of the points of every world frame that is in the window */
} kf_state;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* state: win[] = the window's world frames (oldest first), ids as the library knows them.  new_frame[s] = the world frame step s inserts.
 * seconds[s] = wall time of step s.  Returns 0 or the first error of an sdvgn call. */
int kf_host_loop(sdvgn_ef* ef, const kf_world* W, kf_state* S, int n_steps, const int* new_frame, const kf_step* steps, int its, double* seconds) {
    const int nW = W->F - 1, P = W->P;
    int* host_k = (int*)malloc(sizeof(int) * P);
    int* pid = (int*)malloc(sizeof(int) * 2 * (size_t)(nW - 1) * P);
    int* zero_state = (int*)calloc(2 * (size_t)(nW - 1) * P, sizeof(int));
    float* relbs = (float*)malloc(sizeof(float) * (size_t)nW * P);
    int* ngood = (int*)malloc(sizeof(int) * (size_t)nW * P);
    unsigned char* removed = (unsigned char*)malloc((size_t)nW * nW * P);
    int rc = 0;
    for (int s = 0; s < n_steps && !rc; ++s) {
        const double t0 = now_s();
        const int old = S->win[0], nw = new_frame[s], k = nW - 1;
        if ((rc = sdvgn_ef_remove_points(ef, P, S->ids + (size_t)old * P)) < 0) break;
        if ((rc = sdvgn_ef_remove_frame(ef, 0, W->HM1, W->bM1)) < 0) break;
        memmove(S->win, S->win + 1, sizeof(int) * (nW - 1));
        S->win[k] = nw;
        if ((rc = sdvgn_ef_insert_frame(ef, W->evalPT7 + 7 * nw, W->state10 + 10 * nw, W->state_zero10 + 10 * nw, W->frameID[nw], 1.0f, W->frameTH[nw], 0, W->image[nw])) < 0) break;
        for (int i = 0; i < P; ++i) host_k[i] = k;
        const size_t o = (size_t)nw * P;
        if ((rc = sdvgn_ef_insert_points(ef, P, host_k, W->u + o, W->v + o, W->idepth + o, W->idepth_zero + o, W->color8 + 8 * o, W->weights8 + 8 * o, W->prior + o,
                                         W->sensor + o, S->ids + o)) < 0) break;
        for (int t = 0; t < k; ++t) memcpy(pid + (size_t)t * P, S->ids + (size_t)S->win[t] * P, sizeof(int) * P);
        for (int t = 0; t < k; ++t) memcpy(pid + (size_t)(k + t) * P, S->ids + o, sizeof(int) * P);
        if ((rc = sdvgn_ef_insert_residuals(ef, 2 * k * P, pid, steps[s].r_target, zero_state, steps[s].r_hasMatcher, steps[s].r_matcher)) < 0) break;
        if ((rc = sdvgn_ef_make_idx(ef)) < 0) break;
        if ((rc = sdvgn_ef_set_adjoints(ef)) < 0) break;
        if ((rc = sdvgn_ef_set_precalc(ef)) < 0) break;
        if ((rc = sdvgn_ef_optimize(ef, its, 1, 0, 0, 0)) < 0) break;
        double lastE;
        if ((rc = sdvgn_ef_optimize_finish(ef, &lastE, relbs, ngood, removed)) < 0) break;
        rc = 0;
        seconds[s] = now_s() - t0;
    }
    free(host_k); free(pid); free(zero_state); free(relbs); free(ngood); free(removed);
    return rc;
}
