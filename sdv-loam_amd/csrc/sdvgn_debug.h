/* sdvgn_debug.h -- PRIVATE diagnostics entry points of libsdvgn.so (profiling experiments under tools/, never part of the drop-in
 * boundary declared in include/sdvgn.h).  The symbols are exported so that the experiment scripts can reach them through ctypes. */
/* Environment switches the library reads (A/B measurements and tests; none is needed in normal use):
 *   SDVGN_DEBUG_FLAGS   bit 1 (2) / 2 (4) / 7 (128): experiments of the diagnostic k_ef_linearize instantiation only; bit 5 (32): its stage stamps; bit 6 (64): stamps of the
 *                       small solve; bit 8 (256): one residual group per k_ef_linearize workgroup; bit 9 (512): the accept test of a trial step as a launch of its own
 *                       (k_ef_stats_select) instead of a workgroup of the next body's accumulate (k_ef_acc_stats) -- read when a window is loaded
 *   SDVGN_FUSED_APPLY=0 applyRes as workgroups of the statistics launch (the loop of rounds 3-5) instead of fused into the linearise
 *   SDVGN_PROFILE=1     host wall time per phase of the optimize loop (sdvgn_debug_phase_report); SDVGN_OPT_TIMING=1: the pre-loop phases of every call
 *   (SDVGN_PMC_SAFE is read by tools/pmc_linearize.py, not by the library: it restricts the counter groups and runs the loop without the look-ahead solve) */
#pragma once
#include "../../include/sdvgn.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Diagnostics: k_ef_linearize `reps` times in different company (0 alone, 1 behind accumulate + reduce, 2 behind stitch + tail +
 * resubstitute without a step, 3 behind a one-wave kernel that waits spin_us, 4 behind 1 and 2), for kernel traces */
int sdvgn_debug_launch_pattern(sdvgn_ef* ef, int pattern, int reps, int spin_us);
/* Diagnostics: k_ef_linearize alone, `reps` launches back to back on the library's stream (no statistics, no threshold select) */
int sdvgn_debug_launch_linearize(sdvgn_ef* ef, int reps);
/* Diagnostics (SDVGN_PROFILE=1 in the environment when the library is loaded): prints the accumulated host wall time per phase
 * of the solve / optimize path to stderr, divided by `per`, and clears the counters.  Returns 1 if a report was printed, else 0. */
int sdvgn_debug_phase_report(int per);
/* SDVGN_DEBUG_FLAGS bit 6: wall_clock64() (100 MHz) stamps of the phases of the last device-side small solve; returns the word count (16) */
int sdvgn_debug_solve_stamps(sdvgn_ef* ef, unsigned long long* out16);
/* Diagnostics (SDVGN_DEBUG_FLAGS bit5 = 32 set when the handle is created): wall_clock64() stamps (10 ns) of k_ef_linearize's stages
 * from the last launch, [workgroup][wave][8] 64-bit words (tools/exp_linearize_stages.py).  Returns the number of words copied, 0 if the diagnostics are off. */
int sdvgn_debug_read_stamps(sdvgn_ef* ef, unsigned long long* out, int cap_words);
/* Which forms the LAST sdvgn_ef_optimize call's bodies took: out4 = { rejected cases solved ahead on the side stream, bodies that started from such a solution,
 * accept tests taken as a workgroup of the next body's accumulate launch (k_ef_acc_stats), accumulates queued ahead of the verdict } -- a test asserts that the
 * default loop really runs the fast forms (a silent fall-back to the older ones would only show as a slower bench) */
int sdvgn_debug_loop_counters(sdvgn_ef* ef, int* out4);
#ifdef __cplusplus
}
#endif
