// reproject.hip -- SURVEY.md section 8f row 2: the per-candidate work of class Reprojector (src/FullSystem/Reprojector.h:17-112)
//   Reprojector::reprojectPoint            src/FullSystem/Reprojector.cpp:602-616
//   pointQualityComparator's key           src/FullSystem/Reprojector.cpp:186-194
//   Reprojector::findMatchDirect           src/FullSystem/Reprojector.cpp:236-291
//   getWarpMatrixAffine / getBestSearchLevel / warpAffine / createPatchFromPatchWithBorder   :14-79, :338-347
//   align1D / align2D                      :349-447 / :449-545
//
// The reference walks a grid of candidate lists and calls findMatchDirect lazily until a cell has one match (:196-234); the
// function is pure in (point, reference frame, new frame), so this kernel evaluates it for EVERY candidate of the frame in one
// launch (8 key-frames x 2000 points = 14-16 k candidates) and the host replays the reference's selection on the results
// (sdv-loam_amd/reproject_api.py).
//
// One lane = one candidate.  The lane runs the reference's loops in the reference's order (sequential float accumulation over the
// 8x8 patch), which makes success flags and sub-pixel positions bit-identical to the CPU restatement; the 10x10 uint8 reference
// patch of every lane is assembled in LDS ([100 bytes][64 lanes]) and then packed, with its integer gradients, into 64 registers
// (the alignment loops are fully unrolled; gradients are exact halves of integer differences).  Geometry is fp64 like the reference's Eigen::Vector3d / SE3 code.  Memory behaviour: ~400 scattered 4-B
// taps for the warp and <= 10 x 81 for the alignment per candidate, all inside a 5.6 MB image / pyramid level -> cache resident,
// latency-bound; one workgroup = one wave so that 220+ workgroups spread over the chip.
#include "../../include/sdvgn.h"
#include "gnmath.hpp"
#include "devmem.hpp"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return -(int)_e;        \
    } while (0)

namespace sdvgn {

constexpr int kRpMaxFrames = 16;

struct RpFrameDev {
    gn::Pose camToWorld, worldToCam, T_cur_ref;   // T_cur_ref = cur.camToWorld.inverse() * camToWorld  (:264)
    float affLL[2];                               // fromToVecExposure(frame -> cur).cast<float>()      (:253-255)
    const float* dI;                              // level-0 AoS {I,dx,dy}
};

struct RpConst {
    int levels, nframes;
    int w[SDVGN_MAX_LEVELS], h[SDVGN_MAX_LEVELS];
    double K[9], Kinv[9];
    gn::Pose cur_worldToCam;
    const float* cur[SDVGN_MAX_LEVELS];
    RpFrameDev fr[kRpMaxFrames];
};

__device__ __forceinline__ void mat3_vec(const double* M, double x, double y, double z, double* o) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (M[3 * r] * x + M[3 * r + 1] * y) + M[3 * r + 2] * z;
}
__device__ __forceinline__ void se3_point(const gn::Pose& T, const double* p, double* o) {
    double r[3];
    gn::quat_rotate(T.q, p, r);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = r[i] + T.t[i];
}
__device__ __forceinline__ void ref_to_pixel_cur(const RpConst& C, const gn::Pose& T, const double* p, double* px2) {   // :586-593
    double pc[3], pix[3];
    se3_point(T, p, pc);
    pc[0] = pc[0] / pc[2]; pc[1] = pc[1] / pc[2]; pc[2] = pc[2] / pc[2];
    mat3_vec(C.K, pc[0], pc[1], pc[2], pix);
    px2[0] = pix[0]; px2[1] = pix[1];
}
__device__ __forceinline__ float interp_I(const float* __restrict__ mat, float x, float y, int width) {   // getInterpolatedElement33()[0]
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    return ((dxdy * bp[3 + 3 * width] + (dy - dxdy) * bp[3 * width]) + (dx - dxdy) * bp[3]) + (1 - dx - dy + dxdy) * bp[0];
}

// The 9x9 intensity window an alignment iteration reads (8x8 pixels x 4 bilinear taps), fetched as ONE batch of 81 independent
// loads: the kernel is bound by memory round trips (one wave per SIMD at most), so the loads of an iteration must be in flight
// together instead of row by row.
__device__ __forceinline__ void load_window(const float* __restrict__ cur, int wl, int u_r, int v_r, float (&win)[9][9]) {
    const float* base = cur + 3 * ((size_t)(v_r - 4) * wl + (u_r - 4));
#pragma unroll
    for (int yy = 0; yy < 9; ++yy)
#pragma unroll
        for (int xx = 0; xx < 9; ++xx) win[yy][xx] = base[3 * ((size_t)yy * wl + xx)];
    __builtin_amdgcn_sched_barrier(0);
}

// grid = ceil(n / 64), block = 64.  Outputs per candidate: px0 (projection into the new frame), cell (-1: outside the 8-px border),
// quality (|grad| at the host pixel), success, px (aligned position, valid when success), level.
__global__ void __launch_bounds__(64) k_reproject(const RpConst* __restrict__ Cp, int n, const float* __restrict__ u, const float* __restrict__ v,
                                                  const float* __restrict__ idepth, const int* __restrict__ host_idx,
                                                  const int* __restrict__ ref_idx, const int* __restrict__ ptype, double* __restrict__ px0_out,
                                                  int* __restrict__ cell_out, float* __restrict__ quality_out, int* __restrict__ success_out,
                                                  double* __restrict__ px_out, int* __restrict__ level_out) {
    __shared__ unsigned char s_pwb[100][64];
    const RpConst& C = *Cp;
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    if (i >= n) return;
    const float pu = u[i], pv = v[i], pid = idepth[i];
    const RpFrameDev& host = C.fr[host_idx[i]];
    const RpFrameDev& ref = C.fr[ref_idx[i]];
    const int w0 = C.w[0], h0 = C.h[0];

    // ---- reprojectPoint (:602-616): pixelFrame2PointWorld, pointWorld2PixelFrame, isInFrame(.., 8), cell ----
    double ptWorld[3];
    {
        double Ki[3];
        mat3_vec(C.Kinv, (double)pu, (double)pv, 1.0, Ki);
        const double s = (double)(1 / pid);                    // `1/point->idepth` is a float division
        const double ptRef[3] = {Ki[0] * s, Ki[1] * s, Ki[2] * s};
        se3_point(host.camToWorld, ptRef, ptWorld);
    }
    double px_cur[2];
    {
        double pc[3], pix[3];
        se3_point(C.cur_worldToCam, ptWorld, pc);
        pc[0] = pc[0] / pc[2]; pc[1] = pc[1] / pc[2]; pc[2] = pc[2] / pc[2];
        mat3_vec(C.K, pc[0], pc[1], pc[2], pix);
        px_cur[0] = pix[0]; px_cur[1] = pix[1];
    }
    px0_out[2 * i] = px_cur[0]; px0_out[2 * i + 1] = px_cur[1];
    int cell = -1;
    {
        const int x = (int)px_cur[0], y = (int)px_cur[1];
        if (x >= 8 && x < w0 - 8 && y >= 8 && y < h0 - 8) {
            const int n_cols = (w0 + 24) / 25;                 // ceil(w / cell_size), cell_size = 25 (:98-100)
            cell = (int)(px_cur[1] / 25) * n_cols + (int)(px_cur[0] / 25);
        }
    }
    cell_out[i] = cell;
    {
        const float* d = host.dI + 3 * (size_t)(int)(pv * w0 + pu);   // (int)(pt->v * wG[0] + pt->u)  (:188)
        quality_out[i] = sqrtf(d[1] * d[1] + d[2] * d[2]);
    }
    int success = 0, lvl = -1;
    double pxs[2] = {px_cur[0], px_cur[1]};
    do {
        if (cell < 0) break;                                   // never reaches a grid cell, findMatchDirect is not called
        // ---- findMatchDirect (:236-291) ----
        double ptRef[3], px[2];
        se3_point(ref.worldToCam, ptWorld, ptRef);             // pointWorld2PointFrame
        {
            double pc[3] = {ptRef[0] / ptRef[2], ptRef[1] / ptRef[2], ptRef[2] / ptRef[2]}, pix[3];
            mat3_vec(C.K, pc[0], pc[1], pc[2], pix);
            px[0] = pix[0]; px[1] = pix[1];
        }
        {
            const int x = (int)px[0], y = (int)px[1];
            if (!(x >= 6 && x < w0 - 6 && y >= 6 && y < h0 - 6)) break;   // isInFrame(px.cast<int>(), halfpatch_size_+2)
        }
        // getWarpMatrixAffine (:14-36)
        double A[4];
        {
            double xdu[3], xdv[3];
            mat3_vec(C.Kinv, px[0] + 5, px[1] + 0, 1.0, xdu);
            mat3_vec(C.Kinv, px[0] + 0, px[1] + 5, 1.0, xdv);
            const double su = ptRef[2] / xdu[2], sv = ptRef[2] / xdv[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) { xdu[k] *= su; xdv[k] *= sv; }
            double pc[2], pdu[2], pdv[2];
            ref_to_pixel_cur(C, ref.T_cur_ref, ptRef, pc);
            ref_to_pixel_cur(C, ref.T_cur_ref, xdu, pdu);
            ref_to_pixel_cur(C, ref.T_cur_ref, xdv, pdv);
            A[0] = (pdu[0] - pc[0]) / 5; A[2] = (pdu[1] - pc[1]) / 5;
            A[1] = (pdv[0] - pc[0]) / 5; A[3] = (pdv[1] - pc[1]) / 5;
        }
        const double det = A[0] * A[3] - A[1] * A[2];
        lvl = 0;
        {   // getBestSearchLevel (:38-51)
            double D = det;
            while (D > 3.0 && lvl < C.levels - 1) { lvl += 1; D *= 0.25; }
        }
        // warpAffine (:53-79), halfpatch_size_+1 = 5 -> the 10x10 patch with border, bytes in LDS
        {
            const double invdet = 1.0 / det;
            const float a00 = (float)(A[3] * invdet), a01 = (float)(-A[1] * invdet), a10 = (float)(-A[2] * invdet), a11 = (float)(A[0] * invdet);
            if (isnan(a00)) break;   // the reference would align against the previous candidate's stale patch here; see oracle/orc_reproject.cpp
            const float prx = (float)px[0], pry = (float)px[1];
            const float scale = (float)(1 << lvl);
            const float* __restrict__ img = ref.dI;
            for (int y0 = 0; y0 < 10; y0 += 2) {   // two patch rows = 80 independent tap loads per memory round trip
                float tap[2][10][4], fdx[2][10], fdy[2][10];
                bool inside[2][10];
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int x = 0; x < 10; ++x) {
                        const float ppx = (float)(x - 5) * scale, ppy = (float)(y0 + yy - 5) * scale;
                        const float qx = (a00 * ppx + a01 * ppy) + prx;
                        const float qy = (a10 * ppx + a11 * ppy) + pry;
                        inside[yy][x] = !(qx < 0 || qy < 0 || qx >= w0 - 1 || qy >= h0 - 1);
                        const float sxq = inside[yy][x] ? qx : 0.0f, syq = inside[yy][x] ? qy : 0.0f;
                        const int ix = (int)sxq, iy = (int)syq;
                        fdx[yy][x] = sxq - ix; fdy[yy][x] = syq - iy;
                        const float* bp = img + 3 * (ix + iy * w0);
                        tap[yy][x][0] = bp[0]; tap[yy][x][1] = bp[3]; tap[yy][x][2] = bp[3 * w0]; tap[yy][x][3] = bp[3 + 3 * w0];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int x = 0; x < 10; ++x) {
                        // getInterpolatedElement33(...)[0] (globalFuncs.h:51-65), same operation order as interp_I
                        const float dx = fdx[yy][x], dy = fdy[yy][x], dxdy = dx * dy;
                        const float val = ((dxdy * tap[yy][x][3] + (dy - dxdy) * tap[yy][x][2]) + (dx - dxdy) * tap[yy][x][1]) + (1 - dx - dy + dxdy) * tap[yy][x][0];
                        s_pwb[(y0 + yy) * 10 + x][lane] = inside[yy][x] ? (unsigned char)(int)val : (unsigned char)0;
                    }
            }
        }
#define PWB(yy, xx) ((int)s_pwb[(yy) * 10 + (xx)][lane])   /* patch_with_border_[y][x]; patch_[y][x] = PWB(y+1, x+1) (:338-347) */
        // One packed word per patch pixel, built once: intensity (8 bits) and the two integer central differences (+256, 10 bits
        // each).  The alignment loops are fully unrolled, so these 64 words stay in registers and an iteration costs no LDS traffic
        // (5 byte reads per pixel and iteration made the first version of this kernel LDS-issue-bound).
        unsigned pk[64];
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int x = 0; x < 8; ++x)
                pk[y * 8 + x] = (unsigned)PWB(y + 1, x + 1) | ((unsigned)(PWB(y + 1, x + 2) - PWB(y + 1, x) + 256) << 8) |
                                ((unsigned)(PWB(y + 2, x + 1) - PWB(y, x + 1) + 256) << 18);
#define PK_I(k) ((int)(pk[k] & 255u))
#define PK_DX(k) ((int)((pk[k] >> 8) & 1023u) - 256)
#define PK_DY(k) ((int)((pk[k] >> 18) & 1023u) - 256)
        const int wl = C.w[lvl], hl = C.h[lvl];
        const float* __restrict__ cur = C.cur[lvl];
        const float aff0 = ref.affLL[0], aff1 = ref.affLL[1];
        float uu = (float)(px_cur[0] / (1 << lvl)), vv = (float)(px_cur[1] / (1 << lvl));
        pxs[0] = px_cur[0] / (1 << lvl); pxs[1] = px_cur[1] / (1 << lvl);
        const float min_update_squared = (float)(0.03 * 0.03);
        bool converged = false, nan_exit = false;
        float mean_diff = 0;
        if (ptype[i] == 1) {
            // EDGELET (:275-284) -> align1D (:349-447)
            float dir0, dir1;
            {
                const float* d = ref.dI + 3 * (size_t)(int)(px[0] + px[1] * w0);
                double g0 = (double)d[1], g1 = (double)d[2];
                { const double z = g0 * g0 + g1 * g1; if (z > 0) { const double nn = sqrt(z); g0 /= nn; g1 /= nn; } }   // Eigen 3.3 normalize()
                double d0 = A[0] * g0 + A[1] * g1, d1 = A[2] * g0 + A[3] * g1;
                { const double z = d0 * d0 + d1 * d1; if (z > 0) { const double nn = sqrt(z); d0 /= nn; d1 /= nn; } }
                dir0 = (float)d0; dir1 = (float)d1;
            }
            float H0 = 0, H1 = 0, H2 = 0, H3 = 0;
#pragma unroll
            for (int y = 0; y < 8; ++y)
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float J0 = (float)(0.5 * (double)(dir0 * (float)PK_DX(y * 8 + x) + dir1 * (float)PK_DY(y * 8 + x)));
                    H0 += J0 * J0; H1 += J0 * 1.0f; H2 += 1.0f * J0; H3 += 1.0f * 1.0f;
                }
            const float hdet = H0 * H3 - H1 * H2;
            const float hinv = 1.0f / hdet;
            const float i00 = H3 * hinv, i01 = -H1 * hinv, i10 = -H2 * hinv, i11 = H0 * hinv;
            for (int iter = 0; iter < 10; ++iter) {
                const int u_r = (int)floorf(uu), v_r = (int)floorf(vv);
                if (u_r < 4 || v_r < 4 || u_r >= wl - 4 || v_r >= hl - 4) break;
                if (isnan(uu) || isnan(vv)) { nan_exit = true; break; }
                const float sx = uu - u_r, sy = vv - v_r;
                const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy));
                const float wTR = (float)((double)sx * (1.0 - (double)sy));
                const float wBL = (float)((1.0 - (double)sx) * (double)sy);
                const float wBR = sx * sy;
                float Jr0 = 0, Jr1 = 0;
                float win[9][9];
                load_window(cur, wl, u_r, v_r, win);
#pragma unroll
                for (int y = 0; y < 8; ++y) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const float sp = ((wTL * win[y][x] + wTR * win[y][x + 1]) + wBL * win[y + 1][x]) + wBR * win[y + 1][x + 1];
                        const float res = (sp - (float)(aff0 * (float)PK_I(y * 8 + x) + aff1)) + mean_diff;
                        const float J0 = (float)(0.5 * (double)(dir0 * (float)PK_DX(y * 8 + x) + dir1 * (float)PK_DY(y * 8 + x)));
                        Jr0 -= res * J0;
                        Jr1 -= res;
                    }
                }
                const float up0 = i00 * Jr0 + i01 * Jr1, up1 = i10 * Jr0 + i11 * Jr1;
                uu += up0 * dir0;
                vv += up0 * dir1;
                mean_diff += up1;
                if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; break; }
            }
        } else {
            // CORNER -> align2D (:449-545)
            float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int y = 0; y < 8; ++y)
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float J[3] = {(float)(0.5 * (double)PK_DX(y * 8 + x)), (float)(0.5 * (double)PK_DY(y * 8 + x)), 1.0f};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) H[3 * r + c] += J[r] * J[c];
                }
            float Hi[9];
            gn::inverse3f(H, Hi);
            for (int iter = 0; iter < 10; ++iter) {
                const int u_r = (int)floorf(uu), v_r = (int)floorf(vv);
                if (u_r < 4 || v_r < 4 || u_r >= wl - 4 || v_r >= hl - 4) break;
                if (isnan(uu) || isnan(vv)) { nan_exit = true; break; }
                const float sx = uu - u_r, sy = vv - v_r;
                const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy));
                const float wTR = (float)((double)sx * (1.0 - (double)sy));
                const float wBL = (float)((1.0 - (double)sx) * (double)sy);
                const float wBR = sx * sy;
                float Jr0 = 0, Jr1 = 0, Jr2 = 0;
                float win[9][9];
                load_window(cur, wl, u_r, v_r, win);
#pragma unroll
                for (int y = 0; y < 8; ++y) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const float sp = ((wTL * win[y][x] + wTR * win[y][x + 1]) + wBL * win[y + 1][x]) + wBR * win[y + 1][x + 1];
                        const float res = (sp - (float)(aff0 * (float)PK_I(y * 8 + x) + aff1)) + mean_diff;
                        const float dxv = (float)(0.5 * (double)PK_DX(y * 8 + x));
                        const float dyv = (float)(0.5 * (double)PK_DY(y * 8 + x));
                        Jr0 -= res * dxv;
                        Jr1 -= res * dyv;
                        Jr2 -= res;
                    }
                }
                const float up0 = (Hi[0] * Jr0 + Hi[1] * Jr1) + Hi[2] * Jr2;
                const float up1 = (Hi[3] * Jr0 + Hi[4] * Jr1) + Hi[5] * Jr2;
                const float up2 = (Hi[6] * Jr0 + Hi[7] * Jr1) + Hi[8] * Jr2;
                uu += up0;
                vv += up1;
                mean_diff += up2;
                if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; break; }
            }
        }
#undef PWB
#undef PK_I
#undef PK_DX
#undef PK_DY
        if (!nan_exit) { pxs[0] = (double)uu; pxs[1] = (double)vv; }   // `cur_px_estimate << u, v` (not reached on the NaN return)
        pxs[0] = pxs[0] * (1 << lvl); pxs[1] = pxs[1] * (1 << lvl);   // px_cur = px_scaled * (1<<search_level_)
        success = (converged && !nan_exit) ? 1 : 0;
    } while (false);
    success_out[i] = success;
    level_out[i] = lvl;
    px_out[2 * i] = pxs[0]; px_out[2 * i + 1] = pxs[1];
}

}  // namespace sdvgn

using namespace sdvgn;

struct sdvgn_reproj {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int levels = 1, max_frames = 0, max_points = 0;
    int w[SDVGN_MAX_LEVELS], h[SDVGN_MAX_LEVELS];
    bool haveK = false, haveCur = false;
    RpConst* C_host = nullptr;   // pinned; uploaded before each run
    RpConst* C_dev = nullptr;
    gn::Pose frame_pose[kRpMaxFrames];
    float frame_exposure[kRpMaxFrames];
    double frame_a[kRpMaxFrames], frame_b[kRpMaxFrames];
    bool frame_set[kRpMaxFrames] = {};
    float* frame_img[kRpMaxFrames] = {};   // owned copies (nullptr when borrowed)
    gn::Pose cur_pose;
    float cur_exposure = 1.f;
    double cur_a = 0, cur_b = 0;
    float* cur_img[SDVGN_MAX_LEVELS] = {};  // owned copies (nullptr when borrowed)
    // candidate staging: pinned host (inputs read zero-copy, outputs written zero-copy by the kernel)
    void* stage = nullptr;
    size_t stage_bytes = 0;
};

extern "C" {

int sdvgn_reproj_create(sdvgn_reproj** out, int device, int w0, int h0, int levels, int max_frames, int max_points, void* stream) {
    if (!out || w0 < 32 || h0 < 32 || levels < 1 || levels > SDVGN_MAX_LEVELS || max_frames < 1 || max_frames > kRpMaxFrames || max_points < 1)
        return SDVGN_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) return SDVGN_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    sdvgn_reproj* r = new (std::nothrow) sdvgn_reproj();
    if (!r) return -(int)hipErrorOutOfMemory;
    r->device = device; r->levels = levels; r->max_frames = max_frames; r->max_points = max_points;
    for (int l = 0; l < levels; ++l) { r->w[l] = w0 >> l; r->h[l] = h0 >> l; }
    if (stream) r->stream = (hipStream_t)stream;
    else { HIPCHK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); r->own_stream = true; }
    HIPCHK(SDVGN_HMALLOC((void**)&r->C_host, sizeof(RpConst)));
    HIPCHK(SDVGN_DMALLOC((void**)&r->C_dev, sizeof(RpConst)));
    std::memset(r->C_host, 0, sizeof(RpConst));
    const size_t np = ((size_t)max_points + 63) & ~(size_t)63;
    r->stage_bytes = np * (4 * 6 + 16 + 4 + 4 + 4 + 16 + 4);   // u v idepth host ref type | px0 cell quality success px level
    HIPCHK(SDVGN_HMALLOC(&r->stage, r->stage_bytes));
    *out = r;
    return SDVGN_OK;
}

void sdvgn_reproj_destroy(sdvgn_reproj* r) {
    if (!r) return;
    hipSetDevice(r->device);
    hipStreamSynchronize(r->stream);
    for (int k = 0; k < kRpMaxFrames; ++k) if (r->frame_img[k]) SDVGN_DFREE(r->frame_img[k]);
    for (int l = 0; l < SDVGN_MAX_LEVELS; ++l) if (r->cur_img[l]) SDVGN_DFREE(r->cur_img[l]);
    SDVGN_HFREE(r->C_host); SDVGN_DFREE(r->C_dev); SDVGN_HFREE(r->stage);
    if (r->own_stream) hipStreamDestroy(r->stream);
    delete r;
}

void* sdvgn_reproj_stream(sdvgn_reproj* r) { return r ? (void*)r->stream : nullptr; }

int sdvgn_reproj_set_calib(sdvgn_reproj* r, float fx, float fy, float cx, float cy) {   // Reprojector::Reprojector (:81-87)
    if (!r) return SDVGN_E_ARG;
    RpConst& C = *r->C_host;
    const double K[9] = {(double)fx, 0, (double)cx, 0, (double)fy, (double)cy, 0, 0, 1};
    std::memcpy(C.K, K, sizeof(K));
    // Matrix3d::inverse(): cofactors of column 0, det = their dot product with column 0, cofactor(c,r) * (1/det)
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return K[i1 * 3 + j1] * K[i2 * 3 + j2] - K[i1 * 3 + j2] * K[i2 * 3 + j1];
    };
    const double det = (cof(0, 0) * K[0] + cof(1, 0) * K[3]) + cof(2, 0) * K[6];
    const double invdet = 1.0 / det;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C.Kinv[a * 3 + b] = cof(b, a) * invdet;
    r->haveK = true;
    return SDVGN_OK;
}

static int rp_set_image(sdvgn_reproj* r, float** owned, const float** slot, const float* host_aos3, const float* dev_aos3, size_t npix) {
    if (dev_aos3) {
        if (*owned) { HIPCHK(hipStreamSynchronize(r->stream)); SDVGN_DFREE(*owned); *owned = nullptr; }
        *slot = dev_aos3;
        return SDVGN_OK;
    }
    if (!host_aos3) return SDVGN_OK;   // keep the current image
    if (!*owned) HIPCHK(SDVGN_DMALLOC((void**)owned, sizeof(float) * 3 * npix));
    HIPCHK(hipMemcpyAsync(*owned, host_aos3, sizeof(float) * 3 * npix, hipMemcpyHostToDevice, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));   // host_aos3 may be pageable and reused by the caller
    *slot = *owned;
    return SDVGN_OK;
}

int sdvgn_reproj_set_frame(sdvgn_reproj* r, int idx, const double* camToWorld7, const float* dI_aos3, const float* dI_aos3_dev, float ab_exposure,
                           double aff_a, double aff_b) {
    if (!r || idx < 0 || idx >= r->max_frames || !camToWorld7) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(r->device));
    gn::pose_load(r->frame_pose[idx], camToWorld7);
    r->frame_exposure[idx] = ab_exposure; r->frame_a[idx] = aff_a; r->frame_b[idx] = aff_b;
    int rc = rp_set_image(r, &r->frame_img[idx], &r->C_host->fr[idx].dI, dI_aos3, dI_aos3_dev, (size_t)r->w[0] * r->h[0]);
    if (rc) return rc;
    if (!r->C_host->fr[idx].dI) return SDVGN_E_STATE;
    r->frame_set[idx] = true;
    return SDVGN_OK;
}

int sdvgn_reproj_set_cur(sdvgn_reproj* r, const double* camToWorld7, float ab_exposure, double aff_a, double aff_b) {
    if (!r || !camToWorld7) return SDVGN_E_ARG;
    gn::pose_load(r->cur_pose, camToWorld7);
    r->cur_exposure = ab_exposure; r->cur_a = aff_a; r->cur_b = aff_b;
    r->haveCur = true;
    return SDVGN_OK;
}

int sdvgn_reproj_set_cur_level(sdvgn_reproj* r, int lvl, const float* dIp_aos3, const float* dIp_aos3_dev) {
    if (!r || lvl < 0 || lvl >= r->levels || (!dIp_aos3 && !dIp_aos3_dev)) return SDVGN_E_ARG;
    HIPCHK(hipSetDevice(r->device));
    return rp_set_image(r, &r->cur_img[lvl], &r->C_host->cur[lvl], dIp_aos3, dIp_aos3_dev, (size_t)r->w[lvl] * r->h[lvl]);
}

int sdvgn_reproj_match(sdvgn_reproj* r, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const int* ref_idx,
                       const int* type, double* px0, int* cell, float* quality, int* success, double* px, int* level) {
    if (!r || n < 0 || n > r->max_points) return SDVGN_E_ARG;
    if (!r->haveK || !r->haveCur) return SDVGN_E_STATE;
    if (n == 0) return SDVGN_OK;
    if (!u || !v || !idepth || !host_idx || !ref_idx || !type || !px0 || !cell || !quality || !success || !px) return SDVGN_E_ARG;
    for (int l = 0; l < r->levels; ++l) if (!r->C_host->cur[l]) return SDVGN_E_STATE;
    int nframes = 0;
    for (int i = 0; i < n; ++i) {
        if (host_idx[i] < 0 || host_idx[i] >= r->max_frames || !r->frame_set[host_idx[i]]) return SDVGN_E_ARG;
        if (ref_idx[i] < 0 || ref_idx[i] >= r->max_frames || !r->frame_set[ref_idx[i]]) return SDVGN_E_ARG;
        nframes = std::max(nframes, std::max(host_idx[i], ref_idx[i]) + 1);
    }
    HIPCHK(hipSetDevice(r->device));
    {   // hipGetLastError() below must report THIS launch: drop an error some earlier, unrelated runtime call left behind
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("SDVGN_PROFILE")) fprintf(stderr, "[sdvgn] stale HIP error %d (%s) cleared in sdvgn_reproj_match\n", (int)stale, hipGetErrorString(stale));
    }
    RpConst& C = *r->C_host;
    C.levels = r->levels; C.nframes = nframes;
    for (int l = 0; l < r->levels; ++l) { C.w[l] = r->w[l]; C.h[l] = r->h[l]; }
    const gn::Pose cur_w2c = gn::inverse(r->cur_pose);
    C.cur_worldToCam = cur_w2c;
    for (int k = 0; k < nframes; ++k) {
        if (!r->frame_set[k]) continue;
        RpFrameDev& f = C.fr[k];
        f.camToWorld = r->frame_pose[k];
        f.worldToCam = gn::inverse(r->frame_pose[k]);
        f.T_cur_ref = gn::compose(cur_w2c, r->frame_pose[k]);
        double ab[2];
        gn::aff_from_to(r->frame_exposure[k], r->cur_exposure, r->frame_a[k], r->frame_b[k], r->cur_a, r->cur_b, ab);
        f.affLL[0] = (float)ab[0]; f.affLL[1] = (float)ab[1];
    }
    HIPCHK(hipMemcpyAsync(r->C_dev, r->C_host, sizeof(RpConst), hipMemcpyHostToDevice, r->stream));
    // candidates in, results out: pinned host memory accessed by the kernel directly (coalesced, each element once)
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    char* base = (char*)r->stage;
    float* su = (float*)base, *sv = su + np, *sid = sv + np;
    int* sh = (int*)(sid + np), *sr = sh + np, *st = sr + np;
    double* spx0 = (double*)(st + np);
    int* scell = (int*)(spx0 + 2 * np);
    float* sq = (float*)(scell + np);
    int* ssucc = (int*)(sq + np);
    double* spx = (double*)(ssucc + np);
    int* slvl = (int*)(spx + 2 * np);
    std::memcpy(su, u, 4 * (size_t)n); std::memcpy(sv, v, 4 * (size_t)n); std::memcpy(sid, idepth, 4 * (size_t)n);
    std::memcpy(sh, host_idx, 4 * (size_t)n); std::memcpy(sr, ref_idx, 4 * (size_t)n); std::memcpy(st, type, 4 * (size_t)n);
    k_reproject<<<(n + 63) / 64, 64, 0, r->stream>>>(r->C_dev, n, su, sv, sid, sh, sr, st, spx0, scell, sq, ssucc, spx, slvl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(r->stream));
    std::memcpy(px0, spx0, 16 * (size_t)n); std::memcpy(cell, scell, 4 * (size_t)n); std::memcpy(quality, sq, 4 * (size_t)n);
    std::memcpy(success, ssucc, 4 * (size_t)n); std::memcpy(px, spx, 16 * (size_t)n);
    if (level) std::memcpy(level, slvl, 4 * (size_t)n);
    return SDVGN_OK;
}

}  // extern "C"
