"""Parity of the HIP coarse tracker (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances: per-point terms and integer counters bit-exact; E rel 1e-5 (float tree sum vs sequential float sum);
H, b rel 1e-5; pose increments rel 1e-4 (BASELINE.json north_star)."""
import ctypes as C

import numpy as np
import pytest

from common import load_problem, rel_err, small_problem, start_pose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(sdvgn_lib):
    from sdv_loam_amd import api as A
    return A


def pair(api, orc, P, max_points=None, **kw):
    G = load_problem(api.CoarseTracker(P.w, P.h, P.levels, max_points=max_points or 1 << 16, max_batch=40), P, **kw)
    O = load_problem(orc.OracleTracker(P.w, P.h, P.levels), P, **kw)
    return G, O


@pytest.mark.parametrize("shape", [(1241, 376, 4), (1408, 376, 4), (64, 48, 3), (37, 23, 2)])
def test_pyramid_bit_exact(api, orc, shape):
    from sdv_loam_amd import synthetic as syn
    w, h, levels = shape
    img = syn.make_image(w, h, seed=11)
    G = api.CoarseTracker(w, h, levels, max_points=16, max_batch=1)
    G.set_new_image(img)
    ref = orc.make_images(img, w, h, levels)
    for l in range(levels):
        g = G.get_pyr(l)
        assert np.array_equal(g[..., 0], ref[l][..., 0])
        assert np.array_equal(g[1:-1, :, 1:], ref[l][1:-1, :, 1:])
        assert np.all(g[0, :, 1:] == 0) and np.all(g[-1, :, 1:] == 0)


def test_makeK_identical(api, orc):
    from sdv_loam_amd import synthetic as syn
    G = api.CoarseTracker(1241, 376, 4, max_points=16, max_batch=1)
    O = orc.OracleTracker(1241, 376, 4)
    for T in (G, O):
        T.makeK(**syn.KITTI00)
    for l in range(4):
        kg, kig = G.get_K(l)
        ko, kio = O.get_K(l)
        assert np.array_equal(kg, ko) and np.array_equal(kig, kio)


def check_res_gs(G, O, lvl, pose, a, b, cutoff):
    rg = G.calcRes(lvl, pose, a, b, cutoff)
    ro = O.calcRes(lvl, pose, a, b, cutoff)
    Wg, status = G.warped(lvl)
    Wo = O.warped()
    # integer counters exact
    assert rg[1] == ro[1]
    assert Wg.shape == Wo.shape
    if np.isnan(ro[5]):
        assert np.isnan(rg[5])
    else:
        assert rg[5] == ro[5]
    # per-point terms bit-exact (same float32 operations in the same order, no FMA contraction)
    assert np.array_equal(Wg.view(np.uint32), Wo.view(np.uint32))
    if ro[1] > 0:
        assert rel_err(rg[0], ro[0]) < 1e-5
        assert np.allclose(rg[2:5], ro[2:5], rtol=1e-5, atol=1e-9)
    Hg, bg = G.calcGS(lvl, pose, a, b, cutoff)
    Ho, bo = O.calcGS(lvl, a, b)
    if Wo.shape[1] > 0:
        assert rel_err(Hg, Ho) < 1e-5
        assert rel_err(bg, bo) < 1e-5 or np.linalg.norm(bg - bo) < 1e-5 * np.sqrt(np.abs(np.diag(Ho)) @ np.ones(8))
    else:
        assert np.all(np.isnan(Hg)) == np.all(np.isnan(Ho))
    r2, H2, b2 = G.resAndGS(lvl, pose, a, b, cutoff)
    assert np.array_equal(r2, rg, equal_nan=True) and np.array_equal(H2, Hg, equal_nan=True) and np.array_equal(b2, bg, equal_nan=True)   # deterministic
    return rg, ro


@pytest.mark.parametrize("seed,n", [(0, 400), (1, 401), (2, 2000), (3, 6001)])
def test_res_and_gs_parity(api, orc, seed, n):
    P = small_problem(seed=seed, n=n, noise=2.0)
    G, O = pair(api, orc, P, ref_aff=(0.01, 1.0))
    pose = start_pose(orc, P, seed)
    for lvl in range(P.levels):
        check_res_gs(G, O, lvl, pose, 0.03, 2.0, 20.0)
        check_res_gs(G, O, lvl, P.gt_pose, 0.04, 2.5, 20.0)


def test_edge_cases(api, orc):
    P = small_problem(seed=4, n=300)
    G, O = pair(api, orc, P)
    far = np.array([0, 0, 0, 1, 1e4, 0, 0], float)
    check_res_gs(G, O, 0, far, 0.0, 0.0, 20.0)                   # nothing projects inside: 0 terms, NaN ratio / H
    check_res_gs(G, O, 0, P.gt_pose, 0.04, 102.5, 20.0)          # everything saturated
    check_res_gs(G, O, 0, P.gt_pose, 0.04, 102.5, 160.0)         # doubled cutoffs bring them back
    behind = orc.se3_exp(np.array([0, 0, -50.0, 0, 0, 0]))       # points behind the camera: new_idepth <= 0
    check_res_gs(G, O, 1, behind, 0.0, 0.0, 20.0)
    # empty reference set on one level
    for T in (G, O):
        T.set_ref(2, np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    rg = G.calcRes(2, P.gt_pose, 0.0, 0.0, 20.0)
    ro = O.calcRes(2, P.gt_pose, 0.0, 0.0, 20.0)
    assert rg[1] == ro[1] == 0 and rg[0] == ro[0] == 0


def test_nan_pixels_skipped(api, orc):
    P = small_problem(seed=5, n=500)
    img = P.image.copy()
    img[60:90, 100:160] = np.nan       # non-finite target pixels: `if(!std::isfinite(hitColor[0])) continue;`
    P.image = img
    G, O = pair(api, orc, P)
    rg, ro = check_res_gs(G, O, 0, P.gt_pose, 0.04, 2.5, 20.0)
    assert ro[1] < 500


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_track_parity_host_driven(api, orc, seed):
    P = small_problem(seed=seed, n=600, w=320, h=240, levels=3, noise=1.5)
    G, O = pair(api, orc, P)
    start = start_pose(orc, P, seed)
    okg, pg, ag, lrg, flg, trg = G.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
    oko, po, ao, lro, flo, tro = O.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
    assert okg == oko
    dg = orc.se3_log(orc.se3_mul(pg, orc.se3_inverse(start)))
    do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(start)))
    assert rel_err(dg, do) < 1e-4
    assert abs(ag[0] - ao[0]) < 1e-4 * max(1, abs(ao[0])) and abs(ag[1] - ao[1]) < 1e-4 * max(1, abs(ao[1]))
    assert np.allclose(lrg[:P.levels], lro[:P.levels], rtol=1e-4)
    assert np.allclose(flg, flo, rtol=1e-4)
    # same LM path: same number of trials, same accept/reject decisions, same increments
    assert len(trg) == len(tro)
    assert np.array_equal(trg[:, [0, 1, 3]], tro[:, [0, 1, 3]])
    big = np.abs(tro[:, 4:12]).max(axis=1) > 1e-6
    assert rel_err(trg[big, 4:12], tro[big, 4:12]) < 1e-3


@pytest.mark.parametrize("seed", [0, 1])
def test_track_batch_device_resident(api, orc, seed):
    P = small_problem(seed=seed, n=600, w=320, h=240, levels=3, noise=1.5)
    G, O = pair(api, orc, P)
    B = 5
    starts = np.stack([start_pose(orc, P, seed * 10 + i) for i in range(B)])
    okb, pb, ab, lrb, flb = G.trackBatch(starts, np.zeros((B, 2)), P.levels - 1)
    for i in range(B):
        oko, po, ao, lro, flo, _ = O.trackNewestCoarse(starts[i], (0.0, 0.0), P.levels - 1)
        assert bool(okb[i]) == oko
        dg = orc.se3_log(orc.se3_mul(pb[i], orc.se3_inverse(starts[i])))
        do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(starts[i])))
        assert rel_err(dg, do) < 1e-4, (i, dg, do)
        assert np.allclose(ab[i], ao, rtol=1e-4, atol=1e-4)
        assert np.allclose(lrb[i, :P.levels], lro[:P.levels], rtol=1e-4)


@pytest.mark.parametrize("team", [-1, 1, 2, 3, 8, 32])
def test_track_batch_team_sizes(api, orc, team):
    """k_track (one workgroup per hypothesis) and k_track_team with 1..32 workgroups per hypothesis: the oracle's result for every
    hypothesis, whatever the team size; consecutive calls on the same handle (the exchange counters are restored by the kernel)."""
    P = small_problem(seed=4, n=900, w=320, h=240, levels=3, noise=1.5)
    G, O = pair(api, orc, P)
    G.set_team(team)
    B = 11                                       # not a multiple of 8: the padded tail of the grid leaves at once
    starts = np.stack([start_pose(orc, P, 40 + i) for i in range(B)])
    ref = [O.trackNewestCoarse(starts[i], (0.0, 0.0), P.levels - 1) for i in range(B)]
    first = None
    for rep in range(3):
        okb, pb, ab, lrb, flb = G.trackBatch(starts, np.zeros((B, 2)), P.levels - 1)
        assert G.last_team() == (0 if team < 0 else team)
        for i in range(B):
            oko, po, ao, lro, flo, _ = ref[i]
            assert bool(okb[i]) == oko
            dg = orc.se3_log(orc.se3_mul(pb[i], orc.se3_inverse(starts[i])))
            do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(starts[i])))
            assert rel_err(dg, do) < 1e-4, (i, dg, do)
            assert np.allclose(ab[i], ao, rtol=1e-4, atol=1e-4)
            assert np.allclose(lrb[i, :P.levels], lro[:P.levels], rtol=1e-4)
            assert np.allclose(flb[i], flo, rtol=1e-4)
        if first is None:
            first = (pb.copy(), ab.copy(), lrb.copy())
        else:                                    # deterministic: fixed summation order inside and across the workgroups
            assert np.array_equal(pb, first[0]) and np.array_equal(ab, first[1]) and np.array_equal(lrb, first[2], equal_nan=True)


def test_track_batch_team_abort_paths(api, orc):
    """Level abort (lastRes > 1.5 minRes) and cutoff doubling + level repeat inside the team kernel: every workgroup of a team takes the
    same exit."""
    P = small_problem(seed=6, n=300, noise=8.0)
    G, O = pair(api, orc, P)
    G.set_team(4)
    start = start_pose(orc, P, 6)
    okb, pb, ab, lrb, _ = G.trackBatch(start[None], np.zeros((1, 2)), P.levels - 1, min_res=np.full((1, 5), 1e-3))
    oko, po, _, lro, _, _ = O.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1, min_res=[1e-3] * 5)
    assert G.last_team() == 4 and not okb[0] and not oko
    assert np.allclose(lrb[0], lro, rtol=1e-4, equal_nan=True)
    okb, pb, ab, lrb, _ = G.trackBatch(P.gt_pose[None], np.array([[0.04, 72.5]]), P.levels - 1)
    oko, po, ao, lro, _, tro = O.trackNewestCoarse(P.gt_pose, (0.04, 72.5), P.levels - 1)
    assert tro[:, 14].max() > 1 and bool(okb[0]) == oko
    assert np.allclose(ab[0], ao, rtol=1e-4, atol=1e-4)


def test_track_abort_and_cutoff_repeat(api, orc):
    P = small_problem(seed=6, n=300, noise=8.0)
    G, O = pair(api, orc, P)
    start = start_pose(orc, P, 6)
    okg, pg, _, lrg, _, _ = G.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1, min_res=[1e-3] * 5)
    oko, po, _, lro, _, _ = O.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1, min_res=[1e-3] * 5)
    assert okg == oko == False and np.array_equal(pg, start)
    assert np.allclose(lrg, lro, rtol=1e-4, equal_nan=True)
    # brightness jump of +70: > 60 % saturated at cutoff 20 -> cutoff doubling + level repeat (:694-701, :813-818)
    okg, pg, ag, lrg, _, trg = G.trackNewestCoarse(P.gt_pose, (0.04, 72.5), P.levels - 1)
    oko, po, ao, lro, _, tro = O.trackNewestCoarse(P.gt_pose, (0.04, 72.5), P.levels - 1)
    assert tro[:, 14].max() > 1 and okg == oko
    assert len(trg) == len(tro) and np.array_equal(trg[:, [0, 1, 3, 14]], tro[:, [0, 1, 3, 14]])
    assert np.allclose(ag, ao, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg5"])
def test_full_size_configs(api, orc, cfg):
    """BASELINE.json configs[1] (1241x376, 2000 pts, KITTI-00) and configs[4] (1408x376, 3000 pts, KITTI-360)."""
    from sdv_loam_amd import synthetic as syn
    if cfg == "cfg2":
        P = syn.make_tracker_problem(1241, 376, 4, 2000, seed=0, calib=syn.KITTI00, gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    else:
        P = syn.make_tracker_problem(1408, 376, 4, 3000, seed=0, calib=syn.KITTI360, gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    rng = np.random.default_rng(9)
    for r in P.ref:
        r["color"] = (r["color"] + rng.normal(0, 1.0, r["color"].shape)).astype(np.float32)
    G, O = pair(api, orc, P)
    start = orc.se3_mul(orc.se3_exp(syn.perturbation(0)), P.gt_pose)
    for lvl in range(4):
        check_res_gs(G, O, lvl, start, 0.02, 2.0, 20.0)
    okg, pg, ag, lrg, _, trg = G.trackNewestCoarse(start, (0.02, 2.0), 3)
    oko, po, ao, lro, _, tro = O.trackNewestCoarse(start, (0.02, 2.0), 3)
    dg = orc.se3_log(orc.se3_mul(pg, orc.se3_inverse(start)))
    do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(start)))
    assert okg == oko and rel_err(dg, do) < 1e-4
    # size-independent property: the known motion is recovered
    err = orc.se3_log(orc.se3_mul(pg, orc.se3_inverse(P.gt_pose)))
    assert np.linalg.norm(err) < 2e-3
    okb, pb, ab, _, _ = G.trackBatch(np.stack([start] * 3), np.tile([0.02, 2.0], (3, 1)), 3)
    for i in range(3):
        db = orc.se3_log(orc.se3_mul(pb[i], orc.se3_inverse(start)))
        assert rel_err(db, do) < 1e-4
    assert np.array_equal(pb[0], pb[1]) and np.array_equal(pb[1], pb[2])   # deterministic across workgroups


@pytest.mark.parametrize("B", [6, 64, 260])
def test_res_and_gs_batch_matches_single_calls(orc, B):
    """sdvgn_tracker_res_and_gs_batch (B pose hypotheses in one launch): every
    hypothesis' {Vec6, H, b} equals the single-call result (different partial-sum grouping: rel 1e-5, counters exact)."""
    import torch
    from sdv_loam_amd import api, synthetic as syn
    P = small_problem(seed=12, n=700)
    G = load_problem(api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=512), P)
    poses = np.stack([orc.se3_mul(orc.se3_exp(syn.perturbation(100 + i, 0.03, 0.004)), P.gt_pose) for i in range(B)])
    affs = np.stack([[0.01 * (i % 5), 0.5 * (i % 3)] for i in range(B)]).astype(np.float64)
    out = torch.zeros((B, 80), dtype=torch.float64, device="cuda")
    for lvl in (0, P.levels - 1):
        G.resAndGSBatch(lvl, poses, affs, 20.0, out_dev_ptr=C.c_void_p(out.data_ptr()))
        ext = torch.cuda.ExternalStream(G.stream())
        ext.synchronize()
        got = out.cpu().numpy()
        for i in range(0, B, max(1, B // 16)):
            r, H, b = G.resAndGS(lvl, poses[i], affs[i, 0], affs[i, 1], 20.0)
            assert got[i, 1] == r[1] and got[i, 78] >= 0                      # counters exact
            assert rel_err(got[i, 0], r[0]) < 1e-5 and rel_err(got[i, 2:6], r[2:6]) < 1e-5
            assert rel_err(got[i, 6:70].reshape(8, 8), H) < 1e-5 and rel_err(got[i, 70:78], b) < 1e-5


def _track_both(api, orc, P, start, aff0=(0.0, 0.0), settings=None, **load_kw):
    G, O = pair(api, orc, P, **load_kw)
    if settings is not None:
        G.set_settings(**settings); O.set_settings(**settings)
    rg = G.trackNewestCoarse(start, aff0, P.levels - 1)
    ro = O.trackNewestCoarse(start, aff0, P.levels - 1)
    rb = G.trackBatch(start[None], np.array([aff0]), P.levels - 1)
    return rg, ro, rb


def _assert_same_track(orc, P, start, rg, ro, rb):
    okg, pg, ag, lrg, flg, trg = rg
    oko, po, ao, lro, flo, tro = ro
    assert okg == oko
    do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(start)))
    for p in (pg, rb[1][0]):                       # host-driven LM and the device-resident k_track
        dg = orc.se3_log(orc.se3_mul(p, orc.se3_inverse(start)))
        assert rel_err(dg, do) < 1e-4
    for a in (ag, rb[2][0]):
        assert abs(a[0] - ao[0]) < 1e-4 * max(1, abs(ao[0])) and abs(a[1] - ao[1]) < 1e-4 * max(1, abs(ao[1]))
    assert np.allclose(lrg[:P.levels], lro[:P.levels], rtol=1e-4)
    assert len(trg) == len(tro) and np.array_equal(trg[:, [0, 1, 3]], tro[:, [0, 1, 3]])      # same LM path, same decisions


@pytest.mark.parametrize("modes", [(-1.0, -1.0), (-1.0, 0.0), (0.0, -1.0)])
def test_track_affine_opt_modes(api, orc, modes):
    """setting_affineOptModeA / B < 0 fix a and / or b during tracking: the three reduced-system branches of the LM step
    (CoarseTracker.cpp:726-748; the fork's `mode=2` launch sets both to -1, main.cpp:460-462)."""
    P = small_problem(seed=4, n=600, w=320, h=240, levels=3, noise=1.5)
    start = start_pose(orc, P, 4)
    aff0 = (0.01, 1.0)
    rg, ro, rb = _track_both(api, orc, P, start, aff0, settings=dict(huber=6.0, cutoff=20.0, aff_a=modes[0], aff_b=modes[1]))
    _assert_same_track(orc, P, start, rg, ro, rb)
    # a parameter that is not optimised is reported as 0 (CoarseTracker.cpp:836-837: `if(setting_affineOptModeA < 0) aff_g2l_out.a=0`)
    if modes[0] < 0:
        assert rg[2][0] == 0 and ro[2][0] == 0 and rb[2][0][0] == 0
    if modes[1] < 0:
        assert rg[2][1] == 0 and ro[2][1] == 0 and rb[2][0][1] == 0


@pytest.mark.parametrize("exposures", [(0.7, 1.9), (2.5, 0.4), (0.0, 1.3), (1.2, 0.0)])
def test_track_with_exposures(api, orc, exposures):
    """AffLight::fromToVecExposure with exposure times != 1 and with a zero exposure (both are then forced to 1, NumType.h:149-158)."""
    P = small_problem(seed=5, n=600, w=320, h=240, levels=3, noise=1.0)
    start = start_pose(orc, P, 5)
    rg, ro, rb = _track_both(api, orc, P, start, (0.0, 0.0), ref_aff=(0.02, -1.5), exposures=exposures)
    _assert_same_track(orc, P, start, rg, ro, rb)
    # a trial at the start pose: same affine transfer, same energy and counters
    G, O = pair(api, orc, P, ref_aff=(0.02, -1.5), exposures=exposures)
    check_res_gs(G, O, 0, start, 0.03, 1.0, 20.0)


def test_res_and_gs_multi_independent_problems(orc):
    """sdvgn_tracker_res_and_gs_multi: B independent problems (own template, own level image) in one launch equal the per-handle calls."""
    import torch
    from sdv_loam_amd import api, synthetic as syn
    Ps = [small_problem(seed=20 + k, n=640) for k in range(3)]
    Gs = [load_problem(api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=64), P) for P in Ps]
    B = 12
    which = [i % 3 for i in range(B)]
    poses = np.stack([orc.se3_mul(orc.se3_exp(syn.perturbation(300 + i, 0.03, 0.004)), Ps[which[i]].gt_pose) for i in range(B)])
    affs = np.stack([[0.01 * (i % 4), 0.4 * (i % 3)] for i in range(B)]).astype(np.float64)
    out = torch.zeros((B, 80), dtype=torch.float64, device="cuda")
    for lvl in (0, 2):
        pcs = [Gs[w].ref_dev(lvl) for w in which]
        imgs = [Gs[w].pyr_dev(lvl) for w in which]
        Gs[0].resAndGSMulti(lvl, pcs, imgs, poses, affs, 20.0, out_dev_ptr=C.c_void_p(out.data_ptr()))
        torch.cuda.ExternalStream(Gs[0].stream()).synchronize()
        got = out.cpu().numpy()
        for i in range(B):
            r, H, b = Gs[which[i]].resAndGS(lvl, poses[i], affs[i, 0], affs[i, 1], 20.0)
            assert got[i, 1] == r[1]
            assert rel_err(got[i, 0], r[0]) < 1e-5 and rel_err(got[i, 6:70].reshape(8, 8), H) < 1e-5 and rel_err(got[i, 70:78], b) < 1e-5


@pytest.mark.parametrize("cfg", ["cfg2", "cfg5", "small"])
def test_tolerance_mode_arithmetic(api, orc, cfg):
    """sdvgn_tracker_set_arith(1) (fused multiply-adds, reciprocal divisions) against the exact oracle: BASELINE.json's tolerance -- pose
    increments within 1e-4 relative -- holds; counters stay exact away from the decision boundaries; the exact mode remains the default."""
    from sdv_loam_amd import synthetic as syn
    if cfg == "cfg2":
        P = syn.make_tracker_problem(1241, 376, 4, 2000, seed=0, calib=syn.KITTI00, gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    elif cfg == "cfg5":
        P = syn.make_tracker_problem(1408, 376, 4, 3000, seed=1, calib=syn.KITTI360, gt_xi=[0.08, 0.04, 0.15, -0.008, 0.01, 0.004], gt_aff=(0.03, 2.0))
    else:
        P = small_problem(seed=8, n=600, w=320, h=240, levels=3, noise=1.5)
    G, O = pair(api, orc, P)
    start = start_pose(orc, P, 2)
    # one trial: H, b and the increment they give
    G.set_arith(1)
    r1, H1, b1 = G.resAndGS(0, start, 0.01, 1.0, 20.0)
    G.set_arith(0)
    r0, H0, b0 = G.resAndGS(0, start, 0.01, 1.0, 20.0)
    assert abs(r1[1] - r0[1]) <= 2
    if r1[1] == r0[1] and r1[5] == r0[5]:       # no point sits on a decision boundary (bounds, cutoff) that the last digits could flip
        assert rel_err(r1[0], r0[0]) < 1e-4
        assert rel_err(H1, H0) < 1e-4 and rel_err(b1, b0) < 1e-4
        inc1, inc0 = np.linalg.solve(H1 + 0.01 * np.diag(np.diag(H1)), -b1), np.linalg.solve(H0 + 0.01 * np.diag(np.diag(H0)), -b0)
        assert rel_err(inc1, inc0) < 1e-4
    # a whole trackNewestCoarse call in tolerance mode against the exact oracle
    G.set_arith(1)
    okg, pg, ag, lrg, _, _ = G.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
    oko, po, ao, lro, _, _ = O.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
    assert okg == oko
    dg = orc.se3_log(orc.se3_mul(pg, orc.se3_inverse(start)))
    do = orc.se3_log(orc.se3_mul(po, orc.se3_inverse(start)))
    assert rel_err(dg, do) < 1e-4
    assert np.allclose(lrg[:P.levels], lro[:P.levels], rtol=1e-3, atol=2e-4)     # RMSE at convergence on noise-free images is ~1e-4: absolute


@pytest.mark.parametrize("shape", [(1241, 376, 4), (256, 192, 3)])
def test_record_layout_bit_identical(api, orc, shape):
    """set_precision(4): the fused calcRes + calcGSSSE kernel on the gather-friendly 64-byte neighbourhood records -- the same twelve floats
    per lookup, so E, counters, H, b and a whole host-driven trackNewestCoarse are bit-identical with the row-major AoS pyramid;
    also for independent problems side by side (resAndGSMulti with per-problem record pointers)."""
    from sdv_loam_amd import synthetic as syn
    w, h, levels = shape
    P = small_problem(seed=3, n=1500, w=w, h=h, levels=levels, noise=1.5) if w < 1000 else \
        syn.make_tracker_problem(w=w, h=h, levels=levels, n_points=2000, seed=0, calib=syn.KITTI00, gt_xi=[0.03, -0.02, 0.05, 0.004, -0.006, 0.002], gt_aff=(0.03, 1.5))
    G = load_problem(api.CoarseTracker(P.w, P.h, P.levels, max_points=1 << 16, max_batch=8), P, ref_aff=(0.01, 1.0))
    pose = start_pose(orc, P, 2)
    ref = []
    for lvl in range(P.levels):
        ref.append(G.resAndGS(lvl, pose, 0.03, 2.0, 20.0))
    tr0 = G.trackNewestCoarse(pose, (0.0, 0.0), P.levels - 1)
    G.set_precision(4)
    for lvl in range(P.levels):
        r, H, b = G.resAndGS(lvl, pose, 0.03, 2.0, 20.0)
        assert np.array_equal(r, ref[lvl][0], equal_nan=True) and np.array_equal(H, ref[lvl][1], equal_nan=True) and np.array_equal(b, ref[lvl][2], equal_nan=True)
    tr1 = G.trackNewestCoarse(pose, (0.0, 0.0), P.levels - 1)
    assert tr0[0] == tr1[0] and np.array_equal(tr0[1], tr1[1]) and np.array_equal(tr0[2], tr1[2]) and np.array_equal(tr0[3], tr1[3], equal_nan=True)
    # independent problems: two trackers with their own pyramids / templates, one launch
    P2 = small_problem(seed=9, n=1500, w=w, h=h, levels=levels, noise=1.0) if w < 1000 else \
        syn.make_tracker_problem(w=w, h=h, levels=levels, n_points=2000, seed=5, calib=syn.KITTI00, gt_xi=[0.01, 0.02, -0.03, 0.002, 0.003, -0.001], gt_aff=(0.0, 0.5))
    G2 = load_problem(api.CoarseTracker(P.w, P.h, P.levels, max_points=1 << 16, max_batch=8), P2, ref_aff=(0.01, 1.0))
    poses = np.stack([pose, pose, start_pose(orc, P2, 4), start_pose(orc, P2, 5)])
    affs = np.tile([0.03, 2.0], (4, 1))
    pcs = [G.ref_dev(0), G.ref_dev(0), G2.ref_dev(0), G2.ref_dev(0)]
    import torch
    o0 = torch.zeros(4 * 80, dtype=torch.float64, device="cuda")
    o4 = torch.zeros(4 * 80, dtype=torch.float64, device="cuda")
    G.set_precision(0)
    G.resAndGSMulti(0, pcs, [G.pyr_dev(0), G.pyr_dev(0), G2.pyr_dev(0), G2.pyr_dev(0)], poses, affs, 20.0, o0.data_ptr())
    G.set_precision(4)
    G.resAndGSMulti(0, pcs, [G.records_dev(0), G.records_dev(0), G2.records_dev(0), G2.records_dev(0)], poses, affs, 20.0, o4.data_ptr())
    torch.cuda.synchronize()
    a, b = o0.cpu().numpy().reshape(4, 80), o4.cpu().numpy().reshape(4, 80)
    assert np.abs(a[:, 6:70]).max() > 0 and np.array_equal(a, b, equal_nan=True)
    assert not np.array_equal(a[0], a[2])                                          # (the problems really are different)
