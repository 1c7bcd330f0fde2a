"""The rows of bench.py that are not the headline: tracker (configs[1]), the cfg5 precision study, reprojector / trace / activation batches, marginalisation -- each
returns a dictionary that goes to bench_extras.json (never into the one stdout line).  Moved out of bench.py in round 6 ("bench.py 1 250 lines");
the measuring code is unchanged.  Also the two HIP-event helpers every leg uses."""
import time

import numpy as np

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
TRACKER_BYTES_PER_POINT = 64     # 16 B point record + 4 taps x 12 B {I,dx,dy}


def event_ms(torch, stream, fn, reps):
    """median duration of fn() in ms, HIP events recorded on `stream` (the stream the library launches on)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def event_avg_ms(torch, stream, fn, reps):
    """average duration of fn() in ms over `reps` back-to-back calls bracketed by ONE pair of HIP events on `stream`: launch
    duration incl. the gap to the next launch, without the per-launch event packets of event_ms()."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def tracker_problem():
    from sdv_loam_amd import synthetic as syn
    P = syn.make_tracker_problem(1241, 376, 4, 2000, seed=0, calib=syn.KITTI00,
                                 gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    rng = np.random.default_rng(9)
    for r in P.ref:
        r["color"] = (r["color"] + rng.normal(0, 1.0, r["color"].shape)).astype(np.float32)
    return P


def load_tracker(api, P, local, max_batch):
    G = api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=max_batch, device=local)
    G.makeK(**P.calib)
    for l in range(P.levels):
        G.set_ref(l, **P.ref[l])
    G.set_ref_frame(1.0, 0.0, 0.0)
    G.set_new_image(P.image, 1.0)
    return G


def distinct_batch(api, oracle, P, G, local, batch, n_problems=64, records=False):
    """`batch` LM trials as n_problems INDEPENDENT tracking problems x batch / n_problems poses: every problem has its own reference
    template and its own target pyramid in HBM (n_problems x 5.6 MB of level-0 image = 360 MB at 64, beyond the 256 MB Infinity Cache),
    so the bytes the launch moves are its algorithmic bytes.  Returns (trackers, launch)."""
    from sdv_loam_amd import synthetic as syn
    Gs = [G] + [load_tracker(api, P, local, 1) for _ in range(n_problems - 1)]
    per = batch // n_problems
    poses = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(1000 + i)), P.gt_pose) for i in range(batch)])
    affs = np.tile([0.02, 2.0], (batch, 1))
    pcs = [Gs[i // per].ref_dev(0) for i in range(batch)]
    if records:      # set_precision(4): every problem's level-0 image as 64-byte neighbourhood records (30 MB per problem: 1.9 GB at 64)
        G.set_precision(4)
        imgs = [Gs[i // per].records_dev(0) for i in range(batch)]
    else:
        imgs = [Gs[i // per].pyr_dev(0) for i in range(batch)]
    return Gs, (lambda: G.resAndGSMulti(0, pcs, imgs, poses, affs, 20.0))


def tracker_extras(torch, local, batch, oracle, want_cpu):
    from sdv_loam_amd import api, synthetic as syn
    P = tracker_problem()
    G = load_tracker(api, P, local, max(batch, 64))
    ext = torch.cuda.ExternalStream(G.stream(), device=torch.device("cuda", local))
    start = oracle.se3_mul(oracle.se3_exp(syn.perturbation(0)), P.gt_pose)
    out = {}
    # (i) host-driven LM trial: fused launch + 640-B read-back per trial
    for _ in range(20):
        G.resAndGS(0, start, 0.02, 2.0, 20.0)
    t0 = time.perf_counter()
    for _ in range(300):
        G.resAndGS(0, start, 0.02, 2.0, 20.0)
    out["host_driven_trials_per_s"] = 300 / (time.perf_counter() - t0)
    # (ii) whole trackNewestCoarse calls: host-driven and device-resident (1 and 31 hypotheses)
    _, _, _, _, _, tr = G.trackNewestCoarse(start, (0.02, 2.0), 3)
    ntr = max(len(tr), 1)
    t0 = time.perf_counter()
    for _ in range(30):
        G.trackNewestCoarse(start, (0.02, 2.0), 3)
    dt = (time.perf_counter() - t0) / 30
    out["track_call_host_driven_ms"] = 1e3 * dt
    out["track_call_lm_trials"] = ntr
    for B in (1, 31):
        starts = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(i)), P.gt_pose) for i in range(B)])
        affs = np.tile([0.02, 2.0], (B, 1))
        for team, tag in ((0, ""), (-1, "_one_workgroup")):      # k_track_team (automatic team size) and k_track
            G.set_team(team)
            G.trackBatch(starts, affs, 3)
            t0 = time.perf_counter()
            for _ in range(20):
                G.trackBatch(starts, affs, 3)
            out["track_call_device_resident_B%d%s_ms" % (B, tag)] = 1e3 * (time.perf_counter() - t0) / 20
            if team == 0:
                out["track_call_device_resident_B%d_team" % B] = G.last_team()
        G.set_team(0)
    # (ii-a') roofline view of configs[1] computed like the back end's: algorithmic bytes of one device-resident trackNewestCoarse call -- every
    # evaluation of the LM loop (one initial calcRes per level + one per trial; the host-driven trace lists the trials and their levels, the
    # device-resident loop takes the same ones) x points of that level x 64 B -- over the duration of the call's launch(es) on the
    # library stream.  The working set (7.4 MB pyramid + templates) is cache-resident and the loop is a latency chain: the fraction says so.
    lv = np.asarray(tr)[:, 0].astype(int) if len(tr) else np.zeros(0, int)
    evals = {l: int((lv == l).sum()) + 1 for l in range(P.levels)}
    alg_call = sum(evals[l] * P.ref[l]["u"].size * TRACKER_BYTES_PER_POINT for l in range(P.levels))
    st1r, af1r = start[None].copy(), np.array([[0.02, 2.0]])
    G.set_team(0)
    G.trackBatch(st1r, af1r, 3)
    ms_call = event_avg_ms(torch, ext, lambda: G.trackBatch(st1r, af1r, 3), 20)
    out["roofline"] = dict(bound="hbm", kernel="k_track_team (whole trackNewestCoarse on the device, B = 1, configs[1])", achieved=alg_call / (ms_call * 1e-3) / 1e9,
                           peak=HBM_PEAK_GBS, unit="GB/s", frac=alg_call / (ms_call * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None,
                           evaluations_per_level=evals, algorithmic_bytes_per_call=alg_call, ms_per_call_on_stream=ms_call,
                           note="one call = %d evaluations of <= 2000 points x 64 B = %.2f MB algorithmic over a working set that lives in the caches; "
                                "the call is a chain of dependent evaluations (latency-bound), see `batched*` for the launches where a bandwidth "
                                "fraction means something" % (sum(evals.values()), alg_call / 1e6))
    # (ii-b) PCIe-inclusive: a new frame handed over as a host buffer (H2D of w*h floats + 4 pyramid launches) followed by one
    # device-resident track -- what a caller that does not keep images on the device pays per frame.  Never the headline value.
    st1 = start[None].copy()
    af1 = np.array([[0.02, 2.0]])
    t0 = time.perf_counter()
    for _ in range(20):
        G.set_new_image(P.image, 1.0)
        G.trackBatch(st1, af1, 3)
    out["frame_upload_pyramid_track_ms_pcie_inclusive"] = 1e3 * (time.perf_counter() - t0) / 20
    # (ii-c) structPoseEstimation: 1200 matches, whole 10-iteration LM in one single-workgroup launch; the call
    # includes packing + H2D of the 34 kB of inputs and the 1.7 kB read-back (host buffers at the boundary)
    SP = syn.make_struct_problem(n=1200, seed=0)
    sp_args = (SP.u, SP.v, SP.idepth, SP.host_idx, SP.host_poses7, SP.obs)
    for _ in range(3):
        G.structPoseEstimation(SP.init_curToWorld7, *sp_args)
    t0 = time.perf_counter()
    for _ in range(50):
        _, sp_tr, _ = G.structPoseEstimation(SP.init_curToWorld7, *sp_args)
    out["struct_pose_call_ms"] = 1e3 * (time.perf_counter() - t0) / 50
    out["struct_pose_lm_iterations"] = len(sp_tr)
    # (ii-d) setCoarseTrackingRef: reference template on the device from 14 000 splat tuples (row a3), vs the oracle
    rng_cd = np.random.default_rng(4)
    cd = (rng_cd.integers(0, P.w, 14000).astype(np.int32), rng_cd.integers(0, P.h, 14000).astype(np.int32),
          rng_cd.uniform(0.02, 0.5, 14000).astype(np.float32), rng_cd.uniform(0.3, 3.0, 14000).astype(np.float32))
    GT = api.CoarseTracker(P.w, P.h, P.levels, max_points=P.w * P.h, max_batch=2, device=local)
    GT.makeK(**P.calib)
    GT.set_new_image(P.image, 1.0)
    for _ in range(3):
        GT.makeCoarseDepth(*cd)
    t0 = time.perf_counter()
    for _ in range(20):
        GT.makeCoarseDepth(*cd)
    out["make_coarse_depth_ms"] = 1e3 * (time.perf_counter() - t0) / 20
    out["make_coarse_depth_template_points_lvl0"] = int(GT.n[0])
    # (ii-e) the same tracker calls on that dense template (what the reference really tracks with: tens of thousands of points per level)
    GT.set_ref_frame(1.0, 0.0, 0.0)
    dense_pose = oracle.se3_exp(np.array([0.02, -0.01, 0.03, 0.002, -0.001, 0.0015]))
    GT.trackNewestCoarse(dense_pose, (0.0, 0.0), 3)
    t0 = time.perf_counter()
    for _ in range(10):
        GT.trackNewestCoarse(dense_pose, (0.0, 0.0), 3)
    out["dense_template_track_call_host_driven_ms"] = 1e3 * (time.perf_counter() - t0) / 10
    for team, tag in ((0, ""), (-1, "_one_workgroup")):
        GT.set_team(team)
        GT.trackBatch(dense_pose[None], np.zeros((1, 2)), 3)
        t0 = time.perf_counter()
        for _ in range(10):
            GT.trackBatch(dense_pose[None], np.zeros((1, 2)), 3)
        out["dense_template_track_call_device_resident%s_ms" % tag] = 1e3 * (time.perf_counter() - t0) / 10
        if team == 0:
            out["dense_template_team"] = GT.last_team()
    del GT
    # (iii) batched roofline run of the fused tracker kernel
    poses = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(1000 + i)), P.gt_pose) for i in range(batch)])
    affs = np.tile([0.02, 2.0], (batch, 1))
    for _ in range(3):
        G.resAndGSBatch(0, poses, affs, 20.0)
    torch.cuda.synchronize()
    ms = event_ms(torch, ext, lambda: G.resAndGSBatch(0, poses, affs, 20.0), 20)
    alg = batch * P.ref[0]["u"].size * TRACKER_BYTES_PER_POINT
    out["batched"] = dict(kernel="k_res_gs+k_finalize", lm_trials_per_launch=batch, ms_per_launch=ms, gn_iters_per_s=batch / (ms * 1e-3),
                          algorithmic_GBps=alg / (ms * 1e-3) / 1e9, frac_of_hbm_peak=alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          note="all trials share ONE template and ONE 5.6 MB image: an algorithmic rate of a cache-resident working set, not an HBM measurement")
    G.set_arith(1)
    for _ in range(3):
        G.resAndGSBatch(0, poses, affs, 20.0)
    ms_t = event_ms(torch, ext, lambda: G.resAndGSBatch(0, poses, affs, 20.0), 20)
    G.set_arith(0)
    out["batched"]["tolerance_mode"] = dict(ms_per_launch=ms_t, gn_iters_per_s=batch / (ms_t * 1e-3), algorithmic_GBps=alg / (ms_t * 1e-3) / 1e9)
    # (iv) the same number of trials as 64 independent problems (own template, own pyramid: 360 MB footprint), both arithmetic modes
    Gs, launch = distinct_batch(api, oracle, P, G, local, batch)
    dist = dict(problems=len(Gs), lm_trials_per_launch=batch, footprint_MB=len(Gs) * (P.w * P.h * 12 + 32000) / 1e6)
    for mode, name in ((0, "exact"), (1, "tolerance_mode")):
        G.set_arith(mode)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        msd = event_ms(torch, ext, launch, 20)
        dist[name] = dict(ms_per_launch=msd, gn_iters_per_s=batch / (msd * 1e-3), algorithmic_GBps=alg / (msd * 1e-3) / 1e9,
                          frac_of_hbm_peak=alg / (msd * 1e-3) / 1e9 / HBM_PEAK_GBS)
    G.set_arith(0)
    del Gs
    # the same launch on the gather-friendly record copies of the 64 pyramids (set_precision(4); bit-identical results)
    Gs, launch = distinct_batch(api, oracle, P, G, local, batch, records=True)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    msr = event_ms(torch, ext, launch, 20)
    dist["record_layout"] = dict(ms_per_launch=msr, gn_iters_per_s=batch / (msr * 1e-3), algorithmic_GBps=alg / (msr * 1e-3) / 1e9,
                                 frac_of_hbm_peak=alg / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS, footprint_MB=len(Gs) * (P.w * P.h * 64) / 1e6,
                                 note="level-0 images as 64-byte 2x2-neighbourhood records per pixel (sdvgn_tracker_set_precision(4)): one 128-byte line per lookup")
    G.set_precision(0)
    del Gs
    out["batched_independent_problems"] = dist
    if want_cpu:
        O = oracle.OracleTracker(P.w, P.h, P.levels)
        O.makeK(**P.calib)
        for l in range(P.levels):
            O.set_ref(l, **P.ref[l])
        O.set_ref_frame(1.0, 0.0, 0.0)
        O.set_new_image(P.image, 1.0)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            for _ in range(200):
                O.calcRes(0, start, 0.02, 2.0, 20.0)
                O.calcGS(0, 0.02, 2.0)
            n += 200
        out["cpu_trials_per_s_1thread"] = n / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(50):
            O.trackNewestCoarse(start, (0.02, 2.0), 3)
        out["cpu_track_call_ms_1thread"] = 1e3 * (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50):
            O.structPoseEstimation(SP.init_curToWorld7, *sp_args)
        out["cpu_struct_pose_call_ms_1thread"] = 1e3 * (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(5):
            O.makeCoarseDepth(*cd)
        out["cpu_make_coarse_depth_ms_1thread"] = 1e3 * (time.perf_counter() - t0) / 5
        O.set_ref_frame(1.0, 0.0, 0.0)
        t0 = time.perf_counter()
        for _ in range(3):
            O.trackNewestCoarse(dense_pose, (0.0, 0.0), 3)
        out["cpu_dense_template_track_call_ms_1thread"] = 1e3 * (time.perf_counter() - t0) / 3
    return out


def cfg5_extras(torch, local, oracle, want_cpu, batch=1024):
    """BASELINE.json configs[4]: KITTI-360 calib, 1408 x 376, 3000 points per level, fp32 vs fp16 (tolerance study): per precision mode of
    sdvgn_tracker_set_precision the Gauss-Newton rate (single host-driven LM trials, whole trackNewestCoarse calls, and `batch` trials of level 0
    in one launch) AND the error it costs -- H of the first trial, the pose increment of the whole call against the fp32 CPU oracle, the
    distance to the ground-truth motion."""
    from sdv_loam_amd import api, synthetic as syn
    P = syn.make_tracker_problem(1408, 376, 4, 3000, seed=0, calib=syn.KITTI360, gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    rng = np.random.default_rng(9)
    for r in P.ref:
        r["color"] = (r["color"] + rng.normal(0, 1.0, r["color"].shape)).astype(np.float32)
    start = oracle.se3_mul(oracle.se3_exp(syn.perturbation(0)), P.gt_pose)

    def load(T):
        T.makeK(**P.calib)
        for l in range(P.levels):
            T.set_ref(l, **P.ref[l])
        T.set_ref_frame(1.0, 0.0, 0.0)
        T.set_new_image(P.image, 1.0)
        return T

    O = load(oracle.OracleTracker(P.w, P.h, P.levels))
    t0 = time.perf_counter()
    oko, po, ao, lro, _, tro = O.trackNewestCoarse(start, (0.02, 2.0), 3)
    cpu_call_ms = 1e3 * (time.perf_counter() - t0)
    do = oracle.se3_log(oracle.se3_mul(po, oracle.se3_inverse(start)))
    O.calcRes(0, start, 0.02, 2.0, 20.0)
    Ho, bo = O.calcGS(0, 0.02, 2.0)
    G = load(api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=batch, device=local))
    ext = torch.cuda.ExternalStream(G.stream(), device=torch.device("cuda", local))
    poses = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(2000 + i)), P.gt_pose) for i in range(batch)])
    affs = np.tile([0.02, 2.0], (batch, 1))
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))   # noqa: E731
    names = {0: "fp32 (product path)", 1: "fp16 pyramid", 2: "fp16 pyramid + fp16 J/r operands", 3: "fp16 pyramid + operands + fp16 accumulation"}
    modes = {}
    for mode, name in names.items():
        G.set_precision(mode)
        Hg, bg = G.calcGS(0, start, 0.02, 2.0, 20.0)
        okg, pg, ag, lrg, _, trg = G.trackNewestCoarse(start, (0.02, 2.0), 3)
        dg = oracle.se3_log(oracle.se3_mul(pg, oracle.se3_inverse(start)))
        egt = oracle.se3_log(oracle.se3_mul(pg, oracle.se3_inverse(P.gt_pose)))
        for _ in range(20):
            G.resAndGS(0, start, 0.02, 2.0, 20.0)
        t0 = time.perf_counter()
        for _ in range(200):
            G.resAndGS(0, start, 0.02, 2.0, 20.0)
        trial_rate = 200 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(10):
            G.trackNewestCoarse(start, (0.02, 2.0), 3)
        call_ms = 1e3 * (time.perf_counter() - t0) / 10
        for _ in range(3):
            G.resAndGSBatch(0, poses, affs, 20.0)
        torch.cuda.synchronize()
        msb = event_ms(torch, ext, lambda: G.resAndGSBatch(0, poses, affs, 20.0), 10)
        fin = lambda x: (float(x) if np.isfinite(x) else None)   # noqa: E731
        modes[str(mode)] = dict(name=name, ok=bool(okg), gn_iters_per_s_host_driven_trials=trial_rate, track_call_ms=call_ms, lm_trials_in_call=len(trg),
                                gn_iters_per_s_batched=batch / (msb * 1e-3), batched_ms_per_launch=msb,
                                H_rel_error_vs_fp32_oracle=fin(rel(Hg, Ho)), b_rel_error=fin(rel(bg, bo)),
                                pose_increment_rel_error_vs_fp32_oracle=fin(rel(dg, do)),
                                gt_error_translation=fin(np.linalg.norm(egt[:3])), gt_error_rotation=fin(np.linalg.norm(egt[3:])),
                                affine_abs_error=fin(np.abs(ag - ao).max()), rmse_level0=fin(lrg[0]))
    G.set_precision(0)
    out = dict(workload="configs[4]: KITTI-360 calib 1408x376, 4 levels, 3000 points per level; one GN iteration = one LM trial (calcRes + calcGSSSE of one "
                        "pose); fp32 = the product path, the fp16 modes exist for this study only (sdvgn_tracker_set_precision)",
               batch=batch, tolerance="BASELINE.json north_star: 1e-4 relative on pose increments", modes=modes)
    if want_cpu:
        out["cpu_oracle_fp32_track_call_ms_1thread"] = cpu_call_ms
        out["cpu_oracle_lm_trials"] = len(tro)
    return out


def reproject_extras(W, G, local, want_cpu):
    """SURVEY 8f-2 at the named shape: the last key-frame of the window plays the new frame, the 14 000 active points of the other
    seven are the candidates; one call = reprojectPoint + findMatchDirect for ALL of them (the reference evaluates them lazily, about
    one to two per grid cell until 0.8*desiredImmatureDensity matches are found)."""
    from sdv_loam_amd import reproject_api, synthetic as syn
    P = syn.make_reproject_problem(W, levels=4, seed=0, pose_err=(0.02, 0.002))
    R = reproject_api.Reprojector(P.w, P.h, P.levels, max_frames=8, max_points=P.n, device=local)
    R.set_calib(**P.calib)
    for k in range(len(P.frame_poses7)):
        R.set_frame(k, P.frame_poses7[k], P.frame_images[k])
    R.set_cur(P.cur_pose7, P.cur_pyr)
    a = (P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    for _ in range(3):
        g = R.match(*a)
    t0 = time.perf_counter()
    for _ in range(20):
        g = R.match(*a)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    out = dict(candidates=int(P.n), in_grid=int((g["cell"] >= 0).sum()), matched=int(g["success"].sum()), call_ms=ms,
               candidates_per_s=P.n / (ms * 1e-3))
    if want_cpu:
        from oracle.reproject import OracleReprojector
        O = OracleReprojector(P.w, P.h, P.levels)
        O.set_calib(**P.calib)
        for k in range(len(P.frame_poses7)):
            O.set_frame(k, P.frame_poses7[k], P.frame_images[k])
        O.set_cur(P.cur_pose7, P.cur_pyr)
        t0 = time.perf_counter()
        px0, cell, q = O.project(P.u, P.v, P.idepth, P.host_idx)
        sel = cell >= 0
        ok, pm, _ = O.find_match(P.u[sel], P.v[sel], P.idepth[sel], P.host_idx[sel], P.ref_idx[sel], P.type[sel], px0[sel])
        dt = time.perf_counter() - t0
        out["cpu_all_candidates_ms_1thread"] = 1e3 * dt
        out["cpu_us_per_candidate"] = 1e6 * dt / max(int(sel.sum()), 1)
    return out


def trace_extras(W, local, want_cpu):
    """SURVEY 8f-4 at the named shape: the 14 000 points of seven key-frames as immature points, traced on the eighth frame
    (ImmaturePoint::traceOn for all of them = the loop of FullSystem::traceNewCoarse), first (uninitialised) pass."""
    from sdv_loam_amd import api, synthetic as syn
    P = syn.make_trace_problem(W, seed=0)
    T = api.CoarseTracker(P.w, P.h, 4, max_points=64, device=local)
    T.makeK(**W.calib)
    T.set_new_image(P.image, 1.0)
    T.traceSetPoints(P.u, P.v, P.energyTH, P.gradH, P.color, P.weights, P.host_idx)
    a = (P.KRKi, P.Kt, P.aff, P.idepth_min, P.idepth_max, P.quality, P.status)
    for _ in range(3):
        st = T.tracePoints(*a)
    t0 = time.perf_counter()
    for _ in range(20):
        st = T.tracePoints(*a)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    out = dict(points=int(P.n), good=int((st["status"] == 0).sum()), call_ms=ms, points_per_s=P.n / (ms * 1e-3))
    if want_cpu:
        from oracle.trace import trace_on
        t0 = time.perf_counter()
        trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status)
        dt = time.perf_counter() - t0
        out["cpu_ms_1thread"] = 1e3 * dt
        out["cpu_us_per_point"] = 1e6 * dt / P.n
    return out


def immature_extras(W, G, want_cpu):
    """FullSystem::optimizeImmaturePoint for the window's 16 000 points treated as activation candidates (inverse-depth
    interval +-10-20 % around the truth), one launch on the back-end handle that already holds the frames."""
    rng = np.random.default_rng(3)
    lo = rng.uniform(0.02, 0.2, W.nP).astype(np.float32)
    hi = rng.uniform(0.02, 0.2, W.nP).astype(np.float32)
    a = (W.host, W.u, W.v, (W.idepth * (1 - lo)).astype(np.float32), (W.idepth * (1 + hi)).astype(np.float32), np.full(W.nP, 8 * 144, np.float32),
         W.color, W.weights, W.isFromSensor)
    G.load(W)
    for _ in range(3):
        r = G.optimizeImmature(*a)
    t0 = time.perf_counter()
    for _ in range(20):
        r = G.optimizeImmature(*a)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    out = dict(points=int(W.nP), activated=int((r[0] == 1).sum()), call_ms=ms, points_per_s=W.nP / (ms * 1e-3))
    if want_cpu:
        from oracle.backend import OracleEF
        O = OracleEF(W.w, W.h).load(W)
        t0 = time.perf_counter()
        O.optimizeImmature(*a)
        dt = time.perf_counter() - t0
        out["cpu_ms_1thread"] = 1e3 * dt
        out["cpu_us_per_point"] = 1e6 * dt / W.nP
    return out


def marginalize_extras(torch, W, G, want_cpu):
    """SURVEY 8 row b2 mode 2: the marginalisation step of a key-frame on the cfg3 window -- 10 % of the points leave (resetOOB +
    fixLinearizationF + marginalizePointsF; the window is reloaded, untimed, before every repetition because the points are removed),
    and marginalizeFrame (host algebra)."""
    rng = np.random.default_rng(11)
    mask = (rng.random(W.nP) < 0.10).astype(np.uint8)
    tt, reps = 0.0, 8
    for k in range(reps + 1):
        G.load(W)
        G.linearizeAll(want_energy=False); G.applyRes()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        G.resetOOB(mask)
        G.linearizeAll(want_energy=False); G.applyRes()
        G.fixLinearization(mask)
        G.marginalizePoints(mask)
        dt = time.perf_counter() - t0
        if k:
            tt += dt
    t0 = time.perf_counter()
    for _ in range(50):
        G.marginalizeFrame(3)
    out = dict(points_leaving=int(mask.sum()), flag_fix_marginalize_points_ms=1e3 * tt / reps, marginalize_frame_ms=1e3 * (time.perf_counter() - t0) / 50)
    if want_cpu:
        from oracle.backend import OracleEF
        O = OracleEF(W.w, W.h).load(W)
        O.linearizeAll(); O.applyRes()
        t0 = time.perf_counter()
        O.resetOOB(mask)
        O.linearizeAll(); O.applyRes()
        O.fixLinearization(mask)
        O.marginalizePoints(mask)
        out["cpu_ms_1thread"] = 1e3 * (time.perf_counter() - t0)
        out["cpu_note"] = "the oracle re-linearises the whole window like the device path (the reference touches only the departing points)"
    G.load(W)
    return out
