"""The window edited in place (include/sdvgn.h: sdvgn_ef_insert_frame / _insert_points / _insert_residuals / _drop_residuals / _remove_points /
_remove_frame / _update_residuals / _make_idx, csrc/backend_window.inc) against a full reload of the same graph through the whole-plane
setters: after the commit the residual tables, states, optimize traces, per-point sums and the next solve must be BIT-IDENTICAL.

The reference mutates its EnergyFunctional between two FullSystem::optimize calls (EnergyFunctional.h:51-58, FullSystem::makeKeyFrame
FullSystem.cpp:1040-1180); both ends of such a key-frame step are cut out of ONE larger synthetic window (synthetic.subwindow), so that every
image, point and matcher of the "after" graph is consistent with the "before" graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAL = dict(fx=400., fy=410., cx=319.5, cy=119.5)


@pytest.fixture(scope="module")
def api(sdvgn_lib):
    from sdv_loam_amd import backend_api
    return backend_api


@pytest.fixture(scope="module")
def big():
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=640, h=240, nF=7, pts_per_kf=300, seed=11, calib=CAL)


class Mirror:
    """Test-side book-keeping of one edited window: which big-window frame / point / residual every library-side object is."""

    def __init__(self, api, W8, frames, points, rmask, seed):
        from sdv_loam_amd import synthetic as syn
        self.api, self.syn, self.W8 = api, syn, W8
        self.rng = np.random.default_rng(seed)
        n = 4 + 6 * len(frames)
        A = self.rng.normal(0, 1, (n, n))
        self.frames = list(frames)
        S = syn.subwindow(W8, frames, points, rmask, HM=1e2 * (A @ A.T) / n, bM=self.rng.normal(0, 10, n))
        self.G = api.EnergyFunctional(W8.w, W8.h, max_points=W8.nP).load(S)
        self.id2big = {i: int(p) for i, p in enumerate(S.p_src)}          # a window loaded through the setters: id = dense index
        self.order = list(range(S.nP))                                      # dense index -> id
        self.rmask = np.zeros(W8.nR, bool); self.rmask[S.r_src] = True      # residuals of the big window that exist
        self.W = S

    def snapshot(self, vmz):
        """what the device holds after an optimize, in the big window's indexing; calib re-installed the way a host loop re-sends it"""
        G = self.G
        vs, st, idp = G.state()
        G.set_calib(vs, vmz)
        tb = G.residual_table()
        th = G.frame_energy_th()
        return dict(vs=vs, vmz=np.asarray(vmz, np.float64), st=st, idp=idp, tb=tb, th=th, order=list(self.order), frames=list(self.frames),
                    big_of_idx=np.array([self.id2big[i] for i in self.order]))

    def reload_window(self, snap, ids, th_new, HM, bM, state_rows, evalPT_rows, zero_rows):
        """the Window a full reload of the edited graph needs: big-window constants, device-held values for what survived"""
        W8, syn = self.W8, self.syn
        pts = np.array([self.id2big[i] for i in ids])
        A = syn.subwindow(W8, self.frames, pts, self.rmask, HM=HM, bM=bM)
        old_idx_of_big = {int(b): k for k, b in enumerate(snap["big_of_idx"])}
        old_t_of_big = {int(f): k for k, f in enumerate(snap["frames"])}
        for k, b in enumerate(pts):
            if int(b) in old_idx_of_big:                       # a survivor: the inverse depth the last optimize left
                A.idepth[k] = snap["idp"][old_idx_of_big[int(b)]]
                A.idepth_zero[k] = A.idepth[k]
        for r in range(A.nR):                                   # surviving residuals keep state_state / isActive
            b, f = int(pts[A.r_point[r]]), int(self.frames[A.r_target[r]])
            if b in old_idx_of_big and f in old_t_of_big and snap["tb"]["exists"][old_t_of_big[f], old_idx_of_big[b]]:
                A.r_state[r] = snap["tb"]["state"][old_t_of_big[f], old_idx_of_big[b]]
                A.r_isActive[r] = snap["tb"]["active"][old_t_of_big[f], old_idx_of_big[b]]
        A.value_scaled, A.value_minus_value_zero = snap["vs"], snap["vmz"]
        A.state, A.evalPT, A.state_zero = state_rows, evalPT_rows, zero_rows
        A.frameEnergyTH = np.array([snap["th"][old_t_of_big[f]] if f in old_t_of_big else th_new for f in self.frames], np.float32)
        return A


def check_equal(G, R, its=6):
    tg, tr = G.residual_table(), R.residual_table()
    for k in ("exists", "state", "active"):
        assert np.array_equal(tg[k], tr[k]), k
    for a, b in zip(G.state(), R.state()):
        assert np.array_equal(a, b)
    assert np.array_equal(G.frame_energy_th(), R.frame_energy_th())
    for a, b in zip(G.marg_prior(), R.marg_prior()):
        assert np.array_equal(a, b)
    ta, tb = G.optimize(its), R.optimize(its)
    assert len(ta) >= 2 and (ta[:, 2] == 1).any()
    assert np.array_equal(ta, tb)
    for a, b in zip(G.state(), R.state()):
        assert np.array_equal(a, b)
    assert np.array_equal(G.points(), R.points())
    tg, tr = G.residual_table(), R.residual_table()
    for k in tg:
        assert np.array_equal(tg[k], tr[k]), k
    assert np.array_equal(G.solveSystemF(3, 0.1), R.solveSystemF(3, 0.1))
    eg, er = G.optimize_finish(), R.optimize_finish()
    assert eg[0] == er[0] and np.array_equal(eg[1], er[1]) and np.array_equal(eg[2], er[2])
    return eg


def keyframe_step(M, api, drop_frame_big, new_frame_big, n_remove, n_new, seed):
    """one key-frame's worth of edits on M.G, then the same graph reloaded on a fresh handle; returns that handle"""
    W8, G, rng = M.W8, M.G, np.random.default_rng(seed)
    snap = M.snapshot(vmz=np.array([1e-3, -5e-4, 2e-3, 1e-3]) / 50. * (1 + 0.1 * seed))
    nF_b = len(M.frames)
    # ---- removePoint: random points + every point of the frame that is about to be marginalised ----
    t_drop = M.frames.index(drop_frame_big)
    alive = list(M.order)
    gone = [i for i in alive if W8.host[M.id2big[i]] == drop_frame_big]
    rest = [i for i in alive if i not in gone]
    gone += list(rng.choice(rest, n_remove, replace=False))
    G.removePoints(np.array(gone, np.int32))
    for i in gone:
        M.rmask[W8.r_point == M.id2big[i]] = False
    alive = [i for i in alive if i not in set(gone)]
    big2id = {M.id2big[i]: i for i in alive}
    # ---- dropResidual: a tenth of what is left (by (id, target index BEFORE the frame leaves)) ----
    live_r = np.nonzero(M.rmask & np.isin(W8.r_point, list(big2id)) & (W8.r_target != drop_frame_big))[0]
    dr = rng.choice(live_r, len(live_r) // 10, replace=False)
    G.dropResiduals([big2id[int(W8.r_point[r])] for r in dr], [M.frames.index(int(W8.r_target[r])) for r in dr])
    M.rmask[dr] = False
    # ---- marginalizeFrame (the library's Schur complement), insertFrame ----
    HM_ref, bM_ref = G.marginalizeFrame(t_drop)                      # (pure function of the window before the edits)
    G.removeFrame(t_drop)
    M.rmask[W8.r_target == drop_frame_big] = False
    M.frames.remove(drop_frame_big)
    th_new = np.float32(8 * 8 * 8)
    k = G.insertFrame(W8.evalPT[new_frame_big], W8.state[new_frame_big], W8.state_zero[new_frame_big], int(W8.frameID[new_frame_big]), 1.0, th_new,
                      dI=W8.pyr0[new_frame_big] if seed % 2 else None, image=None if seed % 2 else W8.images[new_frame_big])
    M.frames.append(new_frame_big)
    assert k == len(M.frames) - 1
    # ---- insertPoint: unused points of the big window hosted by surviving frames (and by the new one) ----
    used = set(big2id) | {M.id2big[i] for i in gone}                 # (a point that left at an EARLIER key-frame may come back: as a new point under a new id)
    cand = [p for p in range(W8.nP) if p not in used and W8.host[p] in M.frames]
    newp = np.sort(rng.choice(cand, n_new, replace=False))
    ids = G.insertPoints([M.frames.index(int(W8.host[p])) for p in newp], W8.u[newp], W8.v[newp], W8.idepth[newp], W8.idepth_zero[newp], W8.color[newp],
                         W8.weights[newp], W8.hasDepthPrior[newp], W8.isFromSensor[newp])
    for i, p in zip(ids, newp):
        M.id2big[int(i)] = int(p)
        big2id[int(p)] = int(i)
    # ---- insertResidual: every surviving point towards the new frame + the new points towards every frame; a few old matchers move ----
    add = np.nonzero(~M.rmask & np.isin(W8.r_point, list(big2id)) & np.isin(W8.r_target, M.frames) &
                     ((W8.r_target == new_frame_big) | np.isin(W8.r_point, newp)))[0]
    add = add[rng.random(len(add)) < 0.9]
    hm = (rng.random(len(add)) < 0.95).astype(np.uint8)
    G.insertResiduals([big2id[int(W8.r_point[r])] for r in add], [M.frames.index(int(W8.r_target[r])) for r in add], hasMatcher=hm, matcher=W8.r_matcher[add])
    M.rmask[add] = True
    W8.r_hasMatcher = W8.r_hasMatcher.copy(); W8.r_hasMatcher[add] = hm
    old = np.nonzero(M.rmask & ~np.isin(np.arange(W8.nR), add))[0]
    upd = rng.choice(old, 40, replace=False)
    W8.r_matcher = W8.r_matcher.copy(); W8.r_matcher[upd] += rng.normal(0, 0.2, (len(upd), 2))
    W8.r_hasMatcher[upd] = 1
    # ---- the frames as the host loop leaves them, commit ----
    old_t = {f: t for t, f in enumerate(snap["frames"])}
    st_rows = np.array([snap["st"][old_t[f]] if f in old_t else W8.state[f] for f in M.frames])
    G.updateFrames(W8.evalPT[M.frames], st_rows, W8.state_zero[M.frames], np.ones(len(M.frames), np.float32))
    # (updates go by the CURRENT target index: after the removal / insertion above)
    tb = snap["tb"]
    ust = [int(tb["state"][old_t[int(W8.r_target[r])], snap["order"].index(big2id[int(W8.r_point[r])])]) for r in upd]
    G.insertResiduals([big2id[int(W8.r_point[r])] for r in upd], [M.frames.index(int(W8.r_target[r])) for r in upd], state=np.array(ust, np.int32),
                      hasMatcher=np.ones(len(upd), np.uint8), matcher=W8.r_matcher[upd], update=True)
    order = G.makeIDX()
    M.order = [int(i) for i in order]
    assert G.nF == len(M.frames) and G.nP == len(alive) + n_new
    G.setAdjointsF(); G.setPrecalcValues()
    HM, bM = G.marg_prior()
    n0 = 4 + 6 * (nF_b - 1)
    assert np.array_equal(HM[:n0, :n0], HM_ref) and np.array_equal(bM[:n0], bM_ref) and not HM[n0:].any() and not HM[:, n0:].any() and not bM[n0:].any()
    # what marginalizePointsF on the host would have left on the new frame's block by the time the frame itself is marginalised (a frame whose block
    # of HM is still zero cannot be eliminated: the reference inverts that block, EnergyFunctional.cpp:477-480) -- installed like a host loop does
    Q = rng.normal(0, 1, (6, 6))
    HM[n0:, n0:] += 50.0 * (Q @ Q.T) / 6 + 10.0 * np.eye(6)
    bM[n0:] += rng.normal(0, 5, 6)
    G.set_marg_prior(HM, bM)
    A = M.reload_window(snap, M.order, th_new, HM, bM, st_rows, W8.evalPT[M.frames], W8.state_zero[M.frames])
    R = api.EnergyFunctional(W8.w, W8.h, max_points=W8.nP).load(A, raw_images=not (seed % 2))
    return R


def test_keyframe_updates_equal_full_reloads(api, big):
    """Six key-frame steps in a row on one resident window (frames 0-4 of a 7-frame synthetic window; each step marginalises a frame,
    removes points, drops residuals, inserts the next frame with its points and residuals, moves some matchers; from the second step on the new
    points get ids that earlier points gave back, and frames come back into image slots others have left) -- after every commit a fresh
    handle loaded with the same graph gives the same tables, optimize trace, states, point sums, next solve and optimize tail, bit for bit."""
    import copy
    W8 = copy.copy(big)
    rng = np.random.default_rng(3)
    frames = [0, 1, 2, 3, 4]
    pts = np.nonzero(np.isin(W8.host, frames) & (rng.random(W8.nP) < 0.8))[0]
    M = Mirror(api, W8, frames, pts, rng.random(W8.nR) < 0.9, seed=5)
    t0 = M.G.optimize(4)
    assert (t0[:, 2] == 1).any()
    for step, (drop, new) in enumerate([(1, 5), (0, 6), (3, 1), (2, 0), (5, 3), (4, 2)]):
        R = keyframe_step(M, api, drop, new, n_remove=60, n_new=150, seed=step + 1)
        removed = check_equal(M.G, R)[3]
        # linearizeAll(true)'s toRemove list took residuals out of the window (FullSystemOptimize.cpp:136-155): the mirror follows
        ex = M.G.residual_table()["exists"]
        idx_of_big = {M.id2big[i]: k for k, i in enumerate(M.order)}
        t_of_big = {f: t for t, f in enumerate(M.frames)}
        gone = 0
        for r in np.nonzero(M.rmask)[0]:
            if not ex[t_of_big[int(W8.r_target[r])], idx_of_big[int(W8.r_point[r])]]:
                M.rmask[r] = False
                gone += 1
        assert removed.shape == ex.shape and gone == int(removed.sum())
        del R


def test_edit_entry_points_reject_bad_arguments(api, big):
    from sdv_loam_amd import synthetic as syn
    S = syn.subwindow(big, [0, 1, 2], np.nonzero(np.isin(big.host, [0, 1, 2]))[0][::3])
    G = api.EnergyFunctional(big.w, big.h, max_points=big.nP).load(S)
    L, h = G.L, G.h_
    i32 = lambda *a: np.array(a, np.int32)
    assert L.sdvgn_ef_remove_points(h, 1, i32(S.nP + 5)) < 0                     # no such id
    assert L.sdvgn_ef_drop_residuals(h, 1, i32(0), i32(7)) < 0                   # no such frame
    assert L.sdvgn_ef_drop_residuals(h, 1, i32(0), i32(int(S.host[0]))) < 0      # a point has no residual towards its own host
    assert L.sdvgn_ef_remove_frame(h, 9, None, None) < 0
    G.removePoints(i32(0))
    assert L.sdvgn_ef_remove_points(h, 1, i32(0)) < 0                            # already gone
    # until the commit the handle is the window of the last commit
    assert G.nP == S.nP and len(G.optimize(2)) >= 1
    order = G.makeIDX()
    assert len(order) == S.nP - 1 and 0 not in order
    # EFFrame::points order: the last point of the host took the place of the removed one
    last_of_host0 = np.nonzero(S.host == 0)[0][-1]
    assert order[0] == last_of_host0


def test_edits_that_cancel_each_other_and_reissued_ids(api, big):
    """Residual edits are resolved ON THE DEVICE at the commit (point id -> dense index, image slot -> frame index): an edit whose point or frame
    leaves before the commit must vanish -- also when the frame's image slot goes to a frame inserted in the same session -- and a point id is
    re-issued only after the commit that follows its removal.  Handle A records edits that cancel, handle B never makes them: same window."""
    from sdv_loam_amd import synthetic as syn
    frames = [0, 1, 2, 3]
    pts = np.nonzero(np.isin(big.host, frames))[0][::2]
    rmask = np.ones(big.nR, bool)
    n = 4 + 6 * len(frames)
    Q = np.random.default_rng(9).normal(0, 1, (n, n))
    S = syn.subwindow(big, frames, pts, rmask, HM=1e2 * (Q @ Q.T) / n, bM=np.zeros(n))
    new_f = 5
    th_new = np.float32(300.0)

    def edited(cancelling):
        G = api.EnergyFunctional(big.w, big.h, max_points=big.nP).load(S)
        assert len(G.optimize(2)) >= 1
        hosted1 = np.nonzero(S.host == 1)[0]
        others = np.nonzero(S.host != 1)[0]
        victims = others[:7].astype(np.int32)            # points removed in this session (ids = dense indices of the load)
        keep = others[7:40].astype(np.int32)
        if cancelling:
            # edits towards frame 1 (it leaves below) and on points that leave below
            G.dropResiduals(keep, np.full(len(keep), 1))
            G.insertResiduals(keep, np.full(len(keep), 1), hasMatcher=np.ones(len(keep), np.uint8), matcher=np.full((len(keep), 2), 7.5), update=True)
            tv = np.array([0 if S.host[v] != 0 else 2 for v in victims])
            G.dropResiduals(victims, tv)
        G.removePoints(np.concatenate([hosted1, victims]).astype(np.int32))
        G.removeFrame(1)
        k = G.insertFrame(big.evalPT[new_f], big.state[new_f], big.state_zero[new_f], int(big.frameID[new_f]), 1.0, th_new, dI=big.pyr0[new_f])
        assert k == 3
        newp = np.nonzero(big.host == new_f)[0][:25]
        ids = G.insertPoints(np.full(len(newp), k), big.u[newp], big.v[newp], big.idepth[newp], big.idepth_zero[newp], big.color[newp], big.weights[newp],
                             big.hasDepthPrior[newp], big.isFromSensor[newp])
        retired = set(int(i) for i in np.concatenate([hosted1, victims]))
        assert not (set(int(i) for i in ids) & retired)                       # not before the commit
        tgt = np.repeat(np.arange(3), len(ids))
        G.insertResiduals(np.tile(ids, 3), tgt, hasMatcher=np.zeros(len(tgt), np.uint8), matcher=np.zeros((len(tgt), 2)))
        order = G.makeIDX()
        G.setAdjointsF(); G.setPrecalcValues()
        return G, [int(i) for i in order], retired

    A, oa, retired = edited(True)
    B, ob, _ = edited(False)
    assert oa == ob
    ta, tb = A.residual_table(), B.residual_table()
    for k in ta:
        assert np.array_equal(ta[k], tb[k]), k
    assert np.array_equal(A.points(), B.points())
    assert np.array_equal(A.optimize(3), B.optimize(3), equal_nan=True)
    for a, b in zip(A.state(), B.state()):
        assert np.array_equal(a, b, equal_nan=True)
    # the next session may hand the retired ids out again -- and an edit on a re-issued id means the new point
    newp = np.nonzero(big.host == new_f)[0][25:31]
    for G in (A, B):
        ids2 = G.insertPoints(np.full(len(newp), 3), big.u[newp], big.v[newp], big.idepth[newp], big.idepth_zero[newp], big.color[newp], big.weights[newp],
                              big.hasDepthPrior[newp], big.isFromSensor[newp])
        assert set(int(i) for i in ids2) <= retired
        G.insertResiduals(ids2, np.zeros(len(ids2), np.int32), hasMatcher=np.zeros(len(ids2), np.uint8), matcher=np.zeros((len(ids2), 2)))
        order = [int(i) for i in G.makeIDX()]
        ex = G.residual_table()["exists"]
        for i in ids2:
            assert ex[0, order.index(int(i))] and not ex[1:, order.index(int(i))].any()


def test_ids_stay_bounded_over_many_keyframes(api, big):
    """40 commits in a row, each removing 40 points and inserting 40 (with residuals towards two frames): the ids handed out never exceed the
    points alive + what one commit retires (they are re-issued after the next commit), and the tables follow: every inserted residual exists
    at its point's current dense index, nothing else appeared."""
    from sdv_loam_amd import synthetic as syn
    frames = [0, 1, 2]
    pts = np.nonzero(np.isin(big.host, frames))[0]
    S = syn.subwindow(big, frames, pts[::2])
    G = api.EnergyFunctional(big.w, big.h, max_points=big.nP).load(S)
    rng = np.random.default_rng(4)
    alive = list(range(S.nP))                                                  # a window loaded through the setters: id = dense index
    host_of = {i: int(S.host[i]) for i in alive}
    res = {(int(S.r_point[r]), int(S.r_target[r])) for r in range(S.nR)}       # (id, target)
    pool = pts[1::2]
    n0, top = S.nP, S.nP - 1
    for step in range(40):
        gone = [int(i) for i in rng.choice(alive, 40, replace=False)]
        G.removePoints(np.array(gone, np.int32))
        alive = [i for i in alive if i not in set(gone)]
        res = {(i, t) for (i, t) in res if i not in set(gone)}
        newp = rng.choice(pool, 40, replace=False)
        hosts = np.array([frames.index(int(big.host[p])) for p in newp])
        ids = G.insertPoints(hosts, big.u[newp], big.v[newp], big.idepth[newp], big.idepth_zero[newp], big.color[newp], big.weights[newp],
                             big.hasDepthPrior[newp], big.isFromSensor[newp])
        ids = [int(i) for i in ids]
        assert not (set(ids) & set(gone)) and not (set(ids) & set(alive))
        top = max(top, max(ids))
        for i, h in zip(ids, hosts):
            host_of[i] = int(h)
        alive += ids
        tg = np.array([(host_of[i] + 1 + (step & 1)) % 3 for i in ids])
        G.insertResiduals(np.array(ids, np.int32), tg, hasMatcher=np.zeros(len(ids), np.uint8), matcher=np.zeros((len(ids), 2)))
        res |= {(i, int(t)) for i, t in zip(ids, tg)}
        order = [int(i) for i in G.makeIDX()]
        assert sorted(order) == sorted(alive)
        ex = G.residual_table()["exists"]
        idx = {i: k for k, i in enumerate(order)}
        want = np.zeros_like(ex)
        for (i, t) in res:
            want[t, idx[i]] = 1
        assert np.array_equal(ex, want), step
    assert top < n0 + 2 * 40                                                   # 40 live new + 40 retired and not yet re-issued, never more


def test_point_table_reload_without_set_frames_ends_the_resident_window(api, big):
    """after a commit the handle addresses residuals by slot (`removed` of optimize_finish is nF * nP bytes).  A reload through
    sdvgn_ef_set_points + sdvgn_ef_set_residuals that skips sdvgn_ef_set_frames must end that mode: `removed` is a residual list again (nR bytes -- the
    old build copied nF * nP bytes into it), the stale id tables are gone, and the window behaves like a freshly loaded one.  With an edit session open
    the two setters refuse."""
    import copy
    import ctypes as C
    from sdv_loam_amd import synthetic as syn
    W8 = copy.copy(big)
    rng = np.random.default_rng(9)
    frames = [0, 1, 2, 3, 4]
    pts = np.nonzero(np.isin(W8.host, frames) & (rng.random(W8.nP) < 0.8))[0]
    M = Mirror(api, W8, frames, pts, rng.random(W8.nR) < 0.9, seed=6)
    M.G.optimize(2)
    R = keyframe_step(M, api, 1, 5, n_remove=40, n_new=100, seed=1)      # the window is resident and table-addressed now
    del R
    G = M.G
    assert G.table_mode
    # the same frames (5 of them), a graph loaded through the two setters alone
    S = syn.subwindow(W8, M.frames, np.nonzero(np.isin(W8.host, M.frames))[0][::3], rng.random(W8.nR) < 0.9)
    c = np.ascontiguousarray
    args_p = (S.nP, c(S.host, np.int32), c(S.u, np.float32), c(S.v, np.float32), c(S.idepth, np.float32), c(S.idepth_zero, np.float32),
              c(S.color, np.float32).reshape(-1), c(S.weights, np.float32).reshape(-1), c(S.hasDepthPrior, np.uint8), c(S.isFromSensor, np.uint8))
    args_r = (S.nR, c(S.r_point, np.int32), c(S.r_target, np.int32), c(S.r_state, np.int32), c(S.r_hasMatcher, np.uint8),
              c(S.r_matcher, np.float64).reshape(-1), c(S.r_isLinearized, np.uint8), c(S.r_isActive, np.uint8))
    # (an open edit session: refused)
    G.removePoints(np.array([M.order[0]], np.int32))
    assert G.L.sdvgn_ef_set_points(G.h_, *args_p) == -10002 and G.L.sdvgn_ef_set_residuals(G.h_, *args_r) == -10002      # SDVGN_E_STATE
    G.makeIDX()
    assert G.L.sdvgn_ef_set_points(G.h_, *args_p) == 0
    assert G.L.sdvgn_ef_set_residuals(G.h_, *args_r) == 0
    G.table_mode = False
    G.nP, G.nR = S.nP, S.nR
    G.setAdjointsF(); G.setPrecalcValues()                 # (the commit above invalidated both, like every makeIDX)
    G.make_resident()
    tr = G.optimize(3)
    # a residual list of nR entries, guarded: the bytes behind it stay untouched
    e = C.c_double(0)
    rb, ng = np.zeros(S.nP, np.float32), np.zeros(S.nP, np.int32)
    rm = np.full(S.nR + 4096, 0xAB, np.uint8)
    assert G.L.sdvgn_ef_optimize_finish(G.h_, C.byref(e), rb.ctypes.data_as(C.c_void_p), ng.ctypes.data_as(C.c_void_p), rm.ctypes.data_as(C.c_void_p)) == 0
    assert (rm[S.nR:] == 0xAB).all() and set(np.unique(rm[:S.nR])) <= {0, 1}
    # and it is the window a fresh handle holds after the same calls (same frames / states: taken over from the resident one)
    vs, st, idp = G.state()
    assert len(tr) >= 1 and np.isfinite(tr).all() and idp.shape == (S.nP,)


def test_clear_error_lets_a_resident_window_recover(api, big):
    """a commit that carries a residual with a fixed linearisation raises the handle's sticky error word (code 8) and every later compute
    call fails with SDVGN_E_STATE -- until now only a reload through sdvgn_ef_set_frames cleared it.  sdvgn_ef_clear_error returns the word and clears it; once the
    caller has removed the offending points the resident window optimises again."""
    import copy
    W8 = copy.copy(big)
    rng = np.random.default_rng(3)
    frames = [0, 1, 2, 3, 4]
    pts = np.nonzero(np.isin(W8.host, frames))[0]
    M = Mirror(api, W8, frames, pts, rng.random(W8.nR) < 0.9, seed=2)
    G = M.G
    assert G.clear_error() == 0                                  # nothing raised yet
    G.optimize(2)
    mask = (np.arange(G.nP) % 7 == 0).astype(np.uint8)           # every seventh point: fixLinearizationF, like flagPointsForRemoval's marginalised ones
    G.fixLinearization(mask)
    fixed = [M.order[i] for i in np.nonzero(mask)[0]]
    G.removePoints(np.array([M.order[1]], np.int32))             # an edit session that does NOT remove them: the commit carries fixed linearisations over
    order = [int(i) for i in G.makeIDX()]
    G.setAdjointsF(); G.setPrecalcValues()
    with pytest.raises(RuntimeError):
        G.optimize(2)
    with pytest.raises(RuntimeError):                            # sticky
        G.optimize(2)
    assert G.clear_error() & 8
    assert G.clear_error() == 0
    G.removePoints(np.array([i for i in fixed if i in order], np.int32))
    order2 = [int(i) for i in G.makeIDX()]
    assert not set(fixed) & set(order2)
    G.setAdjointsF(); G.setPrecalcValues()
    tr = G.optimize(3)
    assert len(tr) >= 1 and np.isfinite(tr).all()
    assert G.clear_error() == 0
