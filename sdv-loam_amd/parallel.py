"""Multi-GPU plumbing for the sliding-window back end (BASELINE.json configs[3]).

One process per GPU.  Key-frames are sharded by HOST frame: rank r linearises / accumulates only the residuals whose
host frame lies in its range (target images are replicated on every rank).  Per Gauss-Newton iteration the ranks sum
ONE packed fp64 message -- accumulators (top Gram nF^2 x 121 | Schur Gram nF x 1431 | resInA: 154 kB at nF = 8) | the 4 energy / step
statistics of the trial linearisation | the candidates of setNewFrameEnergyTH's quantile (nP) -- with a single all-reduce (RCCL over xGMI
through the library's own ncclAllReduce or torch.distributed's "nccl" backend; latency-bound at this size): sdvgn_ef_optimize applies and
accumulates the trial speculatively, so that the energy the accept test needs and the accumulators the next solve needs travel together
(one_collective=False keeps the earlier two-collective loop for comparison).  The small solve then runs redundantly on every rank on
bitwise-identical inputs, so all ranks take the same accept / reject decisions without further communication.
"""
import ctypes as C

import numpy as np

TOP_E = 121          # live 11x11 of the (host,target) Gram, row-major
SC_N = 53            # live features of the Schur Gram: 6 x 8 JpJdF | 4 Hcd | bdSum
SC_E = SC_N * (SC_N + 1) // 2   # its upper triangle, row-major (1431)
MAX_FRAMES = 8


_DEVICE_STREAMS = {}   # device index -> the torch stream every ShardedEnergyFunctional of this process issues its work on
_GROUP_IDS = []   # (process group, 128-byte RCCL id) of the communicators this process created through sdvgn_ef_init_rccl


def direct_rccl_disabled():
    import os
    return os.environ.get("SDVGN_NO_DIRECT_RCCL") == "1"


def shard_hosts(nF, world):
    """Contiguous host-frame ranges, as even as possible: rank r owns [lo[r], hi[r])."""
    base, extra = divmod(nF, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((lo, lo + n))
        lo += n
    return out


def acc_count(nF):
    return nF * nF * TOP_E + nF * SC_E + 1


def acc_capacity():
    return acc_count(MAX_FRAMES)


def pack_accumulators(nF, top13, accE, accEB, accD, Hcc, bc, res_in_A):
    """Pack reference-layout accumulators (oracle getters) into the library's buffer layout -- used by the CPU tests.
    top13: [nF*nF][13][13] indexed h + nF*t ; accE [nF*nF][8][4], accEB [nF*nF][8], accD [nF^3][8][8] indexed
    (h + nF*t1) + nF^2*t2."""
    buf = np.zeros(acc_count(nF))
    top = buf[:nF * nF * TOP_E].reshape(nF, nF, 11, 11)                  # device pair index h*nF + t
    idx = list(range(10)) + [12]
    for h in range(nF):
        for t in range(nF):
            top[h, t] = top13[h + nF * t][np.ix_(idx, idx)]
    sc = buf[nF * nF * TOP_E: nF * nF * TOP_E + nF * SC_E].reshape(nF, SC_E)
    iu = np.triu_indices(SC_N)
    for h in range(nF):
        G = np.zeros((64, 64))
        for t1 in range(nF):
            for t2 in range(nF):
                G[6 * t1:6 * t1 + 6, 6 * t2:6 * t2 + 6] = accD[(h + nF * t1) + nF * nF * t2][:6, :6]
            G[6 * t1:6 * t1 + 6, 48:52] = accE[h + nF * t1][:6, :]
            G[6 * t1:6 * t1 + 6, 52] = accEB[h + nF * t1][:6]
        if h == 0:   # accHcc / accbc are window totals in the reference; the stitch only uses their sum over hosts
            G[48:52, 48:52] = Hcc
            G[48:52, 52] = bc
        sc[h] = G[:SC_N, :SC_N][iu]
    buf[-1] = res_in_A
    return buf


class ShardedEnergyFunctional:
    """EnergyFunctional on this rank's GPU restricted to its host-frame shard, wired to torch.distributed.

    All library work and the collectives are issued on one torch stream (`self.stream`), so the all-reduce is ordered
    after the accumulate kernels and before the read-back of the packed buffer without host synchronisation."""

    def __init__(self, W, rank, world, device, group=None, force_collective=False, one_collective=True):
        import torch
        import torch.distributed as dist
        from .backend_api import EnergyFunctional
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = rank, world
        torch.cuda.set_device(device)
        # one stream per device for all sharded windows of the process: a stream of its own per window means a hardware queue per window, and
        # the first launch on a queue that has been idle since its creation costs 0.2-0.5 ms
        if device not in _DEVICE_STREAMS:
            _DEVICE_STREAMS[device] = torch.cuda.Stream(device=device)
        self.stream = _DEVICE_STREAMS[device]
        self.ef = EnergyFunctional(W.w, W.h, max_points=W.nP, device=device, stream=self.stream.cuda_stream)
        L = self.ef.L
        self._fenced = []      # SDVGN_FENCE_EXTERNAL=1 (test rigs): the caller-owned buffers come from the library's fenced / poisoned allocator
        with torch.cuda.stream(self.stream):
            self.acc = self._zeros(acc_capacity())
            self.stats = self._zeros(4 + W.nP)   # 4 statistics + the quantile candidates
        self.stream.synchronize()
        self.ef._check(L.sdvgn_ef_set_external_buffers(self.ef.h_, self.acc.data_ptr(), self.acc.numel(), self.stats.data_ptr(), self.stats.numel()))
        # ONE collective per loop body (BASELINE.json north_star): two message buffers [accumulators | 4 statistics | quantile candidates];
        # sdvgn_ef_optimize applies + accumulates the trial speculatively and all-reduces one message per body (+ one per call)
        self.one_collective = bool(one_collective)
        self.coll = None
        if self.one_collective:
            L.sdvgn_ef_collective_stride.argtypes = [C.c_void_p]
            L.sdvgn_ef_set_collective_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            stride = L.sdvgn_ef_collective_stride(self.ef.h_)
            with torch.cuda.stream(self.stream):
                self.coll = self._zeros(2 * stride)
            self.stream.synchronize()
            self.ef._check(L.sdvgn_ef_set_collective_buffer(self.ef.h_, self.coll.data_ptr(), self.coll.numel()))
        self.max_points = W.nP
        self.lo, self.hi = shard_hosts(W.nF, world)[rank]
        self.ef.set_host_range(self.lo, self.hi)
        self.ef.load(W)
        self._cb = None
        self.n_allreduce = 0
        self.direct_rccl = False
        if (world > 1 or force_collective) and dist.get_backend(self.group) == "nccl" and not direct_rccl_disabled():
            # preferred: the library issues ncclAllReduce itself on its stream (no callback / Python in the iteration).  Rank 0's id
            # travels through the existing process group; any failure falls back to the callback path below.
            # The communicator is shared by every sharded window of this process and group (the library reference-counts it by id): only
            # the first window pays ncclCommInitRank.  An id may be reused only while its communicator is alive on EVERY rank -- the ranks
            # agree on that with one MIN all-reduce (object lifetimes, hence the answer, could differ from rank to rank).
            cached = next((i for g, i in _GROUP_IDS if g is self.group), None)
            alive = torch.tensor([1 if (cached is not None and L.sdvgn_rccl_comm_alive(cached) == 1) else 0], device="cuda")
            dist.all_reduce(alive, op=dist.ReduceOp.MIN, group=self.group)
            ident = [cached if alive.item() == 1 else None]
            if ident[0] is None:
                if rank == 0:
                    buf = (C.c_ubyte * 128)()
                    if L.sdvgn_rccl_unique_id(buf) == 0:
                        ident = [bytes(buf)]
                dist.broadcast_object_list(ident, src=0, group=self.group)  # always executed by every rank (None = no RCCL: all fall back)
                _GROUP_IDS[:] = [(g, i) for g, i in _GROUP_IDS if g is not self.group][-3:] + [(self.group, ident[0])]
            if ident[0] is not None:
                idbuf = (C.c_ubyte * 128).from_buffer_copy(ident[0])
                self.direct_rccl = L.sdvgn_ef_init_rccl(self.ef.h_, idbuf, rank, world) == 0   # collective the first time (ncclCommInitRank)
                if world > 1:   # agree on the outcome: one failing rank sends everybody to the callback path
                    okt = torch.tensor([1 if self.direct_rccl else 0], device="cuda")
                    dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=self.group)
                    self.direct_rccl = bool(okt.item())
                if not self.direct_rccl:
                    L.sdvgn_ef_init_rccl(self.ef.h_, None, 0, 0)          # drop a communicator this rank may have got
        if (world > 1 or force_collective) and not self.direct_rccl:
            acc_ptr, stats_ptr = self.acc.data_ptr(), self.stats.data_ptr()

            via_host = dist.get_backend(self.group) == "gloo"   # test rigs without RCCL peers (e.g. two ranks sharing one GPU)

            coll_ptr = self.coll.data_ptr() if self.coll is not None else 0
            coll_n = self.coll.numel() if self.coll is not None else 0

            def _allreduce(user, buf, count):
                if coll_n and coll_ptr <= buf < coll_ptr + 8 * coll_n:           # a message buffer of the one-collective loop
                    off = (buf - coll_ptr) // 8
                    t = self.coll[off:off + count]
                else:
                    assert buf in (acc_ptr, stats_ptr)
                    t = self.acc[:count] if buf == acc_ptr else self.stats[:count]
                with torch.cuda.stream(self.stream):
                    if via_host:
                        h = t.cpu()                      # synchronises this stream: the accumulate kernels are done
                        dist.all_reduce(h, group=self.group)
                        t.copy_(h)
                    else:
                        dist.all_reduce(t, group=self.group)
                self.n_allreduce += 1

            self._cb = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)(_allreduce)
            self.ef._check(L.sdvgn_ef_set_allreduce(self.ef.h_, C.cast(self._cb, C.c_void_p), None))

    def _zeros(self, n):
        """n zeroed doubles on the device: a torch tensor -- or, with SDVGN_FENCE_EXTERNAL=1, a torch view of a buffer from the library's
        debugging allocator (csrc/devmem.hpp), so that what the caller hands to sdvgn_ef_set_external_buffers / _set_collective_buffer
        sits behind the same fence / poison instruments as the library's own allocations"""
        import os
        torch = self.torch
        if os.environ.get("SDVGN_FENCE_EXTERNAL") != "1":
            return torch.zeros(n, dtype=torch.float64, device="cuda")
        L = self.ef.L
        L.sdvgn_debug_dmalloc.restype = C.c_void_p
        L.sdvgn_debug_dmalloc.argtypes = [C.c_size_t]
        ptr = L.sdvgn_debug_dmalloc(8 * n)
        if not ptr:
            raise MemoryError("sdvgn_debug_dmalloc(%d)" % (8 * n))
        self._fenced.append(ptr)      # (never given back: a test rig's buffers live as long as the process, so no use-after-free of OURS can hide a library one)

        class _Raw:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        t = torch.as_tensor(_Raw(), device="cuda")
        t.zero_()
        return t

    def reload(self, W):
        if W.nP > self.max_points:
            raise ValueError("window has %d points, the handle was created for %d" % (W.nP, self.max_points))
        # the shard follows the window: a different nF re-partitions the host frames
        self.lo, self.hi = shard_hosts(W.nF, self.world)[self.rank]
        self.ef.set_host_range(self.lo, self.hi)
        self.ef.load(W)

    def optimize(self, its, fixed_its=False, want_trace=False):
        with self.torch.cuda.stream(self.stream):
            return self.ef.optimize(its, want_trace=want_trace, fixed_its=fixed_its)

    def collective_count(self):
        """all-reduces this window has issued so far, counted where they are issued (the library for RCCL / its callback, sdvgn_ef_collective_count)"""
        self.ef.L.sdvgn_ef_collective_count.restype = C.c_ulonglong
        self.ef.L.sdvgn_ef_collective_count.argtypes = [C.c_void_p]
        return int(self.ef.L.sdvgn_ef_collective_count(self.ef.h_))


# ---------------------------------------------------------------------------------------------------------------------------------
# Coarse tracker, hypothesis-parallel (tracker row (i)).
#
# FullSystem::trackNewCoarse (FullSystem.cpp:341-470) tries up to ~31 initial poses one after the other (constant motion, half / double
# / zero motion, 26 small rotations) and keeps the one with the smallest level-0 residual.  The tries are independent Levenberg-Marquardt
# runs against the same reference template and the same new frame, so they shard: rank r runs the tries r, r + world, r + 2 world, ...
# (interleaved: the likely winners at the head of the list land on different GPUs) as ONE device-resident batch (k_track, one workgroup
# per try), ONE all-reduce carries every try's 18 result doubles (ok, lastResiduals[5], pose 7, affine 2, flow 3) to every rank, and
# every rank replays the reference's selection loop over the complete table -- same winner on all ranks, no broadcast.
# Difference to the sequential reference (stated in SURVEY.md 8e): every try runs to the end; the reference passes the best residuals so
# far as `minResForAbort` into later tries (a later try that is 1.5x worse on a coarse level is cut short and cannot win) and stops
# trying once a result is below setting_reTrackThreshold x the last frame's RMSE.  The replay keeps both rules on the finished results
# (a try whose coarse-level residuals would have triggered the abort is not eligible and contributes only the residuals of the levels
# it would have finished; tries behind the early-out are ignored), so the winner and the achieved residuals are the reference's
# whenever the per-level residuals of a completed run equal those of the run the reference would have cut.
# ---------------------------------------------------------------------------------------------------------------------------------
HYP_COLS = 18   # ok | lastResiduals[5] | pose7 | aff2 | flow3


def hypothesis_slice(n, rank, world):
    """indices of the tries rank `rank` evaluates"""
    return list(range(rank, n, world))


def select_hypothesis(table, last_coarse_rmse0=None, retrack_threshold=1.5, abort_factor=1.5, coarsest=4):
    """The selection loop of FullSystem::trackNewCoarse (FullSystem.cpp:412-463) replayed on the finished results `table`
    [n][HYP_COLS] in try order.  Returns dict(good, index, pose, aff, flow, achieved_res, tries).

    Every try in the table ran to the end.  In the reference a later try receives the residuals achieved so far as `minResForAbort` and
    is cut on the first level L (coarsest first) whose residual exceeds 1.5 x the achieved one (CoarseTracker.cpp:808-810): it returns
    false with lastResiduals[l < L] still NaN (:674), so it can neither win nor lower the achieved residuals of the finer levels nor
    trigger the early-out with them.  The replay reproduces exactly that view of the finished run."""
    n = table.shape[0]
    achieved = np.full(5, np.nan)
    good, win = False, -1
    tries = 0
    for i in range(n):
        ok = table[i, 0] > 0.5
        res = table[i, 1:6].copy()
        tries += 1
        for l in range(min(coarsest, 4), -1, -1):                        # levels in the order the tracker visits them
            if np.isfinite(res[l]) and res[l] > abort_factor * achieved[l]:          # (NaN achieved: comparison false, no abort)
                ok = False
                res[:l] = np.nan
                break
        if ok and np.isfinite(np.float32(res[0])) and not (res[0] >= achieved[0]):
            good, win = True, i
        if good:
            for l in range(5):
                if not np.isfinite(np.float32(achieved[l])) or achieved[l] > res[l]:
                    achieved[l] = res[l]
        if good and last_coarse_rmse0 is not None and achieved[0] < last_coarse_rmse0 * retrack_threshold:
            break
    if not good:
        return dict(good=False, index=0, pose=None, aff=None, flow=np.zeros(3), achieved_res=achieved, tries=tries)
    return dict(good=True, index=win, pose=table[win, 6:13].copy(), aff=table[win, 13:15].copy(), flow=table[win, 15:18].copy(),
                achieved_res=achieved, tries=tries)


def track_hypotheses(evaluate, poses7, aff, coarsest, rank=0, world=1, group=None, last_coarse_rmse0=None, retrack_threshold=1.5):
    """Shard the tries of trackNewCoarse over the ranks.  `evaluate(poses[B,7], affs[B,2], coarsest) -> (ok[B], poses[B,7], affs[B,2],
    lastResiduals[B,5], flow[B,3])` is CoarseTracker.trackBatch of this rank's tracker (every rank holds the same reference template and
    new frame).  One all-reduce(sum) of an [n][18] fp64 table in which every rank fills only its rows."""
    poses7 = np.asarray(poses7, np.float64).reshape(-1, 7)
    n = poses7.shape[0]
    mine = hypothesis_slice(n, rank, world)
    table = np.zeros((n, HYP_COLS))
    if mine:
        ok, p, a, lr, fl = evaluate(poses7[mine], np.tile(np.asarray(aff, np.float64), (len(mine), 1)), coarsest)
        table[mine, 0] = np.asarray(ok, np.float64)
        table[mine, 1:6] = np.nan_to_num(lr, nan=1e300, posinf=1e300)     # a NaN row would poison the sum; 1e300 loses every comparison
        table[mine, 6:13] = p
        table[mine, 13:15] = a
        table[mine, 15:18] = fl
    if world > 1:
        import torch
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            t = torch.from_numpy(table).cuda()
            dist.all_reduce(t, group=group)
            table = t.cpu().numpy()
        else:
            t = torch.from_numpy(table)
            dist.all_reduce(t, group=group)
            table = t.numpy()
    table[:, 1:6] = np.where(table[:, 1:6] >= 1e299, np.nan, table[:, 1:6])
    return select_hypothesis(table, last_coarse_rmse0, retrack_threshold, coarsest=coarsest), table
