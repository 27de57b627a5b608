"""Batch and multi-GPU front end: the analysis of many independent signals.

The reference analyses one signal per PV object in a Python loop (pypevoc/PVAnalysis.py:213-264).
Independent signals are the natural unit to shard: PVBatch runs run_pv on B equal-length signals in
one device call (pvx_analyze / pvx_analyze_dev with nsig = B), and `shard_range` / `gather_results`
spread a batch over the ranks of a torch.distributed job (one process per GPU, RCCL over xGMI for
the single result gather -- there is no other exchange on this path).
"""
import ctypes

import numpy as np

from . import _lib
from .PVAnalysis import _Plan

FIELDS = ("f", "mag", "ph", "realph", "binno")


def shard_range(nitems, rank, world):
    """Contiguous block partition of `nitems` signals over `world` ranks (first ranks take the
    remainder).  Returns (start, stop)."""
    q, r = divmod(int(nitems), int(world))
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def frame_shard(nsamp, nfft, hop, rank, world):
    """Shard ONE long signal over `world` ranks by frame ranges (no exchange step).

    A frame needs the spectrum of the frame before it, so a rank that starts at frame f0 > 0 also
    analyses frame f0 - 1 (one frame of halo) and drops that row.  Returns (f0, f1, a, b, drop):
    this rank owns frames [f0, f1); it analyses samples x[a:b], whose first `drop` result rows are
    the halo.  The per-shard results are bit-identical to the corresponding rows of the unsharded
    analysis (the kernels' results do not depend on where a frame sits in a launch)."""
    F = _lib.nframes_host(nsamp, nfft, hop)
    f0, f1 = shard_range(F, rank, world)
    if f1 <= f0:
        return f0, f1, 0, 0, 0
    h0 = max(f0 - 1, 0)
    a = h0 * hop
    b = min(a + (f1 - h0 - 1) * hop + nfft + 1, nsamp)         # nframes(b - a) == f1 - h0
    return f0, f1, a, b, f0 - h0


def analyze_frame_shard(x, sr, nfft, hop, npks, rank, world, pkthresh=0.005, wind=np.hanning, precision=None):
    """This rank's rows of PV(x, sr, nfft, hop, npks, pkthresh).run_pv() under `frame_shard`:
    dict(f, mag, ph, realph, binno [n, npks], t, totalmag [n], f0, f1).  Concatenating the ranks' blocks
    in rank order (e.g. with `gather_results` on padded blocks) gives the unsharded arrays; the tracker
    (toSinSum) then runs once on the gathered (F, npks) arrays -- links only need adjacent rows.
    precision=None follows the samples as PV does (float64 samples: the reference's float64 arithmetic), so the shards of a
    signal agree with PV(x).run_pv() on the same data."""
    from .PVAnalysis import PV
    f0, f1, a, b, drop = frame_shard(len(x), nfft, hop, rank, world)
    out = dict(f0=f0, f1=f1)
    if f1 <= f0:
        for k in FIELDS:
            out[k] = np.zeros((0, npks))
        out["t"] = np.zeros(0)
        out["totalmag"] = np.zeros(0)
        return out
    p = PV(x[a:b], sr, nfft=nfft, hop=hop, npks=npks, pkthresh=pkthresh, wind=wind, progress=False, precision=precision)
    p.run_pv()
    assert p.nframes == f1 - f0 + drop
    for k in FIELDS:
        out[k] = getattr(p, k)[drop:]
    out["t"] = (np.arange(f0, f1) * hop + nfft / 2.0) / sr        # PVAnalysis.py:247 with the global position
    out["totalmag"] = np.asarray(p.totalmag)[drop:]
    return out


class PVBatch(object):
    """run_pv for a batch of equal-length signals `x[B, nsamp]` with shared parameters.

    Results: f, mag, ph, realph, binno with shape (B, F, npks); t (F,); totalmag (B, F)."""

    def __init__(self, x, sr, nfft=1024, hop=None, npks=20, pkthresh=0.005, wind=np.hanning,
                 precision=None):
        self._xdev = None
        if _lib.is_device_array(x):
            # a batch that already lives in GPU memory (e.g. a torch tensor on the GPU) is analysed in place
            self._xdev = _lib.DeviceSignal(x)
            if len(self._xdev.shape) != 2:
                raise ValueError("PVBatch expects x[B, nsamp]")
            self.x, self.x_dtype = x, self._xdev.dtype_code
            self.nsig, self.nsamp = self._xdev.shape
        else:
            x = np.asarray(x)
            if x.ndim != 2:
                raise ValueError("PVBatch expects x[B, nsamp]")
            self.x, self.x_dtype = _lib.as_signal(x)
            self.nsig, self.nsamp = self.x.shape
        self.sr = sr
        self.nfft = nfft
        self.nfft2 = int(nfft / 2)
        self.hop = int(self.nfft / 2) if hop is None else int(hop)
        self.npeaks = npks
        self.peakthresh = pkthresh
        self.win = wind(nfft)
        if precision is None:
            # as PV: float64 samples get the reference's float64 arithmetic, float32 / int16 samples the float32 transform
            precision = 64 if self.x_dtype == _lib.PVX_F64 else 32
        self.precision = precision
        self.nframes = 0
        self._plan = None

    def run_pv(self):
        lib = _lib.load()
        B, K = self.nsig, self.npeaks
        F = int(lib.pvx_nframes(self.nsamp, self.nfft, self.hop))
        self.nframes = F
        shape = (B, F, K)
        out = {k: np.zeros(shape) for k in FIELDS}
        t = np.zeros((B, F))
        tm = np.zeros((B, F))
        if F > 0 and B > 0:
            # a pooled plan (PVAnalysis._Plan.acquire: creating one, with its tables and buffers, costs more than a
            # batch of short signals); the results leave for the host inside the call, so the plan is handed back at once
            self._plan = _Plan.acquire(self, self.sr, self.nfft, self.hop, K, self.peakthresh, self.win, self.precision,
                                       max_rows=B * (F + 1))
            # (the pool is shared with PV: a plan last used by a PV with progress output still has that PV's callback)
            _lib.check(lib.pvx_plan_set_progress(self._plan.handle, _lib.PROGRESS_FN(), None), "pvx_plan_set_progress")
            self._plan.progress_owner = None
            if self._xdev is not None:
                n = B * F * K

                def launch(o, stream):
                    ptrs = [ctypes.c_void_p(o + i * n * 8) for i in range(5)]
                    r = lib.pvx_analyze_dev(self._plan.handle, ctypes.c_void_p(self._xdev.ptr), self.x_dtype, self.nsamp, B,
                                            self.nsamp, *ptrs, ctypes.c_void_p(o + 5 * n * 8),
                                            ctypes.c_void_p(o + 5 * n * 8 + B * F * 8), None, stream)
                    _lib.check(r, "pvx_analyze_dev")

                blk = _lib.device_run((5 * n + 2 * B * F) * 8, launch)
                for i, k in enumerate(FIELDS):
                    out[k] = blk[i * n:(i + 1) * n].reshape(shape).copy()
                t = blk[5 * n:5 * n + B * F].reshape(B, F).copy()
                tm = blk[5 * n + B * F:].reshape(B, F).copy()
            else:
                r = lib.pvx_analyze(self._plan.handle, self.x.ctypes.data_as(ctypes.c_void_p), self.x_dtype,
                                    self.nsamp, B, self.nsamp, *[_lib.dptr(out[k]) for k in FIELDS],
                                    _lib.dptr(t), _lib.dptr(tm), None, None)
                _lib.check(r, "pvx_analyze")
        if self._plan is not None:
            self._plan.owner = None
            self._plan = None
        for k in FIELDS:
            setattr(self, k, out[k])
        self.t = t[0] if B > 0 else np.zeros(F)
        self.totalmag = tm
        return self

    # (what _Plan.acquire asks of an owner whose plan it takes away: nothing is resident here)
    _progress_cb = None
    _progress_plan = None

    def _release_resident(self):
        pass


class PVMany(object):
    """run_pv for signals of ANY lengths over the GPUs of this process: the mirror of pvx_batch_* (include/pvx.h; SURVEY.md
    8(b) `pvx_analyze_batch`).  The reference's equivalent is `[PV(x, sr, ...).run_pv() for x in signals]`
    (pypevoc/PVAnalysis.py:213-264); here every device takes the next signal off one queue (longest first) and the
    devices never exchange anything.

        many = PVMany(sr, nfft=2048, hop=512, npks=8, devices=[0, 1, 2, 3])
        results = many.run(signals)          # list of dicts: f, mag, ph, realph, binno (F, npks); t, totalmag (F,); device

    Signals must share a sample type (float64, float32 or int16).  precision=None follows it as PV does."""

    def __init__(self, sr, nfft=1024, hop=None, npks=20, pkthresh=0.005, wind=np.hanning, precision=None, devices=None,
                 workers_per_device=0):
        self.sr, self.nfft = sr, int(nfft)
        self.hop = int(self.nfft / 2) if hop is None else int(hop)
        self.npeaks, self.peakthresh = int(npks), pkthresh
        self.win = np.ascontiguousarray(wind(self.nfft), dtype=np.float64)
        self.precision = precision
        self.devices = None if devices is None else [int(d) for d in devices]
        self.workers_per_device = int(workers_per_device)
        self._handle = None
        self._handle_precision = None

    def _batch(self, precision):
        lib = _lib.load()
        if self._handle is not None and self._handle_precision != precision:
            self.close()
        if self._handle is None:
            h = ctypes.c_void_p()
            dev = None
            nd = 0
            if self.devices:
                dev = (ctypes.c_int32 * len(self.devices))(*self.devices)
                nd = len(self.devices)
            _lib.check(lib.pvx_batch_create(ctypes.byref(h), float(self.sr), self.nfft, self.hop, self.npeaks, float(self.peakthresh),
                                            _lib.dptr(self.win), precision, dev, nd, self.workers_per_device), "pvx_batch_create")
            self._handle, self._handle_precision = h, precision
        return self._handle

    def run(self, signals):
        lib = _lib.load()
        sigs = []
        code = None
        for x in signals:
            a, c = _lib.as_signal(np.asarray(x).reshape(-1))
            if code is None:
                code = c
            elif c != code:
                raise TypeError("the signals of one PVMany.run call must share a sample type")
            sigs.append(a)
        if not sigs:
            return []
        precision = self.precision if self.precision is not None else (64 if code == _lib.PVX_F64 else 32)
        K = self.npeaks
        items = (_lib.BatchItem * len(sigs))()
        out = []
        for i, a in enumerate(sigs):
            F = _lib.nframes_host(len(a), self.nfft, self.hop)
            r = {k: np.zeros((F, K)) for k in FIELDS}
            r["t"] = np.zeros(F)
            r["totalmag"] = np.zeros(F)
            it = items[i]
            it.x, it.nsamp = a.ctypes.data, len(a)
            for k in FIELDS + ("t", "totalmag"):
                setattr(it, k, _lib.dptr(r[k]))
            out.append(r)
        total = lib.pvx_batch_run(self._batch(precision), code, ctypes.cast(items, ctypes.c_void_p), len(sigs))
        _lib.check(total, "pvx_batch_run")
        for i, r in enumerate(out):
            r["device"] = int(items[i].device)
            r["nframes"] = int(items[i].nframes)
        return out

    def close(self):
        if self._handle is not None:
            _lib.load().pvx_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_results(local, nitems, group=None, dst=0):
    """Gather per-rank result blocks to rank `dst` with ONE collective per call.

    `local` is a torch tensor [n_local, ...] (device tensor under the nccl/RCCL backend, CPU tensor
    under gloo) holding this rank's shard in `shard_range` order.  Shards may differ by one item, so
    every rank pads to the largest shard; rank `dst` returns the [nitems, ...] concatenation, the
    others None.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(nitems, r, world)[1] - shard_range(nitems, r, world)[0] for r in range(world)]
    nmax = max(sizes) if sizes else 0
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d items, expected %d" % (rank, local.shape[0], sizes[rank]))
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    pad = pad.contiguous()
    if world == 1:
        return pad[: sizes[0]]
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=bufs, dst=dst, group=group)
        return torch.cat([b[: sizes[r]] for r, b in enumerate(bufs)], dim=0)
    dist.gather(pad, gather_list=None, dst=dst, group=group)
    return None


class PipelinedGather(object):
    """Double-buffered result gather for a stream of analysis steps.

    Step i writes its result block into `buffer(i)`; `submit(i)` starts the gather of that block to
    rank `dst` asynchronously (RCCL runs it on its own stream), so the collective of step i overlaps
    the kernels of step i+1.  A buffer is handed out again only after the gather that read it has
    completed.  On rank `dst`, `consume(step, blocks)` -- if given -- is called once per step with the
    list of per-rank blocks as soon as they have arrived and before their receive buffers are reused
    (on a GPU it runs on a side stream, so unpacking step i also overlaps the kernels of step i+1);
    without it, `result(i)` returns those blocks (valid after `drain()` or after `buffer(i+depth)`).
    There is no other communication on this path.
    """

    def __init__(self, numel, dtype, device, group=None, dst=0, depth=2, consume=None, force=False, host_retire=False):
        import torch
        import torch.distributed as dist
        self._torch = torch
        self._dist = dist
        self.group = group
        self.dst = dst
        self.depth = depth
        self.consume = consume
        # host_retire: a slot is handed out again once the HOST has seen its gather complete (polling the work object) instead of
        # making the caller's stream wait for it.  A stream-level wait is a barrier packet in the compute stream's queue on every step:
        # measured on MI355X it costs a step of 0.10 ms about 0.015 ms whether or not the gather had long finished.  With a few slots
        # of depth the gather of step i - depth is always done by the time the host issues step i, so the poll never spins.
        self.host_retire = bool(host_retire) and consume is None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # world 1: nothing to gather, unless `force` asks for the collective anyway (single-GPU test
        # of the RCCL path)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.bufs = [torch.empty(numel, dtype=dtype, device=device) for _ in range(depth)]
        self.lists = None
        if self.active and self.rank == dst:
            self.lists = [[torch.empty(numel, dtype=dtype, device=device) for _ in range(self.world)]
                          for _ in range(depth)]
        self.works = [None] * depth
        self.steps = [None] * depth
        self.cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if (self.cuda and consume is not None and self.rank == dst) else None
        self.done = [None] * depth            # events: blocks of slot j consumed (side stream)

    def _retire(self, j):
        """The gather in slot j has been waited for (stream-level on a GPU) and consumed."""
        w = self.works[j]
        if w is None:
            return
        self.works[j] = None
        torch = self._torch
        if self.side is not None:
            with torch.cuda.stream(self.side):
                w.wait()                                   # the side stream waits for the collective
                self.consume(self.steps[j], self.lists[j])
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.done[j] = ev
            torch.cuda.current_stream().wait_event(ev)     # before the slot's buffers are written again
        elif self.host_retire and self.cuda:
            import time
            t_end = time.perf_counter() + 30.0
            while not w.is_completed():
                if time.perf_counter() > t_end:            # (never seen; a stuck collective must not hang the caller silently)
                    w.wait()
                    torch.cuda.current_stream().synchronize()
                    break
                time.sleep(0)
        else:
            w.wait()
            if self.consume is not None and self.lists is not None:
                self.consume(self.steps[j], self.lists[j])

    def buffer(self, step):
        j = step % self.depth
        self._retire(j)
        return self.bufs[j]

    def submit(self, step):
        if not self.active:
            return
        j = step % self.depth
        self.steps[j] = step
        self.works[j] = self._dist.gather(self.bufs[j], gather_list=self.lists[j] if self.lists else None,
                                          dst=self.dst, group=self.group, async_op=True)

    def drain(self):
        order = sorted(range(self.depth), key=lambda j: (self.steps[j] is None, self.steps[j] or 0))
        for j in order:
            self._retire(j)

    def result(self, step):
        j = step % self.depth
        if not self.active:
            return [self.bufs[j]]
        return self.lists[j] if self.lists else None


class ResultWire(object):
    """Device-side packing of a shard's result rows into the compact wire format of include/pvx.h
    (pvx_pack_rows_dev / pvx_unpack_rows_dev): 18 B per peak slot instead of 40 B at precision 32 --
    14 B in the plan's wire format 2 (`wire_format=2`: pvx_plan_set_wire_format, precision-32 plans) --,
    bit-exact round trip.  `plan` is a pvx plan handle, `rows` the frames of all signals of a shard;
    every rank of a gather must use the same format."""

    def __init__(self, plan, rows, npks, wire_format=None):
        self.lib = _lib.load()
        self.plan = plan
        if wire_format is not None:
            _lib.check(self.lib.pvx_plan_set_wire_format(plan, int(wire_format)), "pvx_plan_set_wire_format")
        self.wire_format = int(self.lib.pvx_plan_get_wire_format(plan))
        self.rows = int(rows)
        self.npks = int(npks)
        self.nbytes = int(self.lib.pvx_wire_bytes(plan, self.rows))
        if self.nbytes < 0:
            _lib.check(self.nbytes, "pvx_wire_bytes")

    def result_ptrs(self, base):
        """Device pointers (f, mag, ph, realph, binno, totalmag) inside a float64 block of
        `result_numel()` elements starting at address `base`."""
        n = self.rows * self.npks
        return [base + i * n * 8 for i in range(5)] + [base + 5 * n * 8]

    def result_numel(self):
        return 5 * self.rows * self.npks + self.rows

    def pack(self, res_base, wire_ptr, stream=None):
        f, mag, ph, _, binno, tm = self.result_ptrs(res_base)
        _lib.check(self.lib.pvx_pack_rows_dev(self.plan, self.rows, f, mag, ph, binno, tm, wire_ptr, stream),
                   "pvx_pack_rows_dev")

    def unpack(self, wire_ptr, res_base, stream=None):
        f, mag, ph, realph, binno, tm = self.result_ptrs(res_base)
        _lib.check(self.lib.pvx_unpack_rows_dev(self.plan, self.rows, wire_ptr, f, mag, ph, realph, binno, tm, stream),
                   "pvx_unpack_rows_dev")
