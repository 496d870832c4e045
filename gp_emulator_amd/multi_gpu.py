"""Row-sharded predict over the GPUs of one node, and the rank plumbing bench.py uses.

Every output row depends only on its own test row plus read-only per-emulator constants
(gp_emulator/GaussianProcess.py:232-247; "all inputs can be divided and distributed
arbitrarily", doc/report.md:95), so the path shards into independent contiguous row blocks:
GPU g gets rows [g*ceil(M/G), min(M, (g+1)*ceil(M/G))), constants (< 1 MB) are replicated,
and each device writes its own disjoint slice of the caller's arrays.  There is NO
collective on the data path -- nothing here touches RCCL.

Two ways to drive several GPUs:
  * ``predict_sharded``: one process, one Python thread per device (ctypes releases the GIL
    during C-ABI calls; a gp_ctx is per thread/device), host gather into disjoint slices;
  * one process per GPU under ``torch.distributed.run`` (how bench.py is launched): the only
    communication is the timing barrier / max-reduce in ``RankGroup``, over gloo.
"""
import os
import threading
import time

import numpy as np


def row_shards(n_rows, n_shards):
    """Contiguous row blocks [(start, end)] of equal ceil size; trailing shards may be
    empty when n_rows < n_shards."""
    n_rows, n_shards = int(n_rows), int(n_shards)
    if n_shards <= 0:
        raise ValueError("n_shards must be positive")
    per = -(-n_rows // n_shards) if n_rows > 0 else 0
    out = []
    for g in range(n_shards):
        s = min(n_rows, g * per)
        e = min(n_rows, (g + 1) * per)
        out.append((s, e))
    return out


_device_ctx = {}
_device_ctx_lock = threading.Lock()


def _device_context(device):
    """One long-lived (context, lock) per device for the sharded path: its helper threads,
    pinned staging and streams are created once, not per call or per Python thread."""
    from . import _lib
    with _device_ctx_lock:
        hit = _device_ctx.get(device)
        if hit is None:
            hit = _device_ctx[device] = (_lib.Context(device), threading.Lock())
        return hit


_shard_models = {}      # (device, id(gp), dtype) -> (weakref to gp, HostBlocks, Model); used under the device's lock


def _shard_model(ctx, device, gp, precision):
    """The emulator packed and uploaded on ``device``, kept between predict_sharded calls (a call per time step
    of an assimilation re-sends the same emulator).  Keyed by CONTENT: the entry remembers the array objects it
    was made from and a 64-bit digest of their bytes (``_lib.HostBlocks``; ~10 us for a 250-point emulator) and
    the digest is re-taken on every call, so replacing theta / inputs / invQt / invQ or editing any of them in
    place both lead to a re-pack, as in the reference, which uploads the constants on every call
    (gp_emulator/GaussianProcess.py:289-292).  Entries of emulators that no longer exist are dropped; at most
    eight emulators stay resident per device."""
    import weakref
    from . import _lib
    for k in [k for k, v in list(_shard_models.items()) if k[0] == device and v[0]() is None]:   # dead emulators (ids get reused)
        _shard_models.pop(k)[2].close()
    key = (device, id(gp), np.dtype(precision).str)
    arrays = [gp.theta, gp.inputs, gp.invQt, gp.invQ]
    hit = _shard_models.get(key)
    if hit is not None and hit[0]() is gp and hit[1].same_arrays(arrays) and hit[1].unchanged():
        return hit[2]
    if hit is not None:
        hit[2].close()
        del _shard_models[key]
    mine = [k for k in _shard_models if k[0] == device]
    while len(mine) >= 8:
        _shard_models.pop(mine[0])[2].close()
        mine.pop(0)
    blocks = _lib.HostBlocks(arrays)
    model = _lib.Model(ctx, np.exp(gp.theta), gp.inputs, gp.invQt, gp.invQ, precision)
    _shard_models[key] = (weakref.ref(gp), blocks, model)
    return model


def predict_sharded(gp, testing, devices, precision=np.float64, predict_fn=None, out=None):
    """mean, variance, gradient of ``gp`` at ``testing`` with rows sharded over ``devices``.

    Every device gets one contiguous row block and writes its results straight into its own
    slice of the output arrays (``out=(mu, var, deriv)`` to reuse them) through the library's
    slab pipeline (``Model.predict``): the gather is the D2H copy itself.
    ``predict_fn(device, rows) -> (mu, var, deriv)`` replaces the HIP path in the CPU tests,
    which exercise the sharding/gather logic alone.
    """
    testing = np.ascontiguousarray(testing)
    M, D = testing.shape
    if out is None:
        mu, var, deriv = np.empty(M), np.empty(M), np.empty((M, D))
    else:
        mu, var, deriv = out
    shards = row_shards(M, len(devices))
    errors = []

    def hip_shard(device, s, e):
        ctx, lock = _device_context(device)
        with lock:
            model = _shard_model(ctx, device, gp, precision)
            rows = testing[s:e]
            if rows.dtype != np.float64:
                rows = rows.astype(np.float64)
            model.predict(rows, out=(mu[s:e], var[s:e], deriv[s:e]))

    def work(device, s, e):
        try:
            if e > s:
                if predict_fn is None:
                    hip_shard(device, s, e)
                else:
                    mu[s:e], var[s:e], deriv[s:e] = predict_fn(device, testing[s:e])
        except BaseException as exc:   # surfaced to the caller below
            errors.append((device, exc))

    threads = [threading.Thread(target=work, args=(dev, s, e))
               for dev, (s, e) in zip(devices, shards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    return mu, var, deriv


class SharedOutputs:
    """The gathered result of a one-process-per-GPU run: ONE set of host arrays (mu (M,),
    var (M,), deriv (M, D), float64) in a file under /dev/shm that every rank maps.  Rank r's
    predict writes rows [lo, hi) of it directly (``views(lo, hi)`` as ``out=``): the host gather
    of the north star ("independent row blocks gathered on the host") with no collective and no
    second copy.  The creating rank sizes the file; the others open it after a barrier."""

    def __init__(self, path, n_rows, n_inputs, create):
        import mmap
        self.path, self.M, self.D = path, int(n_rows), int(n_inputs)
        nbytes = self.M * (2 + self.D) * 8
        if create:
            with open(path, "wb") as fh:
                fh.truncate(nbytes)
        self._fh = open(path, "r+b")
        self._mm = mmap.mmap(self._fh.fileno(), nbytes) if nbytes else None
        whole = np.frombuffer(self._mm, dtype=np.float64) if nbytes else np.empty(0)
        self.mu, self.var = whole[:self.M], whole[self.M:2 * self.M]
        self.deriv = whole[2 * self.M:].reshape(self.M, self.D)

    def views(self, lo, hi):
        return self.mu[lo:hi], self.var[lo:hi], self.deriv[lo:hi]

    def close(self, unlink=False):
        self.mu = self.var = self.deriv = None
        try:
            if self._mm is not None:
                self._mm.close()
        except BufferError:          # a caller still holds a view: the mapping goes with it
            pass
        self._fh.close()
        if unlink:
            try:
                os.unlink(self.path)
            except OSError:
                pass


class RankGroup:
    """Barrier + max-reduce over the ranks of a torch.distributed.run launch (gloo).

    world == 1 needs no torch at all.  Only scalars cross ranks: the data path has no
    exchange step, so no RCCL collective exists to call.
    """

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def timed_steps(group, step, sync, steps, warmup):
    """bench.py's timing contract: W untimed steps, then EXACTLY K steps bracketed by a
    barrier + device sync on both sides; returns the MAX over ranks of the wall time."""
    for _ in range(warmup):
        step()
    sync()
    group.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    group.barrier()
    dt = time.perf_counter() - t0
    return group.max(dt)
