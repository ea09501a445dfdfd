"""Data-parallel training plumbing for the training path of tools/train.py.

* ``FlatParams``: one contiguous fp32 buffer for all parameters and one for all gradients (the parameters
  of the model become views into it), so the gradient exchange is ONE collective and the optimizer ONE kernel.
* ``allreduce_gradients``: sum over ranks of the flat gradient buffer -- 85.8 MB for PoseNet(21), 7.8 MB for
  the refiner -- with torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
  The reference has no collective at all (single GPU); this is the only one on the training path.
* ``FlatAdam``: Adam on the flat buffer through ``df_adam_step`` (same update rule and defaults as the
  ``optim.Adam`` of tools/train.py:99), the 1/(world x accumulation) average folded into the step.
"""
from __future__ import annotations

import torch

from . import _lib


class FlatParams:
    def __init__(self, module):
        params = [p for p in module.parameters()]
        self.params = params
        self.module = module
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.data = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.data[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + k].view_as(p)            # parameter storage now lives in the flat buffer
            p.grad = self.grad[off:off + k].view_as(p)            # autograd accumulates straight into the flat buffer
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()


def allreduce_gradients(flat: FlatParams, group=None):
    """Sum the flat gradient buffer over all ranks (no-op without an initialised process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)
        return dist.get_world_size(group)
    return 1


def replicas_in_sync(modules, group=None):
    """True when every rank holds the same parameters in `modules`: per tensor a (sum, sum of squares) checksum in fp64, gathered
    through host memory (a few KB, any backend) and compared exactly.  tools/train.py calls it once after initialisation /
    resume: data-parallel replicas that start apart are averaged by every gradient all-reduce and never meet again."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    sums = []
    for m in modules:
        for p in m.parameters():
            d = p.detach().double()
            sums += [d.sum(), (d * d).sum()]
    mine = torch.stack(sums).cpu() if sums else torch.zeros(0, dtype=torch.float64)
    every = [None] * dist.get_world_size(group)
    dist.all_gather_object(every, mine.tolist(), group=group)
    return all(e == every[0] for e in every)


class FlatAdam:
    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, float(lr), betas, eps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.t = 0

    def step(self, grad_scale=1.0):
        self.t += 1
        f = self.flat
        if not f.data.is_cuda:
            raise RuntimeError("FlatAdam needs device tensors (no CPU path)")
        with _lib.device_guard(f.data.device):
            st = _lib.lib().df_adam_step(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                         f.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.t, float(grad_scale),
                                         _lib.current_stream())
        _lib.check(st, "adam_step")
        # the update bypasses torch's version counters: tell the inference engine its parameter copies are stale
        if hasattr(f.module, "_uploaded"):
            f.module._uploaded = {}
        if hasattr(f, "bump_version"):          # NativeTrainer: its cached data-gradient weight copies are stale now
            f.bump_version()


def with_host_index(t_dev, values):
    """Attach the host copy of an object-index tensor to its device copy (``t._host``: list of ints).  The losses need the index on
    the host to pick their branch (symmetric object or not); reading it back from the device (`.item()`) is a device synchronisation
    per frame, which at the reference's one frame per pass leaves the GPU waiting for the Python side.  Callers that already hold the
    index on the host (every data loader does) attach it; without the hint the losses fall back to `.item()`."""
    t_dev._host = [int(v) for v in (values.reshape(-1).tolist() if hasattr(values, "reshape") else values)]
    return t_dev


def host_index(idx):
    """First object index of `idx` without a device synchronisation when the hint is there (or `idx` lives on the host)."""
    h = getattr(idx, "_host", None)
    if h is not None:
        return int(h[0])
    return int(idx.reshape(-1)[0].item())


class _HostHalf:
    """What the loader's worker processes see of a dataset: ``host_item`` only (the CPU half of a fetch)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, i):
        return self.dataset.host_item(int(i))


class _Order:
    """A sampler whose index list can be replaced between epochs (the worker processes persist)."""

    def __init__(self, order):
        self.order = list(order)

    def __iter__(self):
        return iter(list(self.order))

    def __len__(self):
        return len(self.order)


class Prefetcher:
    """Bounded look-ahead over ``dataset[i] for i in order`` (the job of the reference's 10-worker DataLoader, tools/train.py:106).
    Iteration yields the frames IN ORDER as tuples of device tensors whose upload the consumer's current stream has been made to
    wait for (and which that stream is recorded as a user of, so the allocator does not hand their memory to a later upload while
    the consumer's kernels are still queued).  Items that are not tuples of tensors (a loader's sentinel) and tensors that already
    live on the device pass through.

    * ``workers`` threads fetch ``dataset[i]``, pin host tensors and upload them on a copy stream, at most ``depth`` frames ahead.
      ``workers = 0``: fetch synchronously in the caller's thread (same results).
    * ``processes`` > 0 and a dataset that splits its fetch (``host_item(i)`` -> host tensors, ``device_item(i, host)`` -> the item):
      the CPU half -- PNG decoding, the Python-level sampling -- runs in that many worker PROCESSES (torch's DataLoader, "spawn"
      context: they never touch the device; shared-memory + pinned hand-over), one feeder thread runs the device half on the copy stream.  Python
      threads share one interpreter lock: the disk datasets decode ~4x faster on threads and ~N x on N processes.
      The worker processes persist over ``set_order()`` / repeated iteration until ``close()``."""

    def __init__(self, dataset, order, device, workers=4, depth=None, processes=0):
        self.dataset, self.device = dataset, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._order = _Order(int(i) for i in order)
        self.workers = max(0, int(workers))
        self.processes = max(0, int(processes)) if hasattr(dataset, "host_item") else 0
        self.depth = int(depth) if depth else max(2, 2 * max(self.workers, self.processes))
        self.copy_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._loader = None

    @property
    def order(self):
        return self._order.order

    def set_order(self, order):
        self._order.order = [int(i) for i in order]
        return self

    def close(self):
        if self._loader is not None:
            it = getattr(self._loader, "_iterator", None)
            if it is not None and hasattr(it, "_shutdown_workers"):
                it._shutdown_workers()
            self._loader = None

    def _upload(self, item):
        if self.copy_stream is None or not isinstance(item, (tuple, list)):
            return item, None
        with torch.cuda.stream(self.copy_stream):
            out = []
            for t in item:
                if torch.is_tensor(t) and not t.is_cuda:
                    d = t.pin_memory().to(self.device, non_blocking=True)
                    if not t.is_floating_point() and t.numel() <= 16:
                        d._host = t.reshape(-1).tolist()       # small index tensors keep their host values (with_host_index)
                    t = d
                out.append(t)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return tuple(out), ev

    def _fetch(self, i):
        if self.copy_stream is None:
            return self.dataset[i], None
        with torch.cuda.stream(self.copy_stream):              # a dataset that prepares on the device does so on the copy stream
            item = self.dataset[i]
        return self._upload(item)

    def _hand_over(self, item, ev):
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in item:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        return item

    def __len__(self):
        return len(self._order)

    def __iter__(self):
        if self.processes:
            yield from self._iter_processes()
            return
        order = list(self._order.order)
        if self.workers == 0:
            for i in order:
                yield self._hand_over(*self._fetch(i))
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="df-prefetch") as pool:
            pending, it = deque(), iter(order)
            for i in it:
                pending.append(pool.submit(self._fetch, i))
                if len(pending) >= self.depth:
                    break
            while pending:
                item, ev = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(self._fetch, nxt))
                yield self._hand_over(item, ev)

    def _iter_processes(self):
        import queue
        import threading
        from torch.utils.data import DataLoader
        if self._loader is None:
            self._loader = DataLoader(_HostHalf(self.dataset), batch_size=None, sampler=self._order, num_workers=self.processes,
                                      multiprocessing_context="spawn", pin_memory=self.copy_stream is not None, prefetch_factor=2,
                                      persistent_workers=True)
            import warnings                  # torch's own pin thread passes the deprecated `device` argument, once per tensor
            warnings.filterwarnings("ignore", message=r".*(pin_memory|is_pinned).*deprecated.*")
        order = list(self._order.order)
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()

        def put(x):
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def feed():
            try:
                if self.copy_stream is not None:
                    torch.cuda.set_device(self.device)
                for i, host in zip(order, self._loader):
                    if self.copy_stream is not None:
                        with torch.cuda.stream(self.copy_stream):
                            item = self.dataset.device_item(i, host)
                    else:
                        item = self.dataset.device_item(i, host)
                    if not put(self._upload(item)):
                        return
                put(None)
            except BaseException as e:                          # surfaces in the consumer's thread
                put(e)

        th = threading.Thread(target=feed, name="df-feed", daemon=True)
        th.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                yield self._hand_over(*got)
        finally:
            stop.set()
            th.join()
