"""Host side of the native training step (``df_posenet_train_step`` / ``df_refiner_train_step``, csrc/train.hip).

``NativeTrainer(kind, num_points, num_obj, device)`` owns the two flat fp32 buffers a training run needs -- parameters and
accumulated gradients, both in the library's KERNEL layout -- and exposes what tools/train.py does with a network
(reference: tools/train.py:78-99,146-176):

* ``load_state_dict`` / ``state_dict``: the reference's keys and shapes (``torch.save(estimator.state_dict())`` files load
  unchanged); the conversion to / from the kernel layout happens only here.
* ``step_posenet(...)`` / ``step_refiner(...)``: forward + loss + backward of B same-size frames in ONE library call --
  no autograd tape, no per-layer Python, nothing read back; gradients are accumulated into ``.grad``.
* ``.data`` / ``.grad`` / ``.numel`` / ``zero_grad()``: the interface ``train_utils.FlatAdam`` and
  ``train_utils.allreduce_gradients`` already work on (Adam and the all-reduce are layout-agnostic).

PyTorch only provides the device memory and the stream.  There is no CPU path.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

POSENET, REFINER = 0, 1


class NativeTrainer:
    def __init__(self, kind, num_points, num_obj, device):
        self.kind = POSENET if kind in (POSENET, "posenet") else REFINER
        self.num_points, self.num_obj = int(num_points), int(num_obj)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("densefusion_amd needs a GPU device (no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        L = _lib.lib()
        with _lib.device_guard(self.device):
            self._h = L.df_trainer_create(self.kind, self.num_points, self.num_obj)
        if not self._h:
            _lib.check(-1, "trainer_create")
        self.numel = int(L.df_trainer_flat_numel(self._h))
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.version = 0                     # bumped whenever .data changes (optimizer step, load_state_dict)
        self.module = None                   # (FlatAdam looks for an nn.Module to invalidate; there is none)
        self._ws = None
        self._calls = 0
        self._salt = 0                       # a lane's offset into the dropout seed sequence (Lanes)
        self.spec = []
        key, shape, ndim = ctypes.create_string_buffer(256), (ctypes.c_int64 * 4)(), ctypes.c_int()
        for i in range(L.df_trainer_num_params(self._h)):
            _lib.check(L.df_trainer_param_info(self._h, i, key, 256, shape, ctypes.byref(ndim)), "trainer_param_info")
            self.spec.append((key.value.decode(), tuple(int(shape[d]) for d in range(ndim.value))))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().df_trainer_destroy(self._h)
                self._h = None
        except Exception:        # noqa: BLE001
            pass

    # ---- checkpoints: the reference's keys / shapes ----
    def load_state_dict(self, sd, strict=True):
        L = _lib.lib()
        missing = [k for k, _ in self.spec if k not in sd]
        extra = [k for k in sd if k not in dict(self.spec)]
        if strict and (missing or extra):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:4]}, unexpected keys {extra[:4]}")
        with _lib.device_guard(self.device):
            for k, shape in self.spec:
                if k not in sd:
                    continue
                t = sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous()
                if tuple(t.shape) != shape:
                    raise RuntimeError(f"load_state_dict: {k} has shape {tuple(t.shape)}, expected {shape}")
                _lib.check(L.df_trainer_pack_param(self._h, k.encode(), t.data_ptr(), self.data.data_ptr(), _lib.current_stream()), "trainer_pack_param")
        torch.cuda.current_stream(self.device).synchronize()          # the staging tensors above die here
        self.version += 1

    def _unpack(self, flat):
        L = _lib.lib()
        out = {}
        with _lib.device_guard(self.device):
            for k, shape in self.spec:
                t = torch.zeros(shape, dtype=torch.float32, device=self.device)
                _lib.check(L.df_trainer_unpack_param(self._h, k.encode(), flat.data_ptr(), t.data_ptr(), _lib.current_stream()), "trainer_unpack_param")
                out[k] = t
        return out

    def state_dict(self):
        return self._unpack(self.data)

    def grad_dict(self):
        """The accumulated gradients in the reference's layout (tests / inspection)."""
        return self._unpack(self.grad)

    def zero_grad(self):
        self.grad.zero_()

    def bump_version(self):
        self.version += 1

    def _workspace(self, need):
        if need == 0:
            _lib.check(-1, "train_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return self._ws

    @staticmethod
    def _sym(sym_host, B):
        if sym_host is None:
            return None
        vals = [int(bool(v)) for v in sym_host]
        if len(vals) != B:
            raise RuntimeError("symmetric flags: one per frame")
        return (ctypes.c_int * B)(*vals)

    def _next_seed(self):
        """Dropout2d seed of this call: a 64-bit mix of (torch's seed, the lane, the call number) -- distinct per lane and call (an additive
        lane offset would meet another lane's sequence after enough calls), reproducible for a given torch.manual_seed."""
        self._calls += 1
        m = (1 << 64) - 1
        x = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + (self._salt + 1) * 0xD1B54A32D192ED03 + self._calls) & m
        x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & m
        x ^= x >> 27; x = (x * 0x94D049BB133111EB) & m
        x ^= x >> 31
        return x & 0x3FFFFFFF

    @staticmethod
    def _check_graph_dropout(graph_safe, dropout, seed):
        # the seed is a by-value launch argument: a captured step replays the SAME three channel masks on every replay
        if graph_safe and dropout and seed is None:
            raise RuntimeError("graph_safe=True with dropout=True would freeze the Dropout2d masks into the captured graph: pass dropout=False, "
                               "or an explicit seed if one fixed mask set per capture is what you want")

    # ---- PoseNet + Loss (tools/train.py:152-153,161) ----
    def step_posenet(self, img, cloud, choose, obj, target, model_points, symmetric, w, dropout=True, seed=None, want_pred=False,
                     graph_safe=False):
        """img [B,3,H,W], cloud [B,N,3], choose [B,N]|[B,1,N], obj [B]|[B,1] (device), target / model_points [B,M,3];
        symmetric: HOST booleans, one per frame (``idx in sym_list``).  -> dict(loss [B], dis [B], new_points [B,N,3],
        new_target [B,M,3], emb [B,32,N] (+ pred_r / pred_t / pred_c when want_pred)).  Gradients of sum_b loss[b] are ADDED to .grad."""
        assert self.kind == POSENET
        f32 = lambda t: t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        img, cloud, target, model_points = f32(img), f32(cloud), f32(target), f32(model_points)
        B, _, H, W = img.shape
        N, M = self.num_points, target.shape[-2]
        if cloud.shape != (B, N, 3) or target.numel() != B * M * 3 or model_points.numel() != B * M * 3:
            raise RuntimeError(f"step_posenet: expected cloud [{B},{N},3] and target / model_points [{B},M,3]")
        choose = choose.to(device=self.device, dtype=torch.int64).reshape(B, N).contiguous()
        obj = obj.to(device=self.device, dtype=torch.int64).reshape(B).contiguous()
        dev = self.device
        out = dict(loss=torch.empty(B, device=dev), dis=torch.empty(B, device=dev), new_points=torch.empty(B, N, 3, device=dev),
                   new_target=torch.empty(B, M, 3, device=dev), emb=torch.empty(B, 32, N, device=dev))
        if want_pred:
            out.update(pred_r=torch.empty(B, N, 4, device=dev), pred_t=torch.empty(B, N, 3, device=dev), pred_c=torch.empty(B, N, 1, device=dev))
        self._check_graph_dropout(graph_safe, dropout, seed)
        if seed is None:
            seed = self._next_seed()
        P = lambda k: out[k].data_ptr() if k in out else None
        L = _lib.lib()
        with _lib.device_guard(dev):
            ws = self._workspace(L.df_posenet_train_workspace_bytes(self._h, B, H, W, M))
            st = L.df_posenet_train_step(self._h, self.data.data_ptr(), self.grad.data_ptr(), -1 if graph_safe else self.version, B, H, W,
                                         img.data_ptr(), cloud.data_ptr(), choose.data_ptr(), obj.data_ptr(), target.data_ptr(),
                                         model_points.data_ptr(), M, self._sym(symmetric, B), float(w), int(bool(dropout)), int(seed) & 0xFFFFFFFF,
                                         P("loss"), P("dis"), P("new_points"), P("new_target"), P("pred_r"), P("pred_t"), P("pred_c"), P("emb"),
                                         ws.data_ptr(), ws.numel(), _lib.current_stream())
        _lib.check(st, "posenet_train_step")
        return out

    # ---- a window of frames of different crop sizes as ONE pass (df_posenet_train_step_multi) ----
    def step_posenet_multi(self, imgs, cloud, choose, obj, target, model_points, symmetric, w, dropout=True, seed=None, want_pred=False,
                           graph_safe=False):
        """``imgs``: list of ``[B_i,3,H_i,W_i]`` tensors, one per crop-size bucket; every other argument as in ``step_posenet`` with the
        frames of all buckets concatenated in bucket order (``sum B_i`` frames).  Per-point layers, 1x1 convolutions, Winograd-domain
        products and every weight gradient run once over all buckets; the gradient added to ``.grad`` is the sum of the frames'
        one-per-pass gradients up to fp32 summation order."""
        assert self.kind == POSENET
        f32 = lambda t: t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        imgs = [f32(i) for i in imgs]
        cloud, target, model_points = f32(cloud), f32(target), f32(model_points)
        nb = len(imgs)
        Bs = [int(i.shape[0]) for i in imgs]
        B = sum(Bs)
        N, M = self.num_points, target.shape[-2]
        if nb == 0 or any(i.dim() != 4 or i.shape[1] != 3 for i in imgs) or cloud.shape != (B, N, 3) or target.numel() != B * M * 3 \
                or model_points.numel() != B * M * 3:
            raise RuntimeError(f"step_posenet_multi: need images [B_i,3,H_i,W_i], cloud [{B},{N},3] and target / model_points [{B},M,3]")
        choose = choose.to(device=self.device, dtype=torch.int64).reshape(B, N).contiguous()
        obj = obj.to(device=self.device, dtype=torch.int64).reshape(B).contiguous()
        dev = self.device
        out = dict(loss=torch.empty(B, device=dev), dis=torch.empty(B, device=dev), new_points=torch.empty(B, N, 3, device=dev),
                   new_target=torch.empty(B, M, 3, device=dev), emb=torch.empty(B, 32, N, device=dev))
        if want_pred:
            out.update(pred_r=torch.empty(B, N, 4, device=dev), pred_t=torch.empty(B, N, 3, device=dev), pred_c=torch.empty(B, N, 1, device=dev))
        self._check_graph_dropout(graph_safe, dropout, seed)
        if seed is None:
            seed = self._next_seed()
        P = lambda k: out[k].data_ptr() if k in out else None
        arr = ctypes.c_int * nb
        cB, cH, cW = arr(*Bs), arr(*[int(i.shape[2]) for i in imgs]), arr(*[int(i.shape[3]) for i in imgs])
        cimg = (ctypes.c_void_p * nb)(*[i.data_ptr() for i in imgs])
        L = _lib.lib()
        with _lib.device_guard(dev):
            ws = self._workspace(L.df_posenet_train_multi_workspace_bytes(self._h, nb, cB, cH, cW, M))
            st = L.df_posenet_train_step_multi(self._h, self.data.data_ptr(), self.grad.data_ptr(), -1 if graph_safe else self.version, nb, cB, cH, cW, cimg,
                                               cloud.data_ptr(), choose.data_ptr(), obj.data_ptr(), target.data_ptr(), model_points.data_ptr(), M,
                                               self._sym(symmetric, B), float(w), int(bool(dropout)), int(seed) & 0xFFFFFFFF,
                                               P("loss"), P("dis"), P("new_points"), P("new_target"), P("pred_r"), P("pred_t"), P("pred_c"), P("emb"),
                                               ws.data_ptr(), ws.numel(), _lib.current_stream())
        _lib.check(st, "posenet_train_step_multi")
        return out

    def step_posenet_window(self, frames, w, dropout=True, seed=None):
        """``frames``: list of dicts / tuples ``(img [3,H,W], cloud [N,3], choose [N]|[1,N], obj [1], target [M,3], model_points [M,3], symmetric)``
        of ANY crop sizes: buckets them by (H, W) and runs them as one multi-bucket pass.  Returns (out, order): the step's outputs and, for
        every output row, the index of its frame in ``frames``."""
        keys = ("img", "cloud", "choose", "obj", "target", "model_points", "symmetric")
        fr = [f if isinstance(f, dict) else dict(zip(keys, f)) for f in frames]
        by_size = {}
        for j, f in enumerate(fr):
            by_size.setdefault((int(f["img"].shape[-2]), int(f["img"].shape[-1])), []).append(j)
        order = [j for idxs in by_size.values() for j in idxs]
        N = self.num_points
        imgs = [torch.stack([fr[j]["img"].reshape(3, *hw) for j in idxs]) for hw, idxs in by_size.items()]
        cat = lambda k, shape: torch.stack([fr[j][k].reshape(shape) for j in order])
        out = self.step_posenet_multi(imgs, cat("cloud", (N, 3)), cat("choose", (N,)), cat("obj", (1,)), cat("target", (-1, 3)), cat("model_points", (-1, 3)),
                                      [bool(fr[j]["symmetric"]) for j in order], w, dropout=dropout, seed=seed)
        return out, order

    def set_splitk(self, enable):
        """Split-K of the small-grid launches (default on).  Off: a frame's gradient no longer depends on what shares its pass."""
        _lib.check(_lib.lib().df_trainer_set_splitk(self._h, int(bool(enable))), "trainer_set_splitk")

    # ---- executed-FLOP profile of the step's MFMA launches (df_trainer_profile) ----
    def profile(self, enable=True):
        _lib.check(_lib.lib().df_trainer_profile(self._h, int(bool(enable))), "trainer_profile")

    def profile_read(self):
        """After a device sync: {kind: (ms, executed FLOPs, launches)} for kind in fwd / dgrad / wgrad since the last read."""
        ms, fl, n = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int * 3)()
        _lib.check(_lib.lib().df_trainer_profile_read(self._h, ms, fl, n), "trainer_profile_read")
        return {k: (ms[i], fl[i], n[i]) for i, k in enumerate(("fwd", "dgrad", "wgrad"))}

    # ---- PoseRefineNet + Loss_refine (tools/train.py:156-159) ----
    def step_refiner(self, points, emb, obj, target, model_points, symmetric, graph_safe=False):
        """One refine iteration: points [B,N,3] (in the current pose's frame), emb [B,32,N], target / model_points [B,M,3]
        -> dict(dis [B], new_points, new_target).  Gradients of sum_b dis[b] are ADDED to .grad."""
        assert self.kind == REFINER
        f32 = lambda t: t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        points, emb, target, model_points = f32(points), f32(emb), f32(target), f32(model_points)
        B, N, M = points.shape[0], self.num_points, target.shape[-2]
        if points.shape != (B, N, 3) or emb.shape != (B, 32, N):
            raise RuntimeError(f"step_refiner: expected points [{B},{N},3], emb [{B},32,{N}]")
        obj = obj.to(device=self.device, dtype=torch.int64).reshape(B).contiguous()
        dev = self.device
        out = dict(dis=torch.empty(B, device=dev), new_points=torch.empty(B, N, 3, device=dev), new_target=torch.empty(B, M, 3, device=dev))
        L = _lib.lib()
        with _lib.device_guard(dev):
            ws = self._workspace(L.df_refiner_train_workspace_bytes(self._h, B, M))
            st = L.df_refiner_train_step(self._h, self.data.data_ptr(), self.grad.data_ptr(), -1 if graph_safe else self.version, B, points.data_ptr(),
                                         emb.data_ptr(), obj.data_ptr(), target.data_ptr(), model_points.data_ptr(), M, self._sym(symmetric, B),
                                         out["dis"].data_ptr(), out["new_points"].data_ptr(), out["new_target"].data_ptr(), ws.data_ptr(),
                                         ws.numel(), _lib.current_stream())
        _lib.check(st, "refiner_train_step")
        return out


class Lanes:
    """Run the passes of one accumulation window on `n` concurrent lanes.

    A bs = 1 pass of the reference (tools/train.py:146-163) is a chain of a few hundred launches on sub-chip grids: one pass at a
    time leaves most of the 256 CUs idle and the host thread is busy for as long as the GPU.  The frames of a window are
    independent until their gradients are summed, so pass j runs on lane j % n: every lane has its own HIP stream, host thread
    (the library call releases the GIL), workspace and gradient buffer, and reads the SAME parameter buffer; after the window the
    lanes' gradients are added into lane 0's in lane order.  The result depends on n (a different summation order) but not on
    timing: a fixed n gives bit-identical windows run to run."""

    def __init__(self, trainer: NativeTrainer, n):
        from concurrent.futures import ThreadPoolExecutor
        self.tr, self.n = trainer, max(1, int(n))
        self.lanes = [trainer] + [NativeTrainer(trainer.kind, trainer.num_points, trainer.num_obj, trainer.device) for _ in range(self.n - 1)]
        for li, lane in enumerate(self.lanes):
            lane._salt = li                       # mixed into the seed hash (_next_seed)
        from .streams import concurrent_streams
        self.streams = concurrent_streams(trainer.device, self.n)      # tested to run side by side (streams.py)
        self.pool = ThreadPoolExecutor(max_workers=self.n, thread_name_prefix="df-lane") if self.n > 1 else None

    def run(self, jobs):
        """jobs: callables ``f(lane_trainer) -> result``; returns their results in job order."""
        if self.n == 1 or len(jobs) <= 1:
            return [f(self.tr) for f in jobs]
        dev = self.tr.device
        main = torch.cuda.current_stream(dev)
        results = [None] * len(jobs)

        def work(li):
            torch.cuda.set_device(dev)
            lane = self.lanes[li]
            with torch.cuda.stream(self.streams[li]):
                if li:
                    lane.data, lane.version = self.tr.data, self.tr.version
                for j in range(li, len(jobs), self.n):
                    results[j] = jobs[j](lane)

        used = min(self.n, len(jobs))
        for st in self.streams[:used]:
            st.wait_stream(main)
        for fut in [self.pool.submit(work, li) for li in range(used)]:
            fut.result()
        for st in self.streams[:used]:
            main.wait_stream(st)
        for lane in self.lanes[1:used]:
            self.tr.grad.add_(lane.grad)
            lane.grad.zero_()
        return results

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
            self.pool = None
