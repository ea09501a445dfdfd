"""``PoseNet`` / ``PoseRefineNet`` -- host-side mirrors of the reference modules (lib/network.py:70-132,
170-206) over the HIP engine in libdfusion_hip.so.

What is kept from the reference: class names, constructor arguments, positional ``forward``
signatures, output shapes, ``.cuda()/.eval()/.parameters()/.state_dict()/.load_state_dict()``
and the exact checkpoint key/shape layout (77 tensors for PoseNet, 24 for the refiner), so
``estimator.load_state_dict(torch.load(path))`` works on a reference checkpoint.

What is different: there is no module tree of nn.Conv layers -- parameters are plain
``nn.Parameter`` leaves registered under the reference's keys, and ``forward`` hands device
pointers to the C ABI (include/dfusion.h).  The reference evaluates batch element 0 only
(``b = 0``, network.py:123,202); here a leading batch of B same-sized objects is evaluated
independently and all B results are returned (B = 1 reproduces the reference shapes).
``eval()`` mode runs the fused inference engine; ``train()`` mode runs the differentiable
layer-by-layer path of ``train_graph`` (convolution / GEMM forward, data- and weight-gradients on the MFMA
kernels, autograd as the tape, Dropout2d active), one object per call like the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _lib
from ..synth import posenet_spec, refiner_spec


class _Leaf(nn.Module):
    """Anonymous container so dotted checkpoint keys map onto a module tree."""


class _EngineModule(nn.Module):
    _kind = None  # "posenet" | "refiner"

    def __init__(self, num_points, num_obj):
        super().__init__()
        self.num_points = int(num_points)
        self.num_obj = int(num_obj)
        self._spec = posenet_spec(self.num_obj) if self._kind == "posenet" else refiner_spec(self.num_obj)
        gen = torch.Generator().manual_seed(0)
        for key, shape in self._spec:
            parts = key.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, _Leaf())
                mod = getattr(mod, p)
            # fresh modules start like the reference's (torch defaults of nn.Conv*/nn.Linear: weight and bias ~
            # U(-1/sqrt(fan_in), 1/sqrt(fan_in)); nn.PReLU 0.25) -- with un-normalised 0..255-scale inputs and no
            # normalisation layers a hotter start (He-normal) overflows within a few optimizer steps
            if key.endswith(".conv.2.weight"):
                t = torch.full(shape, 0.25)
            else:
                wshape = shape if not key.endswith(".bias") else dict(self._spec)[key[:-4] + "weight"]
                bound = 1.0 / math.sqrt(int(math.prod(wshape[1:])) or 1)
                t = (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound
            mod.register_parameter(parts[-1], nn.Parameter(t))
        self._handle = None
        self._handle_dev = None
        self._uploaded = {}          # key -> (data_ptr, version)
        self._plist = None
        self._ws = None

    # -- engine handle ---------------------------------------------------------------------------
    def _engine(self, device):
        L = _lib.lib()
        if self._handle is None or self._handle_dev != device:
            self._release()
            with _lib.device_guard(device):
                create = L.df_posenet_create if self._kind == "posenet" else L.df_refiner_create
                self._handle = create(self.num_points, self.num_obj)
            if not self._handle:
                _lib.check(-1, "create")
            self._handle_dev = device
            self._uploaded = {}
        synced = False
        self._upload_keep = []       # converted temporaries of this batch of uploads: alive until the next forward has waited for the copies
        if self._plist is None:      # (the module tree never changes: the walk is a quarter of a millisecond per call)
            self._plist = list(self.named_parameters())
        for key, p in self._plist:
            if p.device != device:
                raise RuntimeError(f"parameter {key} is on {p.device}, input on {device}: call .cuda() first")
            tag = (p.data_ptr(), p._version)
            if self._uploaded.get(key) != tag:
                if not synced:
                    # df_net_load_param enqueues its copies on the null stream, outside torch's streams: an optimizer update
                    # still in flight on the current stream must have landed before the parameters are read
                    torch.cuda.current_stream(device).synchronize()
                    synced = True
                src = p.detach()
                if src.dtype != torch.float32 or not src.is_contiguous():
                    src = src.float().contiguous()
                    self._upload_keep.append(src)
                _lib.check(L.df_net_load_param(self._handle, key.encode(), src.data_ptr(), src.numel()), f"load_param({key})")
                self._uploaded[key] = tag
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None):
            _lib.lib().df_net_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._ws

    def debug_taps(self, enable=True):
        """Arm / disarm the engine's debug taps (``df_net_debug_taps``): the next single-bucket forward keeps copies of its
        named intermediates, read back with ``debug_tap``."""
        dev = next(self.parameters()).device
        with _lib.device_guard(dev):
            _lib.check(_lib.lib().df_net_debug_taps(self._engine(dev), 1 if enable else 0), "debug_taps")

    def debug_tap(self, name):
        """One intermediate of the last forward as a host tensor in the engine's channels-last layout (see include/dfusion.h)."""
        import ctypes
        torch.cuda.synchronize()
        L = _lib.lib()
        shape = (ctypes.c_int64 * 4)()
        _lib.check(L.df_net_debug_tap_read(self._handle, name.encode(), None, 0, shape), f"debug_tap_read({name})")
        out = torch.empty([int(v) for v in shape], dtype=torch.float32)
        _lib.check(L.df_net_debug_tap_read(self._handle, name.encode(), out.data_ptr(), out.numel(), shape), f"debug_tap_read({name})")
        return out

    def _check_mode(self):
        if self.training:
            raise RuntimeError(f"{type(self).__name__}: this entry point is inference-only; call .eval() first")


def _dev_f32(t, device=None):
    if not t.is_cuda:
        raise RuntimeError("densefusion_amd needs device tensors (no CPU path): call .cuda() on the inputs")
    return t.detach().float().contiguous()


class PoseNet(_EngineModule):
    """``PoseNet(num_points, num_obj)`` (lib/network.py:70-93)."""
    _kind = "posenet"

    def forward(self, img, x, choose, obj):
        """img [B,3,H,W], x [B,N,3], choose [B,1,N]|[B,N] int64, obj [B,1]|[B] int64 ->
        (out_rx [B,N,4], out_tx [B,N,3], out_cx [B,N,1], emb [B,32,N])   (lib/network.py:95-132).

        In ``train()`` mode (or when gradients are required) the differentiable layer-by-layer path of
        ``train_graph`` runs instead of the fused inference engine: Dropout2d active, autograd graph built (the reference
        trains with bs = 1; B same-size objects per call are accepted and are equivalent to B calls)."""
        if self.training:
            from . import train_graph
            return train_graph.posenet_forward(self, img.float(), x.float(), choose, obj, dropout=True)
        img, x = _dev_f32(img), _dev_f32(x)
        dev = img.device
        B, C, H, W = img.shape
        N = self.num_points
        if C != 3 or x.shape != (B, N, 3):
            raise RuntimeError(f"PoseNet.forward: expected img [B,3,H,W] and x [B,{N},3], got {tuple(img.shape)}, {tuple(x.shape)}")
        choose = choose.to(device=dev, dtype=torch.int64).reshape(B, N).contiguous()
        obj = obj.to(device=dev, dtype=torch.int64).reshape(B).contiguous()
        out_r = torch.empty(B, N, 4, device=dev)
        out_t = torch.empty(B, N, 3, device=dev)
        out_c = torch.empty(B, N, 1, device=dev)
        emb = torch.empty(B, 32, N, device=dev)
        L = _lib.lib()
        with _lib.device_guard(dev):
            h = self._engine(dev)
            need = L.df_posenet_workspace_bytes(h, B, H, W)
            if need == 0:
                _lib.check(-1, "posenet_workspace_bytes")
            ws = self._workspace(need, dev)
            st = L.df_posenet_forward(h, B, H, W, img.data_ptr(), x.data_ptr(), choose.data_ptr(), obj.data_ptr(),
                                      out_r.data_ptr(), out_t.data_ptr(), out_c.data_ptr(), emb.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _lib.current_stream())
        _lib.check(st, "posenet_forward")
        return out_r, out_t, out_c, emb


    def forward_multi(self, imgs, x, choose, obj):
        """``forward`` over crops of DIFFERENT sizes in one device-side pass (``df_posenet_forward_multi``), eval mode only.

        ``imgs``: list of ``[B_i,3,H_i,W_i]`` tensors, one per crop-size bucket; ``x [sum B,N,3]``, ``choose [sum B,N]``, ``obj [sum B]``:
        the objects of all buckets concatenated in bucket order.  Returns ``(out_rx, out_tx, out_cx, emb)`` in that order, bit-identical
        to per-bucket ``forward`` calls -- the frozen estimator of the refiner phase (tools/train.py:139-145) over a whole window."""
        import ctypes
        self._check_mode()
        imgs = [_dev_f32(i) for i in imgs]
        x = _dev_f32(x)
        dev, N, nb = x.device, self.num_points, len(imgs)
        Bs = [int(i.shape[0]) for i in imgs]
        Bt = sum(Bs)
        if nb == 0 or any(i.dim() != 4 or i.shape[1] != 3 for i in imgs) or x.shape != (Bt, N, 3):
            raise RuntimeError(f"PoseNet.forward_multi: need images [B_i,3,H_i,W_i] and x [{Bt},{N},3], got {[tuple(i.shape) for i in imgs]}, {tuple(x.shape)}")
        choose = choose.to(device=dev, dtype=torch.int64).reshape(Bt, N).contiguous()
        obj = obj.to(device=dev, dtype=torch.int64).reshape(Bt).contiguous()
        out_r, out_t = torch.empty(Bt, N, 4, device=dev), torch.empty(Bt, N, 3, device=dev)
        out_c, emb = torch.empty(Bt, N, 1, device=dev), torch.empty(Bt, 32, N, device=dev)
        arr = ctypes.c_int * nb
        cB, cH, cW = arr(*Bs), arr(*[int(i.shape[2]) for i in imgs]), arr(*[int(i.shape[3]) for i in imgs])
        cimg = (ctypes.c_void_p * nb)(*[i.data_ptr() for i in imgs])
        L = _lib.lib()
        with _lib.device_guard(dev):
            h = self._engine(dev)
            need = L.df_posenet_multi_workspace_bytes(h, nb, cB, cH, cW)
            if need == 0:
                _lib.check(-1, "posenet_multi_workspace_bytes")
            ws = self._workspace(need, dev)
            st = L.df_posenet_forward_multi(h, nb, cB, cH, cW, cimg, x.data_ptr(), choose.data_ptr(), obj.data_ptr(), out_r.data_ptr(),
                                            out_t.data_ptr(), out_c.data_ptr(), emb.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream())
        _lib.check(st, "posenet_forward_multi")
        return out_r, out_t, out_c, emb


class PoseRefineNet(_EngineModule):
    """``PoseRefineNet(num_points, num_obj)`` (lib/network.py:170-185)."""
    _kind = "refiner"

    def forward(self, x, emb, obj):
        """x [B,N,3], emb [B,32,N], obj [B,1]|[B] -> (out_rx [B,4], out_tx [B,3])   (lib/network.py:187-206).
        ``train()`` mode: differentiable path (``train_graph.refiner_forward``)."""
        if self.training:
            from . import train_graph
            return train_graph.refiner_forward(self, x.float(), emb.float(), obj)
        x, emb = _dev_f32(x), _dev_f32(emb)
        dev = x.device
        B, N = x.shape[0], self.num_points
        if x.shape != (B, N, 3) or emb.shape != (B, 32, N):
            raise RuntimeError(f"PoseRefineNet.forward: expected x [B,{N},3], emb [B,32,{N}], got {tuple(x.shape)}, {tuple(emb.shape)}")
        obj = obj.to(device=dev, dtype=torch.int64).reshape(B).contiguous()
        out_r = torch.empty(B, 4, device=dev)
        out_t = torch.empty(B, 3, device=dev)
        L = _lib.lib()
        with _lib.device_guard(dev):
            h = self._engine(dev)
            need = L.df_refiner_workspace_bytes(h, B)
            ws = self._workspace(need, dev)
            st = L.df_refiner_forward(h, B, x.data_ptr(), emb.data_ptr(), obj.data_ptr(), out_r.data_ptr(),
                                      out_t.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream())
        _lib.check(st, "refiner_forward")
        return out_r, out_t


class PoseEstimator:
    """The inner loop of tools/eval_ycb.py:192-229 as one device-side call (no host round trips).

    ``estimate(img, cloud, choose, obj, iteration)`` -> (pose_wo_refine [B,7] f64, pose [B,7] f64);
    each row is quaternion (w,x,y,z) then translation -- what the reference appends to
    ``my_result_wo_refine`` / ``my_result``.
    """

    def __init__(self, estimator: PoseNet, refiner: PoseRefineNet):
        self.estimator, self.refiner = estimator, refiner
        self._ws = None

    def workspace_bytes(self, B, H, W, device):
        L = _lib.lib()
        with _lib.device_guard(device):
            return L.df_estimate_workspace_bytes(self.estimator._engine(device), self.refiner._engine(device), B, H, W)

    def estimate(self, img, cloud, choose, obj, iteration, out=None):
        self.estimator._check_mode(); self.refiner._check_mode()
        img, cloud = _dev_f32(img), _dev_f32(cloud)
        dev = img.device
        B, _, H, W = img.shape
        N = self.estimator.num_points
        choose = choose.to(device=dev, dtype=torch.int64).reshape(B, N).contiguous()
        obj = obj.to(device=dev, dtype=torch.int64).reshape(B).contiguous()
        if out is None:
            out = (torch.empty(B, 7, dtype=torch.float64, device=dev), torch.empty(B, 7, dtype=torch.float64, device=dev))
        pose_wo, pose = out
        L = _lib.lib()
        with _lib.device_guard(dev):
            hp, hr = self.estimator._engine(dev), self.refiner._engine(dev)
            need = L.df_estimate_workspace_bytes(hp, hr, B, H, W)
            if need == 0:
                _lib.check(-1, "estimate_workspace_bytes")
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
            st = L.df_estimate_poses(hp, hr, B, H, W, img.data_ptr(), cloud.data_ptr(), choose.data_ptr(), obj.data_ptr(),
                                     int(iteration), pose_wo.data_ptr(), pose.data_ptr(), self._ws.data_ptr(),
                                     self._ws.numel(), _lib.current_stream())
        _lib.check(st, "estimate_poses")
        return pose_wo, pose

    def estimate_multi(self, imgs, cloud, choose, obj, iteration, out=None):
        """A window of detections of different crop sizes in one device-side call (``df_estimate_poses_multi``).

        ``imgs``: list of ``[B_i,3,H_i,W_i]`` tensors, one per crop-size bucket; ``cloud [sum B,N,3]``, ``choose [sum B,N]``,
        ``obj [sum B]``: the objects of all buckets concatenated in bucket order.  Returns ``(pose_wo_refine, pose)``, each
        ``[sum B,7]`` f64 in the same order; bit-identical to per-bucket ``estimate`` calls."""
        import ctypes
        self.estimator._check_mode(); self.refiner._check_mode()
        imgs = [_dev_f32(i) for i in imgs]
        cloud = _dev_f32(cloud)
        dev = cloud.device
        N = self.estimator.num_points
        nb = len(imgs)
        Bs = [int(i.shape[0]) for i in imgs]
        Btot = sum(Bs)
        if nb == 0 or any(i.dim() != 4 or i.shape[1] != 3 for i in imgs) or cloud.shape != (Btot, N, 3):
            raise RuntimeError(f"estimate_multi: need images [B_i,3,H_i,W_i] and cloud [{Btot},{N},3], got "
                               f"{[tuple(i.shape) for i in imgs]}, {tuple(cloud.shape)}")
        choose = choose.to(device=dev, dtype=torch.int64).reshape(Btot, N).contiguous()
        obj = obj.to(device=dev, dtype=torch.int64).reshape(Btot).contiguous()
        if out is None:
            out = (torch.empty(Btot, 7, dtype=torch.float64, device=dev), torch.empty(Btot, 7, dtype=torch.float64, device=dev))
        pose_wo, pose = out
        arr = ctypes.c_int * nb
        cB, cH, cW = arr(*Bs), arr(*[int(i.shape[2]) for i in imgs]), arr(*[int(i.shape[3]) for i in imgs])
        cimg = (ctypes.c_void_p * nb)(*[i.data_ptr() for i in imgs])
        L = _lib.lib()
        with _lib.device_guard(dev):
            hp, hr = self.estimator._engine(dev), self.refiner._engine(dev)
            need = L.df_estimate_multi_workspace_bytes(hp, hr, nb, cB, cH, cW)
            if need == 0:
                _lib.check(-1, "estimate_multi_workspace_bytes")
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
            st = L.df_estimate_poses_multi(hp, hr, nb, cB, cH, cW, cimg, cloud.data_ptr(), choose.data_ptr(), obj.data_ptr(),
                                           int(iteration), pose_wo.data_ptr(), pose.data_ptr(), self._ws.data_ptr(),
                                           self._ws.numel(), _lib.current_stream())
        _lib.check(st, "estimate_poses_multi")
        return pose_wo, pose
