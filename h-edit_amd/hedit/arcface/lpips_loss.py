"""The perceptual reward of the face-swapping task -- drop-in for ``LPIPS_Loss`` of the reference's
face-swapping/arcface/arcface_model.py:69-94, which wraps the third-party ``lpips.LPIPS(net='vgg')`` (lpips==0.1.4,
absent offline) against the source image: ``get_lpips_loss(x) = LPIPS(x, src).mean()``.

``LPIPSNet`` restates that package's published network -- ScalingLayer (shift / scale buffers) -> torchvision VGG16
``features`` cut into five slices (taps relu1_2, relu2_2, relu3_3, relu4_3, relu5_3) -> ``normalize_tensor`` over
channels (eps 1e-10) -> squared difference -> ``lin{k}`` 1x1 convolutions without bias -> spatial mean -> sum -- with
the package's state_dict names (``net.slice1.0.weight``, ``lin0.model.1.weight``, ``scaling_layer.shift`` ...), so a
locally saved ``lpips.LPIPS(net='vgg').state_dict()`` loads directly; nothing is downloaded.  PARITY UNPINNED: the
package is not in the reference tree and the reference holds no vector for it.

The loss AND its gradient w.r.t. the image come from one native call (``hedit_lpips_fwd_bwd``, csrc/lpips.hip),
wrapped in an autograd node so that the loop's ``torch.autograd.grad(lpips_loss, x_{t-1})``
(inversion/h_edit_R.py:124-132) works unchanged.  ``LPIPSNet`` is a PARAMETER CONTAINER (no forward): there is no
torch / CPU execution path in the product -- the fp32 restatement the native path is tested against lives in
oracle/reward_nets.py (test infrastructure)."""
import ctypes as C

import torch
import torch.nn as nn

_VGG = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
        (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512))
_SLICE = (1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5)
_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)          # torchvision vgg16.features indices of the convolutions


class _Lin(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(c, 1, 1, bias=False))


class _Scaling(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])


class LPIPSNet(nn.Module):
    """Parameters of lpips.LPIPS(net='vgg', version '0.1', lpips=True, spatial=False)."""

    def __init__(self):
        super().__init__()
        self.scaling_layer = _Scaling()
        self.net = nn.Module()
        for k in range(1, 6):
            self.net.add_module(f"slice{k}", nn.Module())
        for (cin, cout), sl, idx in zip(_VGG, _SLICE, _IDX):
            getattr(self.net, f"slice{sl}").add_module(str(idx), nn.Conv2d(cin, cout, 3, padding=1))
        for t, c in enumerate((64, 128, 256, 512, 512)):
            self.add_module(f"lin{t}", _Lin(c))

    def init_random(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self.state_dict().items():
                if name.startswith("scaling_layer"):
                    continue
                if name.startswith("lin"):
                    t.copy_(torch.rand(t.shape, generator=g) / t.shape[1])          # the trained lin weights are non-negative
                elif t.dim() > 1:
                    t.copy_(torch.randn(t.shape, generator=g) * (2.0 / float(t[0].numel())) ** 0.5)
                else:
                    t.copy_(0.05 * torch.randn(t.shape, generator=g))
        return self


class _NativeLpips(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner):
        loss, grad = owner._native_loss_and_grad(x.detach())
        ctx.save_for_backward(grad)
        return loss.mean()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class LPIPS_Loss(nn.Module):
    """``LPIPS_Loss(src_path)`` as the reference; plus ``src`` (tensor in [-1, 1], (1,3,H,W) or one source per image
    (n,3,H,W) for lock-step batches), ``weights`` (local path of a saved lpips state_dict, or a dict; None = seeded
    random weights for synthetic runs).  CUDA tensors only: the HIP executor is the one execution path."""

    def __init__(self, src_path=None, src=None, weights=None, device=None, seed=0):
        super().__init__()
        self.lpips_loss = LPIPSNet()
        if weights is None:
            self.lpips_loss.init_random(seed)
        else:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            sd = {k: v for k, v in sd.items() if not k.startswith("lins.")}        # lpips >= 0.1.4 lists the lin layers twice
            self.lpips_loss.load_state_dict(sd)
        self.lpips_loss.eval()
        for p in self.lpips_loss.parameters():
            p.requires_grad_(False)
        if src is None:
            if src_path is None:
                raise ValueError("LPIPS_Loss needs the source image: src_path (file) or src (tensor in [-1, 1])")
            from .arcface_model import load_face_image
            src = load_face_image(src_path)
        self.register_buffer("src", src.float())
        self._h = None
        self._ws = None
        self._src_feats = None
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ native executor (csrc/lpips.hip)
    def _native(self, device):
        from .. import _lib
        if self._h is not None:
            return self._h
        lib = _lib.lib()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.hedit_lpips_create(C.byref(h)))
            sd = self.lpips_loss.state_dict()
            for i in range(lib.hedit_lpips_num_params(h)):
                name = lib.hedit_lpips_param_name(h, i).decode()
                w = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
                _lib.check(lib.hedit_lpips_load(h, name.encode(), _lib.ptr(w), w.numel(), _lib.cur_stream()))
                torch.cuda.current_stream().synchronize()
            _lib.check(lib.hedit_lpips_finalize(h, _lib.cur_stream()))
        self._h, self._lib = h, lib
        return h

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.hedit_lpips_destroy(self._h)
            except Exception:
                pass

    def _workspace(self, B, H, W, device):
        need = self._lib.hedit_lpips_workspace_bytes(self._h, B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _source_features(self, device, H, W):
        from .. import _lib
        key = (str(device), H, W)
        if self._src_feats is None or self._src_feats[0] != key:
            src = self.src.to(device).float().contiguous()
            if src.shape[2:] != (H, W):
                raise ValueError(f"source image is {tuple(src.shape[2:])}, the sample {H}x{W}")
            h = self._native(device)
            n = src.shape[0]
            feats = torch.empty(n, self._lib.hedit_lpips_feature_floats(H, W), device=device)
            ws = self._workspace(n, H, W, device)
            with torch.cuda.device(device):
                _lib.check(self._lib.hedit_lpips_source(h, _lib.ptr(src), n, H, W, _lib.ptr(feats), _lib.ptr(ws), ws.numel(),
                                                        _lib.cur_stream()))
            self._src_feats = (key, feats)
        return self._src_feats[1]

    def _native_loss_and_grad(self, x):
        """(loss [B] = LPIPS(x_b, src), d mean(loss) / d x)"""
        from .. import _lib
        x = x.float().contiguous()
        B, _, H, W = x.shape
        h = self._native(x.device)
        feats = self._source_features(x.device, H, W)
        per_image = feats.shape[0] > 1
        if per_image and feats.shape[0] != B:
            raise ValueError("one source image per batch item expected")
        ws = self._workspace(B, H, W, x.device)
        loss = torch.empty(B, device=x.device)
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hedit_lpips_fwd_bwd(h, _lib.ptr(x), _lib.ptr(feats), int(per_image), B, H, W, 1.0 / B,
                                                     _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
        return loss, grad

    # ------------------------------------------------------------------ the reference's surface
    def get_lpips_loss(self, x):
        if not x.is_cuda:
            raise RuntimeError("LPIPS_Loss runs on the HIP executor only: pass CUDA tensors (there is no CPU / torch path)")
        return _NativeLpips.apply(x, self)
