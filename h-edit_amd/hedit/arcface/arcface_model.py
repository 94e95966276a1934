"""The identity reward of the face-swapping task -- drop-in for ``IDLoss`` of the reference's
face-swapping/arcface/arcface_model.py:11-67 and its IR-SE50 backbone
(arcface/facial_recognition/model_irse.py:9-48, helpers.py:28-119): crop [35:223, 32:220] of the
256 x 256 image -> adaptive average pool to 112 x 112 -> IR-SE50 (BatchNorm in eval mode, PReLU,
squeeze-excitation) -> l2-normalised 512-d feature; ``get_cosine_loss`` = 1 - cos(feature(image),
feature(reference face)).

The network runs natively and only natively (``hedit_irse50_cos_fwd_bwd`` of libhedit_hip.so, csrc/irse.hip):
the face loop only needs the loss and its gradient w.r.t. the image (h_edit_R.py:103-106), which one native
call produces together (fp32-quality split-bf16 contractions on the MFMA GEMM kernel); ``get_cosine_loss`` and
``get_cosine_sim`` wrap it in autograd nodes so the caller's ``torch.autograd.grad(loss, x_{t-1})`` works unchanged.
The torch modules below are PARAMETER CONTAINERS (the checkpoint's state_dict layout, no forward): there is no
torch / CPU execution path in the product -- the fp32 restatement the golden vectors pin lives in
oracle/reward_nets.py (test infrastructure).
state_dict keys are the reference's (``input_layer.0.weight``, ``body.3.res_layer.5.fc1.weight`` ...), so
its ``model_ir_se50.pth`` loads directly from a local path; nothing is downloaded.
``LPIPS_Loss`` (arcface_model.py:69-94) wraps the third-party ``lpips`` package, absent offline: pass any
module with ``get_lpips_loss`` or None (the loop guards it like the reference, h_edit_R.py:124)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# IR-50 layout (helpers.py:37-44): (in_channel, depth, units) per stage, first unit of a stage has stride 2
_STAGES_50 = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))


class _SE(nn.Module):
    def __init__(self, c, reduction=16):
        super().__init__()
        self.fc1 = nn.Conv2d(c, c // reduction, 1, bias=False)
        self.fc2 = nn.Conv2d(c // reduction, c, 1, bias=False)


class _Unit(nn.Module):
    """Parameters of bottleneck_IR_SE (helpers.py:97-119): BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN -> SE, plus a
    strided identity (MaxPool2d(1, stride)) or 1x1 conv + BN shortcut."""

    def __init__(self, cin, depth, stride):
        super().__init__()
        self.stride = stride
        if cin != depth:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(cin, depth, 1, stride, bias=False), nn.BatchNorm2d(depth))
        else:
            self.shortcut_layer = None
        self.res_layer = nn.Sequential(nn.BatchNorm2d(cin), nn.Conv2d(cin, depth, 3, 1, 1, bias=False), nn.PReLU(depth),
                                       nn.Conv2d(depth, depth, 3, stride, 1, bias=False), nn.BatchNorm2d(depth), _SE(depth))


class Backbone(nn.Module):
    """Parameters of IR-SE50 at input_size 112 (what IDLoss builds: Backbone(112, 50, mode='ir_se'))."""

    def __init__(self, input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se", affine=True):
        super().__init__()
        if input_size != 112 or num_layers != 50 or mode != "ir_se":
            raise NotImplementedError("only the configuration IDLoss uses is built: input 112, 50 layers, ir_se")
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        units = []
        for cin, depth, n in _STAGES_50:
            units.append(_Unit(cin, depth, 2))
            units.extend(_Unit(depth, depth, 1) for _ in range(n - 1))
        self.body = nn.Sequential(*units)
        self.output_layer = nn.Sequential(nn.BatchNorm2d(512), nn.Dropout(drop_ratio), nn.Flatten(),
                                          nn.Linear(512 * 7 * 7, 512), nn.BatchNorm1d(512, affine=affine))

    def init_random(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self.state_dict().items():
                if name.endswith("num_batches_tracked"):
                    continue
                if name.endswith("running_var"):
                    t.copy_(1.0 + 0.1 * torch.rand(t.shape, generator=g))
                elif t.dim() > 1:
                    t.copy_(torch.randn(t.shape, generator=g) * float(t[0].numel()) ** -0.5)
                elif "weight" in name:
                    t.copy_(1.0 + 0.05 * torch.randn(t.shape, generator=g) if "res_layer.2" not in name and "input_layer.2" not in name
                            else 0.25 + 0.02 * torch.randn(t.shape, generator=g))
                else:
                    t.copy_(0.02 * torch.randn(t.shape, generator=g))
        return self


def load_face_image(path, size=256):
    """PIL -> bilinear resize to size x size -> [-1, 1] CHW (arcface_model.py:29-36)."""
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    return (x * 2 - 1).unsqueeze(0)


class _NativeCosLoss(torch.autograd.Function):
    """mean_b (1 - cos) with the image gradient computed by the same native call (forward + backward fused)."""

    @staticmethod
    def forward(ctx, image, owner):
        loss, grad = owner._native_loss_and_grad(image.detach(), 1.0 / image.shape[0])
        ctx.save_for_backward(grad)
        return loss.mean()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class _NativeCosSim(torch.autograd.Function):
    """cos(feature(image_b), feature(reference)) per image, differentiable (the reference's get_cosine_sim is): the
    native call returns 1 - cos_b and the gradient of each image's own loss"""

    @staticmethod
    def forward(ctx, image, owner):
        loss, grad = owner._native_loss_and_grad(image.detach(), 1.0)
        ctx.save_for_backward(grad)
        return 1.0 - loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return -grad * g.view(-1, 1, 1, 1), None


class IDLoss(nn.Module):
    """``IDLoss(ref_path)`` as the reference, plus where the backbone weights come from: ``weights`` = local
    path of model_ir_se50.pth or a state_dict; None = seeded random weights (synthetic runs).  Every method takes CUDA
    tensors and runs on the HIP executor; there is no CPU / torch path."""

    def __init__(self, ref_path=None, weights=None, ref=None, device=None, seed=0):
        super().__init__()
        self._h = None
        self._ws = None
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
        if weights is None:
            self.facenet.init_random(seed)
        else:
            self.facenet.load_state_dict(torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights)
        self.facenet.eval()
        for p in self.facenet.parameters():
            p.requires_grad_(False)
        if ref is None:
            if ref_path is None:
                raise ValueError("IDLoss needs the reference face: ref_path (image file) or ref (tensor in [-1, 1])")
            ref = load_face_image(ref_path)
        self.register_buffer("ref", ref.float())
        self._ref_feat = None
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ native executor (csrc/irse.hip)
    @staticmethod
    def _require_gpu(x):
        if not x.is_cuda:
            raise RuntimeError("IDLoss runs on the HIP executor only: pass CUDA tensors (there is no CPU / torch path)")

    def _native(self, device):
        """create / load / finalize the native backbone on first use (parameters by the reference's names)"""
        import ctypes as C
        from .. import _lib
        if self._h is not None:
            return self._h
        lib = _lib.lib()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.hedit_irse50_create(C.byref(h)))
            sd = self.facenet.state_dict()
            for i in range(lib.hedit_irse50_num_params(h)):
                name = lib.hedit_irse50_param_name(h, i).decode()
                w = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
                _lib.check(lib.hedit_irse50_load(h, name.encode(), _lib.ptr(w), w.numel(), _lib.cur_stream()))
                torch.cuda.current_stream().synchronize()
            _lib.check(lib.hedit_irse50_finalize(h, _lib.cur_stream()))
        self._h, self._lib = h, lib
        return h

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.hedit_irse50_destroy(self._h)
            except Exception:
                pass

    def _workspace(self, B, device):
        B = max(B, self.ref.shape[0])
        need = self._lib.hedit_irse50_workspace_bytes(self._h, B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _to256(self, x):
        """the reference's ``F.adaptive_avg_pool2d(x, (256, 256))`` for inputs of another size (arcface_model.py:41-42).  When
        the size divides 256 every pooling window is one pixel, i.e. the map is pixel replication: written as
        repeat_interleave its backward is a plain (deterministic, batch-independent) block sum, where torch's
        adaptive-pool backward accumulates with atomics."""
        if x.shape[2] == 256 and x.shape[3] == 256:
            return x
        if 256 % x.shape[2] == 0 and 256 % x.shape[3] == 0:
            return x.repeat_interleave(256 // x.shape[2], 2).repeat_interleave(256 // x.shape[3], 3)
        return F.adaptive_avg_pool2d(x, (256, 256))

    def _native_features(self, x):
        from .. import _lib
        x = self._to256(x).detach().float().contiguous()
        h = self._native(x.device)
        ws = self._workspace(x.shape[0], x.device)
        feat = torch.empty(x.shape[0], 512, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hedit_irse50_features(h, _lib.ptr(x), x.shape[0], _lib.ptr(feat), _lib.ptr(ws), ws.numel(),
                                                       _lib.cur_stream()))
        return feat

    def _ref_feature(self, device):
        if self._ref_feat is None or self._ref_feat.device != device:
            with torch.no_grad():      # constant w.r.t. the image; the reference re-encodes it on every call (:51)
                self._ref_feat = self._native_features(self.ref.to(device))
        return self._ref_feat

    def _native_loss_and_grad(self, x, scale):
        """(loss [B] = 1 - cos per image, d (scale * sum(loss)) / d x) for x (B,3,256,256)"""
        from .. import _lib
        x = x.float().contiguous()
        B = x.shape[0]
        h = self._native(x.device)
        ref = self._ref_feature(x.device)
        per_image = ref.shape[0] > 1           # lock-step batches: one reference face per image
        if per_image and ref.shape[0] != B:
            raise ValueError("one reference face per batch item expected")
        ws = self._workspace(B, x.device)
        loss = torch.empty(B, device=x.device)
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hedit_irse50_cos_fwd_bwd(h, _lib.ptr(x), _lib.ptr(ref), int(per_image), B, float(scale), _lib.ptr(loss),
                                                          _lib.ptr(grad), _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
        return loss, grad

    # ------------------------------------------------------------------ the reference's surface
    def extract_feats(self, x):
        """l2-normalised 512-d features (arcface_model.py:40-47).  Not differentiable here: differentiate through
        get_cosine_sim / get_cosine_loss, which is what the loop does."""
        self._require_gpu(x)
        if x.requires_grad:
            raise NotImplementedError("IDLoss.extract_feats has no autograd node; use get_cosine_sim / get_cosine_loss")
        return self._native_features(x)

    def get_cosine_sim(self, image):
        self._require_gpu(image)
        return _NativeCosSim.apply(self._to256(image), self)

    def get_cosine_loss(self, image):
        self._require_gpu(image)
        # the 256 x 256 pooling (if any) stays a torch op in front of the native node
        return _NativeCosLoss.apply(self._to256(image), self)
