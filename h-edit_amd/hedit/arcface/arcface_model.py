"""The identity reward of the face-swapping task -- drop-in for ``IDLoss`` of the reference's
face-swapping/arcface/arcface_model.py:11-67 and its IR-SE50 backbone
(arcface/facial_recognition/model_irse.py:9-48, helpers.py:28-119): crop [35:223, 32:220] of the
256 x 256 image -> adaptive average pool to 112 x 112 -> IR-SE50 (BatchNorm in eval mode, PReLU,
squeeze-excitation) -> l2-normalised 512-d feature; ``get_cosine_loss`` = 1 - cos(feature(image),
feature(reference face)).

Like the CLIP encoders this reward network stays a torch module on PyTorch-ROCm: the face loop only
needs its value and its gradient w.r.t. the image (h_edit_R.py:103-106), which torch autograd provides
exactly as in the reference; the eps-network it guides is the HIP executor (hedit.diffusion.Model).
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

    def forward(self, x):
        s = torch.sigmoid(self.fc2(F.relu(self.fc1(x.mean((2, 3), keepdim=True)))))
        return x * s


class _Unit(nn.Module):
    """bottleneck_IR_SE (helpers.py:97-119): BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN -> SE, plus a
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

    def forward(self, x):
        sc = self.shortcut_layer(x) if self.shortcut_layer is not None else x[:, :, ::self.stride, ::self.stride]
        return self.res_layer(x) + sc


class Backbone(nn.Module):
    """IR-SE50 at input_size 112 (what IDLoss builds: Backbone(112, 50, mode='ir_se'))."""

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

    def forward(self, x):
        x = self.output_layer(self.body(self.input_layer(x)))
        return x / torch.norm(x, 2, 1, True)

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


class IDLoss(nn.Module):
    """``IDLoss(ref_path)`` as the reference, plus where the backbone weights come from: ``weights`` = local
    path of model_ir_se50.pth or a state_dict; None = seeded random weights (synthetic runs)."""

    def __init__(self, ref_path=None, weights=None, ref=None, device=None, seed=0):
        super().__init__()
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

    def extract_feats(self, x):
        if x.shape[2] != 256:
            x = F.adaptive_avg_pool2d(x, (256, 256))
        x = x[:, :, 35:223, 32:220]                       # crop the face region
        return self.facenet(F.adaptive_avg_pool2d(x, (112, 112)))

    def get_cosine_sim(self, image):
        img_feat = F.normalize(self.extract_feats(image), p=2, dim=-1)
        if self._ref_feat is None or self._ref_feat.device != img_feat.device:
            with torch.no_grad():      # constant w.r.t. the image; the reference re-encodes it on every call (:51)
                self._ref_feat = F.normalize(self.extract_feats(self.ref), p=2, dim=-1)
        return F.cosine_similarity(self._ref_feat, img_feat, dim=-1)

    def get_cosine_loss(self, image):
        return (1 - self.get_cosine_sim(image)).mean()
