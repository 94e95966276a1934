"""The face model's pixel-space DDPM UNet on the native HIP executor (csrc/ddpm.hip) behind the interface
of the reference's ``face-swapping/diffusion/diffusion.py::Model`` (:192-341): constructed from the
same config keys, called as ``model(x, t)`` with ``x`` (n, 3, S, S) and ``t`` the (n,) float tensor of
equal timesteps the reference passes (h_edit_R.py:70-71; a scalar works too) -> eps (n, 3, S, S) fp32;
attributes ``in_channels`` / ``resolution`` read by the SDE inversion (sde_inversion.py:32-33);
state_dict key names of the reference class.  GPU only -- there is no eager path."""
import ctypes as C

import torch

from .. import _lib
from ..unet import random_state_dict

# the CelebA-HQ 256 x 256 model the reference loads (face-swapping/main_edit.py: config of the DDPM checkpoint)
CELEBA_HQ_CONFIG = dict(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2,
                        attn_resolutions=(16,), dropout=0.0, image_size=256, resamp_with_conv=True,
                        num_diffusion_timesteps=1000)
TINY_DDPM_CONFIG = dict(type="simple", in_channels=3, out_ch=3, ch=64, ch_mult=(1, 2), num_res_blocks=1,
                        attn_resolutions=(16,), dropout=0.0, image_size=32, resamp_with_conv=True,
                        num_diffusion_timesteps=1000)


class Model:
    # the timestep is a launch parameter: a (n,) tensor that still lives on the host is read without a device
    # synchronisation, so callers that know t on the host (the loops do) should not move it to the GPU first
    accepts_host_timesteps = True

    def __init__(self, config=None, device="cuda:0"):
        cfg = dict(CELEBA_HQ_CONFIG)
        cfg.update(config or {})
        if not cfg.get("resamp_with_conv", True):
            raise NotImplementedError("resamp_with_conv=False (average-pool resampling) is not built")
        if cfg.get("type", "simple") == "bayesian":
            raise NotImplementedError("the 'bayesian' variant's logvar parameter is unused by the sampler and not built")
        self.config = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hedit.diffusion.Model runs on the GPU only (HIP kernels)")
        self.in_channels = cfg["in_channels"]
        self.resolution = cfg["image_size"]
        self.ch = cfg["ch"]
        self._lib = _lib.lib()
        c = _lib.DdpmCfg()
        c.in_channels, c.out_ch, c.ch = cfg["in_channels"], cfg["out_ch"], cfg["ch"]
        mult = list(cfg["ch_mult"])
        c.n_levels = len(mult)
        res = cfg["image_size"]
        for i, m in enumerate(mult):
            c.ch_mult[i] = m
            c.attn_level[i] = int(res in tuple(cfg["attn_resolutions"]))
            if i != len(mult) - 1:
                res //= 2
        c.num_res_blocks, c.image_size = cfg["num_res_blocks"], cfg["image_size"]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_ddpm_create(C.byref(c), C.byref(h)))
        self._h = h
        self.param_shapes = {}
        nd, dims = C.c_int(), (C.c_int * 4)()
        for i in range(self._lib.hedit_ddpm_num_params(self._h)):
            name = self._lib.hedit_ddpm_param_name(self._h, i).decode()
            _lib.check(self._lib.hedit_ddpm_param_shape(self._h, i, C.byref(nd), dims))
            self.param_shapes[name] = tuple(dims[k] for k in range(nd.value))
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.hedit_ddpm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_state_dict(self, sd, strict=True):
        sd = {k: v for k, v in sd.items() if k != "logvar"}
        missing = [k for k in self.param_shapes if k not in sd]
        extra = [k for k in sd if k not in self.param_shapes]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} ({len(missing)}), "
                           f"unexpected {extra[:5]} ({len(extra)})")
        st = _lib.cur_stream()
        with torch.cuda.device(self.device):
            for name, shape in self.param_shapes.items():
                if name not in sd:
                    continue
                w = sd[name]
                if tuple(w.shape) != shape:
                    raise ValueError(f"{name}: expected shape {shape}, got {tuple(w.shape)}")
                w = w.detach().to(device=self.device, dtype=torch.float32).contiguous()
                _lib.check(self._lib.hedit_ddpm_load(self._h, name.encode(), _lib.ptr(w), w.numel(), st))
                torch.cuda.current_stream().synchronize()
        return self

    def init_random(self, seed=0):
        sd = random_state_dict(self.param_shapes, seed)
        self.load_state_dict(sd)
        return sd

    # the reference toggles these on the module; nothing to do for an inference-only executor
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("the HIP executor is bound to its creation device")
        return self

    # ---------------------------------------------------------------- forward
    def __call__(self, x, t):
        return self.forward(x, t)

    @torch.no_grad()
    def forward(self, x, t):
        assert x.shape[2] == x.shape[3] == self.resolution
        if torch.is_tensor(t):
            tv = t.detach().float().reshape(-1)
            t0 = float(tv[0])
            if tv.numel() > 1 and not bool((tv == tv[0]).all()):
                raise NotImplementedError("one timestep per call (the reference passes ones(n) * t)")
        else:
            t0 = float(t)
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        need = self._lib.hedit_ddpm_workspace_bytes(self._h, B)
        if need == 0:
            raise RuntimeError("hedit_ddpm_workspace_bytes failed: " + self._lib.hedit_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, self.config["out_ch"], self.resolution, self.resolution, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_ddpm_forward(self._h, _lib.ptr(x), t0, B, _lib.ptr(out), _lib.ptr(self._ws),
                                                    self._ws.numel(), _lib.cur_stream()))
        return out
