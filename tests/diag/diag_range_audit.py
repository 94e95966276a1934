"""Range audit for half storage (HEDIT_STORAGE=f16, csrc/common.h): IEEE half tops out at 65504, every storage conversion of the
half build rounds a larger magnitude to inf.  This script runs the fp32 ORACLE (same arithmetic as the kernels, pinned on the
reference) with forward hooks on every leaf module and on the residual-stream sums and prints the largest |activation| per level,
for the SD-1.5-shaped UNet (synthetic weights at FULL output gain, inputs at the scale of x_T and of a late step) and for the SD VAE
decoder / encoder at 512^2.  CPU only -- no GPU minutes (VERDICT r5 weak 1 / advisor).   python tests/diag/diag_range_audit.py [threads]

What the numbers mean: a tensor the HIP path stores in HBM is one of these (leaf-module outputs, residual sums); values that only
live in fp32 registers (accumulators, softmax statistics, GroupNorm sums) never pass through half.  Random-init weights are scaled
N(0, 1/fan_in) (tests/helpers: every layer roughly variance-preserving), so the maxima are those of a well-conditioned network; a
trained SD-1.5 has heavier tails in a few channels (activations of a few hundred in the 1280-channel levels are known from fp16
inference with diffusers, which is the mode users run that checkpoint in) -- still two orders of magnitude below 65504."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "h-edit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
from helpers.models import make_oracle  # noqa: E402
from hedit.unet import SD15_CONFIG, random_state_dict  # noqa: E402
from hedit.vae import SD15_VAE_CONFIG  # noqa: E402
from oracle import sd_unet as OSU, sd_vae as OSV  # noqa: E402

HALF_MAX = 65504.0


def audit(module, run, label, level_of):
    peaks = {}

    def note(name, t):
        if isinstance(t, torch.Tensor) and t.is_floating_point():
            lvl = level_of(name)
            v = float(t.detach().abs().max())
            if v > peaks.get(lvl, (0.0, ""))[0]:
                peaks[lvl] = (v, name)
    hooks = [m.register_forward_hook(lambda m_, i_, o_, n=n: note(n, o_)) for n, m in module.named_modules() if len(list(m.children())) == 0]
    resid_peak = [0.0]
    if hasattr(OSU, "RESID_STORE"):
        def rs(x):
            resid_peak[0] = max(resid_peak[0], float(x.abs().max()))
            return x
        OSU.RESID_STORE = rs
    with torch.no_grad():
        run()
    OSU.RESID_STORE = None
    for h in hooks:
        h.remove()
    print(f"# {label}")
    worst = 0.0
    for lvl in sorted(peaks):
        v, name = peaks[lvl]
        worst = max(worst, v)
        print(f"  {lvl:34s} max |x| = {v:10.3f}   ({name})   headroom to 65504: {HALF_MAX / max(v, 1e-30):9.1f}x")
    if resid_peak[0] > 0:
        worst = max(worst, resid_peak[0])
        print(f"  {'residual-stream sums':34s} max |x| = {resid_peak[0]:10.3f}   headroom {HALF_MAX / resid_peak[0]:9.1f}x")
    print(f"  -> largest stored magnitude {worst:.3f} = 2^{torch.log2(torch.tensor(worst)).item():.1f}; half overflows at 2^16")
    return worst


def unet_level(name):
    parts = name.split(".")
    if parts[0] in ("down_blocks", "up_blocks"):
        kind = parts[2] if len(parts) > 2 else ""
        return f"{parts[0]}.{parts[1]} {kind}"
    return parts[0]


def vae_level(name):
    parts = name.split(".")
    return ".".join(parts[:3]) if len(parts) > 3 else ".".join(parts[:2])


om, _ = make_oracle(SD15_CONFIG, 50, seed=3, out_scale=1.0)
torch.manual_seed(5)
ctx = om.text_encoder(om.tokenizer(["a green lizard is sitting on a branch"] * 2, padding="max_length", max_length=77, return_tensors="pt").input_ids)[0]
for t, scale in ((981, 1.0), (1, 1.0), (981, 4.0)):
    x = torch.randn(2, 4, 64, 64) * scale
    audit(om.unet, lambda: om.unet(x, torch.tensor(t), encoder_hidden_states=ctx), f"SD-1.5-shaped UNet, t = {t}, latent rms {scale}", unet_level)

vae = OSV.AutoencoderKL(SD15_VAE_CONFIG)
vae.load_state_dict(random_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, 3))
vae.eval()
z = torch.randn(1, 4, 64, 64) * 5.5              # latents / 0.18215: rms ~ 5.5
audit(vae.decoder, lambda: vae.decode(z), "SD VAE decoder, 64x64 latent -> 512^2, input rms 5.5 (= unit latent / 0.18215)", vae_level)
img = torch.rand(1, 3, 512, 512) * 2 - 1
audit(vae.encoder, lambda: vae.encode(img), "SD VAE encoder, 512^2 image in [-1, 1]", vae_level)
