"""The fp32 ORACLE's trajectory of BASELINE configs[1] at SD-1.5 shape, generated in the build container (no GPU needed) so that the
GPU box only runs the HIP side of the divergence curve (round 5 held an MI355X lease for 146 minutes of host time).

    python tests/golden/make_loop_trajectory.py [out_scale=0.3] [threads=4]
      -> tests/golden/t1_sd15_loop_trajectory_g<gain>.npz

Contents (all fp32): `xT` (1,4,64,64) and `zs` (50,1,4,64,64) = oracle/loops.py::ddpm_inversion of a seeded latent under the source
prompt; `trace` (50,2,4,64,64) = [x_orig, x_edit] after each of the 50 steps of oracle/loops.py::h_edit_p2p_implicit (Replace + Reweight +
LocalBlend, K = 1, cfg 1.0 / 5.0 / 7.5, weight_reconstruction 0.1); `w0`; `timesteps`.  Weights are the seeded synthetic ones of
tests/helpers/models.py (seed 3, output layer damped by out_scale: a random-init eps network at full gain is not contractive).  This is the
ORACLE's output (oracle/ is pinned on the reference by g1-g16), not a vector of the reference itself: the reference's UNet body is
diffusers', absent offline.  Consumers: tests/test_gpu_loop_trajectory.py, tests/diag/diag_loop_divergence.py.
Deterministic: torch CPU generators, fixed thread-independent kernels are NOT guaranteed by torch, so the file records the thread count;
re-generation on another host agrees to fp32 rounding noise (1e-6), not bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "h-edit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

out_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.set_num_threads(threads)
from helpers.models import make_oracle  # noqa: E402
from helpers.tiny import PROMPT_PAIRS  # noqa: E402
from hedit.unet import SD15_CONFIG  # noqa: E402
from oracle import loops as OL, p2p as OP  # noqa: E402

T = 50
om, _ = make_oracle(SD15_CONFIG, T, seed=3, out_scale=out_scale)
src, tar, blend, is_replace = PROMPT_PAIRS[0]
torch.manual_seed(11)
w0 = torch.randn(1, 4, 64, 64) * 0.8
t0 = time.time()
with torch.no_grad():
    torch.manual_seed(100)
    zs, wts, _ = OL.ddpm_inversion(om, w0, eta=1.0, prompt=src, cfg_src=1.0, T=T)
print(f"oracle DDPM inversion, {T} steps: {time.time() - t0:.0f} s", flush=True)
bw = ((blend[0],), (blend[1],))
eq = {"words": (blend[1],), "values": (2.0,)}
oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer)
trace = []
inner = oc.step_callback


def cb(xt):
    xt = inner(xt)
    trace.append(xt.detach().float().cpu().clone())
    print(f"step {len(trace)} / {T}  {time.time() - t0:.0f} s", flush=True)
    return xt


oc.step_callback = cb
OP.register(om, oc)
kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], after_skip_steps=T, is_ddim_inversion=False,
          weight_reconstruction=0.1, optimization_steps=1)
with torch.no_grad():
    OL.h_edit_p2p_implicit(om, xT=wts[T], zs=zs[:T], controller=oc, **kw)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"t1_sd15_loop_trajectory_g{out_scale:g}.npz")
np.savez(out, w0=w0.numpy(), xT=wts[T].numpy(), zs=zs[:T].numpy(), trace=torch.stack(trace).numpy(),
         timesteps=np.asarray([int(v) for v in om.scheduler.timesteps], dtype=np.int64), out_scale=np.float32(out_scale),
         threads=np.int64(threads))
print("wrote", out, f"{os.path.getsize(out) / 1e6:.1f} MB, total {time.time() - t0:.0f} s")
