"""Diagnostic (not collected by pytest): the eps error of ONE SD-1.5-shaped UNet evaluation against the fp32 oracle in the storage
format the process runs in -- `HEDIT_STORAGE=f16 python tests/diag/diag_storage_eps_error.py [rows]` for the half-storage build,
without the variable for bfloat16.  Same weights, inputs and metric as tests/test_gpu_unet.py::test_sd15_unet_forward_full_size
(which also prints the three storage emulations of the oracle: bf16 1.32e-2, bf16 + fp32 residual stream 1.09e-2, fp16 1.72e-3)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import gpu as G  # noqa: E402
from helpers.models import make_pair  # noqa: E402
from hedit import _lib  # noqa: E402
from hedit.unet import SD15_CONFIG  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    t0 = time.time()
    hip, om, _ = make_pair(SD15_CONFIG, 50, seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 64, 64, generator=g)[:rows]
    ctx = torch.randn(2, 77, SD15_CONFIG["cross_attention_dim"], generator=g)[:rows]
    got = hip.unet(G.f32(x), 481, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"use_controller": False}).sample
    G.sync()
    print(f"storage {_lib.STORAGE}: HIP evaluated after {time.time() - t0:.0f} s, finite {bool(torch.isfinite(got).all())}", flush=True)
    with torch.no_grad():
        want = om.unet(x, torch.tensor(481), encoder_hidden_states=ctx).sample
    print(f"storage {_lib.STORAGE}: sd15 eps error vs fp32 oracle, {rows} row(s): {G.rel_err(got, want):.3e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
