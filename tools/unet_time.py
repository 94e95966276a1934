"""ms per SD-1.5-shaped UNet call (random weights, no controller) at a given number of rows, in the storage format of the process
(HEDIT_STORAGE=f16 for the half-storage build).  `python tools/unet_time.py [rows] [calls]`"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit import _lib  # noqa: E402
if os.environ.get("HEDIT_LIB_VARIANT"):       # tools/build_variant.sh side library (A/B of a kernel source on one box)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
from hedit.unet import SD15_CONFIG, UNet2DConditionModel  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    flags = int(os.environ.get("HEDIT_TEST_FLAGS", "0"))        # A/B runs: 8 = one-shot igemm instead of the persistent kernels
    if flags:
        _lib.check(_lib.lib().hedit_test_set_flags(flags))
    unet = UNet2DConditionModel(dict(SD15_CONFIG), device="cuda:0")
    g = torch.Generator().manual_seed(3)
    sd = {k: torch.randn(*s, generator=g) * (0.02 if len(s) > 1 else 0.1) + (1.0 if k.endswith("norm.weight") or "norm" in k and k.endswith("weight") else 0.0)
          for k, s in unet.param_shapes.items()}
    unet.load_state_dict(sd)
    x = torch.randn(rows, 4, 64, 64, generator=g).to(dev)
    ctx = torch.randn(rows, 77, SD15_CONFIG["cross_attention_dim"], generator=g).to(dev)
    kw = dict(encoder_hidden_states=ctx, cross_attention_kwargs={"use_controller": False})
    out = unet(x, 481, **kw).sample
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    h0 = time.perf_counter()
    for _ in range(calls):
        out = unet(x, 481, **kw).sample
    host_ms = (time.perf_counter() - h0) * 1e3 / calls           # host time to ENQUEUE one call (no sync inside): the launch-rate headroom
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / calls
    print(f"storage {_lib.STORAGE}{' flags ' + str(flags) if flags else ''}: {rows} rows, {ms:.2f} ms per UNet call = {rows * 803.2 / ms:.0f} TFLOP/s algorithmic, finite "
          f"{bool(torch.isfinite(out).all())}; host enqueue {host_ms:.2f} ms per call", flush=True)


if __name__ == "__main__":
    main()
