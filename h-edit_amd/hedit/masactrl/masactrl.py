"""``MutualSelfAttentionControl`` -- mirrors text-guided/masactrl/masactrl.py:11-69.

Reference semantics (:53-69): in the self-attention of transformer block ``cur_att_layer // 2`` in
``layer_idx``, at editor step ``cur_step`` in ``step_idx``, the batch is split into its unconditional and
conditional halves and EVERY row of a half attends with its own queries to the keys and values of the
FIRST row of that half (the source image).  Here that is a per-row index handed to the self-attention
kernel (``hedit_p2p_plan.kv_src`` + ``kv_first_block``): no probabilities are materialised.  ``layer_idx``
must be a contiguous range up to the last block (what ``start_layer`` produces and the drivers use)."""
import torch

from .. import _lib
from .masactrl_utils import AttentionBase


class MutualSelfAttentionControl(AttentionBase):
    MODEL_TYPE = {"SD": 16, "SDXL": 70}

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, model_type="SD"):
        super().__init__()
        self.total_steps = total_steps
        self.total_layers = self.MODEL_TYPE.get(model_type, 16)
        self.start_step = start_step
        self.start_layer = start_layer
        self.layer_idx = layer_idx if layer_idx is not None else list(range(start_layer, self.total_layers))
        self.step_idx = step_idx if step_idx is not None else list(range(start_step, total_steps))
        if list(self.layer_idx) != list(range(min(self.layer_idx, default=self.total_layers), self.total_layers)):
            raise NotImplementedError("layer_idx must be range(k, total_layers) (the kernel gate is 'block >= k')")
        self._cache = {}

    def _plan(self, unet, B, H, W, save_attn):
        """B rows = [unconditional half | conditional half], each half laid out [source rows..., target rows...]
        with n images per kind (n = B / 4 in the h-Edit passes: [x_orig|null]*n, [x_k|null]*n, [x_orig|src]*n, [x_k|tar]*n):
        row (kind, i) reads the keys / values of row (first kind of its half, i)."""
        if B % 4:
            raise ValueError("a MasaCtrl pass expects [x_orig|null]*n, [x_edit|null]*n, [x_orig|src]*n, [x_edit|tar]*n rows")
        key = (id(unet), B)
        st = self._cache.get(key)
        if st is None:
            n = B // 4
            ar = torch.arange(B, dtype=torch.int32)
            kv = ar.clone()
            kv[n:2 * n] = ar[0:n]
            kv[3 * n:4 * n] = ar[2 * n:3 * n]
            st = (ar.to(unet.device), kv.to(unet.device))
            self._cache = {key: st}
        p = _lib.P2PPlan()
        p.mode = 1
        p.n_pairs = 0
        p.singles = st[0].data_ptr()
        p.n_single = B
        if self.cur_step in self.step_idx and len(self.layer_idx):
            p.kv_src = st[1].data_ptr()
            p.kv_first_block = int(min(self.layer_idx))
        p._keep = st
        return p
