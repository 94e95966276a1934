"""MasaCtrl mutual self-attention (oracle; see oracle/__init__.py).  Restates
  AttentionBase / MutualSelfAttentionControl   text-guided/masactrl/masactrl_utils.py:6-33, masactrl/masactrl.py:11-69
  the patched attention forward                text-guided/masactrl/masactrl_utils.py:39-87
as a diffusers-style attention processor, so that it plugs into the same toy / oracle UNets as the P2P
processor.  PINNED on vectors from running the reference classes (tests/golden/g13_masactrl.npz)."""
import torch


class MutualSelfAttention:
    def __init__(self, start_step=4, start_layer=10, total_steps=50, total_layers=16):
        self.cur_step, self.cur_att_layer, self.num_att_layers = 0, 0, -1
        self.step_idx = list(range(start_step, total_steps))
        self.layer_idx = list(range(start_layer, total_layers))

    def active(self, is_cross):
        return (not is_cross) and self.cur_step in self.step_idx and self.cur_att_layer // 2 in self.layer_idx

    def count(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1

    def step_callback(self, x):
        return x


class MasaProcessor:
    """q, k, v per head; plain softmax(q k^T scale) v, except where the editor is active: each half of the batch
    (unconditional | conditional) attends with its own queries to the keys / values of its FIRST row."""

    def __init__(self, editor):
        self.editor = editor

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, use_editor=True):
        x = hidden_states
        is_cross = encoder_hidden_states is not None
        ctx = encoder_hidden_states if is_cross else x
        h = attn.heads
        b, n, c = x.shape
        q = attn.to_q(x).reshape(b, n, h, c // h).transpose(1, 2)                 # (b, h, n, d)
        k = attn.to_k(ctx).reshape(b, ctx.shape[1], h, c // h).transpose(1, 2)
        v = attn.to_v(ctx).reshape(b, ctx.shape[1], h, c // h).transpose(1, 2)
        if use_editor and self.editor.active(is_cross):
            half = b // 2
            k = torch.cat([k[:1].expand(half, -1, -1, -1), k[half:half + 1].expand(b - half, -1, -1, -1)])
            v = torch.cat([v[:1].expand(half, -1, -1, -1), v[half:half + 1].expand(b - half, -1, -1, -1)])
        p = torch.softmax(q @ k.transpose(-1, -2) * attn.scale, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(b, n, c)
        if use_editor:
            self.editor.count()
        to_out = attn.to_out[0] if isinstance(attn.to_out, torch.nn.ModuleList) else attn.to_out
        return to_out(o)


def register_editor(model, editor):
    procs = {name: MasaProcessor(editor) for name in model.unet.attn_processors.keys()}
    model.unet.set_attn_processor(procs)
    editor.num_att_layers = len(procs)
