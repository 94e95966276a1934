"""Plug-and-Play injection (oracle; see oracle/__init__.py).  Restates the hooks of
text-guided/plug_n_play/pnp_utils.py:12-154 for the oracle SD UNet (oracle/sd_unet.py):
  register_time(model, t)            -> the current timestep seen by the hooks
  register_pnp(model, qk, conv)      -> q/k injection in the self-attention of up_blocks[1].attentions[1:],
                                        up_blocks[2], up_blocks[3]; conv2-output injection in up_blocks[1].resnets[1]
Both fire only for a batch whose half is ONE row (``B // 2 == 1``) and copy row 0 into row 1 (:46-55, :131-140).
PINNED on vectors from running the reference's own pnp_utils / pnp_h_edit on this same oracle UNet (g14)."""
import torch
import torch.nn.functional as F


class PnPState:
    def __init__(self, qk_schedule, conv_schedule):
        self.qk, self.conv, self.t = qk_schedule, conv_schedule, None

    def on(self, sched, batch):
        return sched is not None and (self.t in sched or self.t == 1000) and batch // 2 == 1


class PnPSelfProcessor:
    def __init__(self, state, inject):
        self.state, self.inject = state, inject

    def __call__(self, attn, x, encoder_hidden_states=None, attention_mask=None, temb=None, **_kw):
        ctx = x if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = attn.to_q(x), attn.to_k(ctx), attn.to_v(ctx)
        if self.inject and encoder_hidden_states is None and self.state.on(self.state.qk, q.shape[0]):
            q, k = q.clone(), k.clone()
            q[1:2], k[1:2] = q[:1], k[:1]
        q, k, v = (attn.head_to_batch_dim(t) for t in (q, k, v))
        p = torch.softmax(torch.einsum("bid,bjd->bij", q, k) * attn.scale, dim=-1)
        return attn.to_out[0](attn.batch_to_head_dim(torch.einsum("bij,bjd->bid", p, v)))


def register_time(model, t):
    model.unet._pnp_state.t = int(t)


def register_pnp(model, qk_schedule, conv_schedule):
    st = PnPState(qk_schedule, conv_schedule)
    unet = model.unet
    unet._pnp_state = st
    procs = {}
    for name in unet.attn_processors.keys():
        inject = False
        if ".attn1." in name and name.startswith("up_blocks."):
            res, blk = int(name.split(".")[1]), int(name.split(".")[3])
            inject = (res == 1 and blk in (1, 2)) or res in (2, 3)
        procs[name] = PnPSelfProcessor(st, inject)
    unet.set_attn_processor(procs)
    rb = unet.up_blocks[1].resnets[1]

    def forward(x, temb):
        h = rb.conv1(F.silu(rb.norm1(x))) + rb.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = rb.conv2(F.silu(rb.norm2(h)))
        if st.on(st.conv, h.shape[0]):
            h = h.clone()
            h[1:2] = h[:1]
        return (rb.conv_shortcut(x) if rb.conv_shortcut is not None else x) + h

    rb.forward = forward
