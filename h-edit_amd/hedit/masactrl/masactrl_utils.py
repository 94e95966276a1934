"""Host side of the MasaCtrl attention editor -- mirrors text-guided/masactrl/masactrl_utils.py:
``AttentionBase`` (:6-33, the step / layer counters) and ``regiter_attention_editor_diffusers`` (:35-105,
spelling as in the reference).  The attention body the reference patches into every ``Attention.forward``
runs inside the HIP kernels; the editor object only compiles, per UNet call, the plan that tells the
self-attention kernel whose keys and values each batch row reads."""


class AttentionBase:
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def after_step(self):
        pass

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    # ---- protocol with hedit.engine / hedit.unet (same as the P2P controllers)
    def _plan(self, unet, B, H, W, save_attn):
        return None

    def _after_pass(self, save_attn):
        """one full UNet pass with the editor on = num_att_layers editor calls (masactrl_utils.py:15-24)"""
        self.cur_att_layer = 0
        self.cur_step += 1
        self.after_step()

    def step_callback(self, x_t):
        return x_t

    def __call__(self, *a, **k):
        raise RuntimeError("the attention editor is executed inside the HIP attention kernels; it cannot be called from Python")


def regiter_attention_editor_diffusers(model, editor):
    """Attach ``editor`` to ``model.unet`` (every UNet call without ``use_editor: False`` then runs under it)
    and set ``editor.num_att_layers`` to the number of attention layers (32 for SD-1.x)."""
    model.unet._attention_editor = editor
    editor.num_att_layers = len(model.unet.attn_processors)
