from .masactrl import MutualSelfAttentionControl  # noqa: F401
from .masactrl_utils import AttentionBase, regiter_attention_editor_diffusers  # noqa: F401
