from .pnp_utils import (get_timesteps, register_attention_control_efficient, register_conv_control_efficient,  # noqa: F401
                        register_time)
