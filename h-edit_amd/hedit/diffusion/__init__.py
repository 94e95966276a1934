from .diffusion import Model, CELEBA_HQ_CONFIG, TINY_DDPM_CONFIG  # noqa: F401
