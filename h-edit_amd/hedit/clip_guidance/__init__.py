from .base_clip import CLIPEncoder

__all__ = ["CLIPEncoder"]
