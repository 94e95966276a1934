from .utils import image_grid, load_512, tensor_to_pil  # noqa: F401
