from .utils import dataset_from_json, image_grid, load_512, tensor_to_pil  # noqa: F401
