from .arcface_model import IDLoss, Backbone  # noqa: F401
