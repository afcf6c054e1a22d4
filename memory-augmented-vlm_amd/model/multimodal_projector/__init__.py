from .builder import build_vision_projector, IdentityMap, MlpGeluProjector  # noqa: F401
