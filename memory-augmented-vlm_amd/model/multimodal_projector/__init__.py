from .builder import build_vision_projector, IdentityMap, LinearProjector, MlpGeluProjector  # noqa: F401
