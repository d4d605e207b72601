"""Model factory with the reference's entry point (models/build.py:4-12): ``build_model(config)`` dispatches on
``config.model.type`` and raises NotImplementedError for anything but 'VLMO'."""
from .vlmo_module import VlmoModule

# model type -> constructor taking the whole config node
_REGISTRY = {'VLMO': VlmoModule}


def build_model(config):
    kind = config.model.type
    ctor = _REGISTRY.get(kind)
    if ctor is None:
        raise NotImplementedError(f'Unkown model: {kind}')       # message spelled as upstream: callers grep it
    return ctor(config=config)
