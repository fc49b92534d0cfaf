"""`create_model(opt)` factory -- drop-in for the reference's
models/__init__.py:21-42 restricted to the two sampling model types and the
inference subset of the hierarchy (encode side) model."""
import logging

from .hierarchy_model import VQGANTextureAwareSpatialHierarchyInferenceModel
from .transformer_model import TransformerTextureAwareModel
from .sample_model import (BaseSampleModel, SampleFromParsingModel,  # noqa: F401
                           SampleFromPoseModel)

_MODELS = {
    'SampleFromParsingModel': SampleFromParsingModel,
    'SampleFromPoseModel': SampleFromPoseModel,
    'VQGANTextureAwareSpatialHierarchyInferenceModel': VQGANTextureAwareSpatialHierarchyInferenceModel,
    'TransformerTextureAwareModel': TransformerTextureAwareModel,
}


def create_model(opt, state_dicts=None):
    """models/__init__.py:21-42.  state_dicts (not in the reference): module -> state_dict already in host memory
    (the multi-GPU entry point reads the `.pth` files once, on rank 0, and broadcasts them) instead of reading the
    files named by `opt`."""
    model_type = opt['model_type']
    model_cls = _MODELS.get(model_type)
    if model_cls is None:
        raise ValueError(f'Model {model_type} is not found.')
    model = model_cls(opt) if state_dicts is None else model_cls(opt, state_dicts=state_dicts)
    logging.getLogger('base').info(f'Model [{model.__class__.__name__}] is created.')
    return model
