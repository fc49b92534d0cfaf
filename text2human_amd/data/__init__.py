"""Dataset / file formats on the input side of the sampling path:
the DeepFashion-MultiModal directory layout the reference's data loaders read."""
from .deepfashion import DeepFashionAttrPoseDataset, DeepFashionAttrSegmDataset  # noqa: F401
