from dtt_amd.model import DTTNetModel, DTTNetVisionModel, DTTNetPlusModel  # noqa: F401

__all__ = ["DTTNetModel", "DTTNetVisionModel", "DTTNetPlusModel"]
