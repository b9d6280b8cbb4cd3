from dtt_amd.model import DTTNetModel  # noqa: F401

__all__ = ["DTTNetModel"]
