from dtt_amd.loss import DTTNetLoss, PITLossWrapper, DTTNetLossWrapper  # noqa: F401

__all__ = ["DTTNetLoss", "PITLossWrapper", "DTTNetLossWrapper"]
