from dtt_amd.metrics import (SiSnrIPITInference, SiSnrIPITTraining, SiSnrPITInference,  # noqa: F401
                             SiSnrPITTraining)
