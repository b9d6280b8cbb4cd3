"""dtt_amd: MI355X-native (gfx950) DTTNet separation forward + backward.

Host side of libdtt_hip.so; mirrors the reference's model / loss / metric
classes (see ../src for the reference's own import paths)."""
from .model import DTTNetModel, DTTNetVisionModel, DTTNetPlusModel  # noqa: F401
from .loss import DTTNetLoss, PITLossWrapper, DTTNetLossWrapper  # noqa: F401
from .metrics import (SiSnr, SiSnrI, SiSnrPITTraining, SiSnrPITInference, SiSnrIPITTraining,  # noqa: F401
                      SiSnrIPITInference)
from .ops import stft_mag_phase, handoff_kernels, handoff_poll, handoff_check, handoff_spin_limit, wgrad_lane, join_wgrad_lane, video_lane  # noqa: F401
from ._lib import DttError  # noqa: F401
from .optim import ClipAdamW  # noqa: F401
from .io import (load_reference_checkpoint, save_reference_checkpoint, align_state_dict_keys,  # noqa: F401
                 write_wav_float32, save_separated)
from .feeder import collate_on_device, spec_augment_on_device  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from .dist import GradBucketReducer, ShardedClipAdamW, LocalGrads  # noqa: F401
