"""Host-side data contract around the hot path: device image pre-processing (f1), result formatting (f3) and
the streaming dataset / sampler semantics (f4).  Mirrors the reference's pipeline / dataset interfaces; arithmetic that touches
pixels runs in HIP kernels (far3d_amd/csrc/preproc.hip), the rest is small host logic."""
from . import av2_metric  # noqa: F401
from .preprocess import ImagePreprocessor, ida_matrix, sample_augmentation, sample_augmentation_portrait  # noqa: F401
from .resample import pil_resample_coeffs  # noqa: F401
from .results import box_to_av2, format_results, yaw_to_quat  # noqa: F401
from .streaming import StreamingIndex, contiguous_shard, sequence_group_flags  # noqa: F401
