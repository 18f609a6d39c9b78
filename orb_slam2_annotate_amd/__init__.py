"""MI355X-native ORB front-end: the ORBextractor / ORBmatcher hot path of ORB-SLAM2 as
hand-written HIP (gfx950) behind a C-ABI (include/orbfe.h).  This package is the host-side
mirror of the reference's class interface; it binds liborbfe.so with ctypes and has no
compute of its own (and no CPU fallback)."""
from ._lib import KP_DTYPE, OrbfeError, LIB_PATH  # noqa: F401
from .extractor import ORBextractor, gaussian_blur7, resize_linear, set_blur_pass_order  # noqa: F401
from .matcher import ComputeStereoMatches, FeatureVector, FrameView, ORBmatcher, ResidentFrame  # noqa: F401
from .vocabulary import ORBVocabulary, synthetic_vocabulary_arrays, write_synthetic_vocabulary, write_vocabulary_text  # noqa: F401
from .ingest import (ComputeDistinctiveDescriptors, ComputeImageBounds, ComputeStereoFromRGBD,  # noqa: F401
                     Rectifier, UndistortKeyPoints, cvtColorToGray, initUndistortRectifyMap, undistortPoints)  # noqa: F401
