"""MI355X-native engine for the VLMo pretraining forward/backward path of
fanzhongyi/ExploreMultiModal (models/vlmo, dall_e encoder).  Host side mirrors
the reference's Python interface; compute is hand-written gfx950 HIP behind the
C-ABI of include/vlmo_hip.h (exploremultimodal_amd/lib/libvlmo_hip.so)."""
__version__ = '0.1.0'

import os as _os

# The engine runs the dgrad chain, the weight-gradient GEMMs and the RCCL reductions on three HIP streams.
# ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with RCCL's own streams); when
# two of ours alias one queue they serialise (measured: +3 ms per VLMo-Base step).  Must be set before the
# HIP runtime initialises, i.e. before the first CUDA/HIP call of the process.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
