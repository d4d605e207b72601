"""MI355X-native engine for the VLMo pretraining forward/backward path of
fanzhongyi/ExploreMultiModal (models/vlmo, dall_e encoder).  Host side mirrors
the reference's Python interface; compute is hand-written gfx950 HIP behind the
C-ABI of include/vlmo_hip.h (exploremultimodal_amd/lib/libvlmo_hip.so)."""
__version__ = '0.1.0'
