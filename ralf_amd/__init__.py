"""ralf_amd -- MI355X-native (gfx950) implementation of the RALF hot path.

Host side mirrors the reference's operator interface for the path; all device
arithmetic goes through the C ABI of libralf_hip.so (include/ralf_hip.h).
"""
__version__ = "0.1.0"
