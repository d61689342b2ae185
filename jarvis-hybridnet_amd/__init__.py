"""MI355X-native implementation of the JARVIS-HybridNet multi-view inference
hot path (EfficientTrack 2D CNN -> ReprojectionLayer -> V2V 3D CNN ->
soft-argmax) behind the reference's own Python API.

All arithmetic on the path runs in hand-written HIP kernels for gfx950 that
live in `csrc/` and are reached through the C ABI declared in
`include/jarvis_hip.h` (`libjarvis_hip.so`).  There is no CPU fallback: every
entry point raises if the library cannot be loaded.
"""
__version__ = "0.1.0"
