"""magicdrive_amd — MI355X-native drop-in for MagicDrive's multi-view diffusion sampler hot path.

See DESIGN.md.  The compute path is the hand-written HIP library `libmdx.so` (include/mdx.h);
this package is the host-side mirror of the reference's pipeline / network API.
"""
__version__ = "0.1.0"
