"""far3d_amd -- MI355X-native Far3D inference hot path (HIP kernels behind a C ABI + a Python host
mirror of the reference's mmdet3d_plugin registry surface).  See DESIGN.md."""
__version__ = "0.1.0"
