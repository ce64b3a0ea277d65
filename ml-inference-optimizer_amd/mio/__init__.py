"""mio -- MI355X-native transformer-inference hot path (FA3 attention, FusedMLP, ring attention,
tensor-parallel all-reduce) behind the reference's operator surface.

Importing the kernel-backed modules loads libmio_hip.so through ctypes (mio._lib); there is no
CPU / PyTorch fallback for the compute path.
"""
__version__ = "0.1.0"
