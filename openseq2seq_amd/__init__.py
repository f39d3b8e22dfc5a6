"""openseq2seq_amd — MI355X (gfx950) native hot path for OpenSeq2Seq.

The compute path is hand-written HIP behind the C ABI in include/os2s.h
(libos2s_hip.so); this package is the Python host layer that mirrors the
reference's Encoder / Decoder / Loss / optimizer plugin interface on top of it.
PyTorch is used for device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
