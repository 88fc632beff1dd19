"""viditq_amd - MI355X-native quantized-DiT denoising path behind the qdiff operator API.

Layout (only what the hot path needs):
  csrc/      hand-written gfx950 HIP kernels + the C ABI (libviditq_hip.so, include/viditq.h)
  build.py   hipcc driver (in-tree build; no torch extension machinery)
  _lib.py    ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py     tensor-level wrappers (device pointers, current HIP stream)
  qdiff/     host-side mirror of the reference operator API
             (quantizer/, models/: QuantLayer family, QuantModel, QuantAttention)
  t2v/       STDiT block / model forward and the IDDPM-DDIM sampling loop
  t2i/       PixArt block / model forward
  shard.py   prompt sharding over the GPUs of a node + RCCL broadcast of packed weights
"""
__version__ = "0.1.0"
