"""vitpe -- MI355X-native host side of the ViT attention-with-positional-encoding hot path.

Layout:
  _lib.py      ctypes binding of libvitpe.so (prototypes parsed from include/vitpe.h)
  kernels.py   tensor-level wrappers, one per C-ABI entry point
  ops.py       torch.library custom ops (`torch.ops.vitpe.*`) with registered backward
  positional_encoding.py / rope_utils.py / vit.py
               drop-in mirrors of the reference's models/ package (same class names,
               constructor arguments, attribute and state_dict surface)
  engine.py    the train-step engine: flat parameter/gradient buffers, captured HIP graph,
               fused AdamW, RCCL gradient all-reduce (replaces train.py:94-125)
"""
__version__ = "0.1.0"
