"""murcl_amd: MI355X-native hot path of MuRCL (MIL aggregators, NT-Xent, PPO sub-bag sampler).

Python mirrors of the reference's modules live in murcl_amd.models / murcl_amd.utils with the
reference's class names, constructor arguments, return tuples and state-dict keys; all tensor
math inside them runs in hand-written gfx950 HIP kernels loaded from libmurcl_amd.so
(C-ABI: include/murcl_amd.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
