"""thor_amd - Python host mirror of libthor_hip.so (MI355X-native Thor per-block encode path).

PyTorch is not involved in the data path; this module is a thin ctypes binding of the C ABI
declared in include/thor_hip.h (sequence API + kernel-level entry points).  There is no CPU
fallback: importing works anywhere, but every call that computes requires the HIP library and a
gfx950 device and fails loudly otherwise.
"""
from .binding import (ThorParams, Encoder, lib, lib_path, load_config, sad_batch, interp_luma, code_tu_batch, deblock_frame,
                      build_native, REPO_ROOT)

__all__ = ['ThorParams', 'Encoder', 'lib', 'lib_path', 'load_config', 'sad_batch', 'interp_luma', 'code_tu_batch', 'deblock_frame',
           'build_native', 'REPO_ROOT']
