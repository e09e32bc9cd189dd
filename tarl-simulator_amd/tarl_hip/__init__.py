"""tarl_hip — Python binding of libtarl_hip.so (hand-written HIP kernels for gfx950 behind a C ABI, include/tarl_hip.h).

Importing the package never touches the GPU; ``tarl_hip.lib.load()`` loads the shared library and raises if it is
missing (there is no CPU fallback anywhere in the product path).
"""
