"""elasticfusion_amd — MI355X-native (gfx950, hand-written HIP) engine for ElasticFusion's per-frame hot path.

Layout:
  csrc/        HIP kernels + the C ABI implementation (libefusion_hip.so, built in-tree by build.py)
  api.py       host-side mirror of the reference interface (class ElasticFusion, operator tier) over the C ABI
  synth.py     synthetic RGB-D sequence generator (inputs only)
  build.py     hipcc driver

Importing the package is cheap and never loads the library; ``api.lib()`` does, and raises if it is missing.
"""
__all__ = ["api", "synth", "build"]
