"""flope_amd -- MI355X-native flower 6-DoF pose inference (the wvu-irl/flope hot path).

Hand-written gfx950 HIP kernels behind a C-ABI (include/flope_amd.h); this package
is the thin Python host layer: ctypes binding (``_lib``), engine wrapper
(``engine.PoseEngine``), checkpoint helpers (``weights``), the data-parallel
driver (``distributed``) and the drop-in mirror of the reference's
``sunflower.*`` API (``flope_amd/sunflower``; put ``flope_amd/`` on PYTHONPATH to
import it as ``sunflower`` exactly like the reference's README.md:44 asks).
There is no CPU fallback: without the built library and a HIP device every
compute entry point raises.
"""
__version__ = "0.1.0"
