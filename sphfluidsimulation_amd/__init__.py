"""MI355X-native engine for the per-frame particle update of leandro-barcelos/SPHFluidSimulation.

Only the hot path lives here: HIP kernels + C ABI under ``csrc/`` (built to
``lib/libsph_hip.so``), the host-side mirror of the reference driver
(``simulation.SphFluidSimulation``), the z-slab multi-GPU driver (``distributed``) and
the synthetic dam-break input generator (``presets``).
"""
__version__ = "0.1.0"
