"""ohm_amd -- MI355X-native GPU ray integration for ohm occupancy maps (GpuMap / GpuNdtMap / GpuTsdfMap path only).

The compute path is libohmhip.so (hand-written HIP for gfx950) behind the C ABI in include/ohmhip.h.  This package
is the thin host-side mirror of the reference's interface used by the tests and the benchmark harness.
"""
from ._lib import OhmHipError, LIB_PATH, EXPORTED_SYMBOLS  # noqa: F401
from .gpumap import (GpuMap, GpuNdtMap, GpuTransformSamples, GpuTsdfMap, LineKeysQueryGpu, NdtMode, OccupancyMap, RayFlag,  # noqa: F401
                     RayMapper, LAYERS,
                     device_count, device_info, probability_to_value, value_to_probability)
