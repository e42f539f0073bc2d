"""splat_amd -- MI355X-native drop-in for the rasterisation hot path of thomasantony/splat.

Host-side mirror of the reference's operator interface (`Camera`, `GaussianList`,
`GaussianSplatPipeline01/02.render_to_buffer`) over the C ABI of libsplat_hip.so
(include/splat_hip.h), whose kernels are hand-written HIP for gfx950.  There is no CPU
fallback: anything that computes needs the built library and a GPU, and says so loudly.
"""
from .camera import Camera  # noqa: F401
from .gaussians import (GaussianList, naive_gaussians, load_from_ply, synthetic_scene, synthetic_surface_scene,  # noqa: F401
                        synthetic_raw, synthetic_surface_raw, write_ply, trim_ply)
from .pipelines import GaussianSplatPipeline01, GaussianSplatPipeline02  # noqa: F401
from .renderer import Renderer, SplatError  # noqa: F401
from .multi import MultiRenderer, slab_partition_native  # noqa: F401

MODE_EXACT = 0                  # SPLAT_MODE_EXACT: the reference's arithmetic
MODE_CORRECTED_PROJECTION = 1   # SPLAT_MODE_CORRECTED_PROJECTION: EWA Jacobian with its shear terms (not the reference)
MODE_FAST = 4                   # SPLAT_MODE_FAST: every colour byte within 1 of the exact frame's, by construction; ~1/3 less compositor work
MODE_LIBM_EXP = 2               # SPLAT_MODE_LIBM_EXP: fragment()'s exp as glibc's expf computes it (bit-exact frames, slower)
