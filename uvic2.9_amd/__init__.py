"""uvic2.9_amd -- MI355X-native ocean tracer time-step for UVic ESCM 2.9.

Only the hot path of SURVEY.md §8 lives here: hand-written HIP kernels for
gfx950 behind a C ABI (include/uvic_gpu.h) plus the host-side mirror of the
reference's `tracer`/`isopyc` entry points.  The directory name contains a dot,
so the package is imported through the top-level alias module `uvic29_amd`.
"""
from .configs import OPTION_SETS, OptionSet, performance_set  # noqa: F401

__all__ = ["OPTION_SETS", "OptionSet", "performance_set"]
