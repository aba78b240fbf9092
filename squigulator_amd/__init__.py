"""squigulator_amd -- MI355X-native per-read nanopore signal generator (hot path of squigulator).

The product is the C-ABI shared library (include/sqg.h, squigulator_amd/csrc/libsqg_hip.so);
this package holds the build recipe, a ctypes binding for tests/bench, and the host-side
mirrors of the reference's option/profile/model/text-format logic.
"""
from . import profiles, model, options, slow5_text, aln_text  # noqa: F401

__all__ = ["profiles", "model", "options", "slow5_text", "aln_text"]
