"""Parameter presets (`-x`) and option-flag bits of the per-read signal path.

Mirrors the reference's `profile_t` literals and `set_profile`
(src/sim.c:55-188) and the `opt_t.flag` bit values (src/sq.h:32-42).  These are
data constants of the domain; the kernel's config struct (`sqg_profile_t` in
include/sqg.h) carries the same ten doubles in the same order.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict, replace

# opt_t.flag bits, src/sq.h:32-42
SQ_RNA = 0x001
SQ_FULL_CONTIG = 0x002
SQ_IDEAL = 0x004
SQ_IDEAL_TIME = 0x008
SQ_IDEAL_AMP = 0x010
SQ_PREFIX = 0x020
SQ_R10 = 0x040
SQ_PAF_REF = 0x080
SQ_TRANS_TRUNC = 0x100
SQ_CDNA = 0x200
SQ_ONT = 0x400
# not an opt_t.flag bit: the reference keys CpG methylation on opt.meth_freq != NULL (src/sim.c:231,297, src/gensig.c:231,251);
# here it is a flag of the context: 5-letter (A C G M T) pore table of 5^k rows, ranks of src/seq.h:45-74
SQ_METH = 0x1000
# implementation option of this library (include/sqg.h SQG_ORDER_FREE), no effect on results: the few-worker stream hand-out by
# the order-free kernels (claim protocol, lane masks) instead of the lane-ordered LDS atomics
SQ_ORDER_FREE = 0x2000


@dataclass(frozen=True)
class Profile:
    """profile_t, src/sq.h:47-58 (field order is the C-ABI order)."""
    digitisation: float
    sample_rate: float
    bps: float
    range: float
    offset_mean: float
    offset_std: float
    median_before_mean: float
    median_before_std: float
    dwell_mean: float
    dwell_std: float

    def as_tuple(self):
        return tuple(asdict(self).values())

    def replace(self, **kw) -> "Profile":
        return replace(self, **kw)


# src/sim.c:55-150
_PRESETS = {
    "dna-r9-min": (Profile(8192, 4000, 450, 1443.030273, 13.7222605, 10.25279688,
                           200.815801, 20.48933762, 9.0, 4.0), 0),
    "dna-r9-prom": (Profile(2048, 4000, 450, 748.5801, -237.4102, 14.1575,
                            214.2890337, 18.0127916, 9.0, 4.0), 0),
    "rna-r9-min": (Profile(8192, 3012, 70, 1126.47, 4.65491888, 4.115262472,
                           242.6584118, 10.60230888, 43.0, 35.0), SQ_RNA),
    "rna-r9-prom": (Profile(2048, 3000, 70, 548.788269, -231.9440589, 12.87185278,
                            238.5286796, 21.1871794, 43.0, 35.0), SQ_RNA),
    "dna-r10-prom": (Profile(2048, 5000, 400, 281.345551, -127.5655735, 19.377283387665,
                             189.87607393756, 15.788097978713, 13.0, 4.0), SQ_R10),
    "dna-r10-min": (Profile(8192, 5000, 400, 1536.598389, 13.380569389019, 16.311471649012,
                            202.15407438804, 13.406139241768, 13.0, 4.0), SQ_R10),
    "rna004-prom": (Profile(2048, 4000, 130, 299.432068, -259.421128, 16.010841823643,
                            205.63935594369, 8.3994882799157, 31.0, 0.0), SQ_R10 | SQ_RNA),
    "rna004-min": (Profile(8192, 4000, 130, 1437.976685, 12.47686423863, 10.442126577137,
                           205.08496731088, 8.6671292866233, 31.0, 0.0), SQ_R10 | SQ_RNA),
}


def get_profile(name: str):
    """`set_profile`, src/sim.c:152-188 -> (Profile, flag bits it sets)."""
    if name not in _PRESETS:
        raise ValueError(f"Unknown profile: {name}")
    return _PRESETS[name]


def default_kmer_size(flags: int) -> int:
    """k of the built-in table `init_core` would pick (src/sim.c:272-291, src/model.c:151-184)."""
    if flags & SQ_R10:
        return 9
    return 5 if flags & SQ_RNA else 6


def profile_names():
    return list(_PRESETS)
