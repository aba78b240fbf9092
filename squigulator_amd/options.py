"""Command-line option semantics of the simulator, as far as the signal path needs them.

Mirrors `sim_main` (src/sim.c:864-1064): `-x` is applied when it is seen, later
options override single profile fields, and dwell/bps/sample-rate are
reconciled after parsing (src/sim.c:1035-1045).  Only host logic -- no compute.
"""
from __future__ import annotations

import shlex
from dataclasses import dataclass, field
from typing import Optional

from . import profiles as P


def _c_round(x: float) -> float:
    """C round(): half away from zero."""
    import math
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


@dataclass
class SimOptions:
    ref: str = ""
    output: Optional[str] = None
    profile_name: str = "dna-r9-prom"
    profile: P.Profile = field(default_factory=lambda: P.get_profile("dna-r9-prom")[0])
    flags: int = 0
    nreads: int = 4000
    coverage: int = -1
    rlen: int = 10000
    seed: int = 0
    threads: int = 8
    batch: int = 1000
    amp_noise: float = 1.0
    model_file: Optional[str] = None
    fasta: Optional[str] = None
    paf: Optional[str] = None
    sam: Optional[str] = None
    trans_count: Optional[str] = None
    meth_freq: Optional[str] = None

    @property
    def kmer_size_default(self) -> int:
        return P.default_kmer_size(self.flags)


_LONG_WITH_ARG = {
    "--seed", "--dwell-mean", "--kmer-model", "--prefix", "--dwell-std", "--amp-noise",
    "--digitisation", "--sample-rate", "--range", "--offset-mean", "--offset-std", "--bps",
    "--median-before-mean", "--median-before-std", "--trans-count", "--trans-trunc",
    "--ont-friendly", "--meth-freq", "--meth-model", "--verbose", "--output", "--nreads",
    "--fasta", "--rlen", "--profile", "--threads", "--batchsize", "--paf", "--sam", "--coverage",
}
_LONG_NO_ARG = {"--ideal", "--full-contigs", "--ideal-time", "--ideal-amp", "--paf-ref", "--cdna"}
_SHORT_WITH_ARG = set("onqrxvKtcaf")


def _yes_no(o: SimOptions, flag: int, arg: str) -> None:
    if arg in ("yes", "y"):
        o.flags |= flag
    elif arg in ("no", "n"):
        o.flags &= ~flag


def parse_args(argv) -> SimOptions:
    """Parse a squigulator command line (list of tokens or one string, without argv[0])."""
    if isinstance(argv, str):
        argv = shlex.split(argv)
    o = SimOptions()
    given = set()
    pos = []
    i = 0
    toks = list(argv)
    items = []  # (name, value)
    while i < len(toks):
        t = toks[i]
        if t.startswith("--"):
            name, eq, val = t.partition("=")
            if name in _LONG_NO_ARG:
                # getopt_long accepts an optional-looking "=x" only for required_argument; ignore
                items.append((name, None))
            elif name in _LONG_WITH_ARG:
                if not eq:
                    # NB: the reference declares --trans-trunc as required_argument, so a bare
                    # `--trans-trunc` swallows the next token (scripts/test.sh:117 relies on it)
                    i += 1
                    val = toks[i]
                items.append((name, val))
            else:
                raise ValueError(f"unknown option {name}")
        elif t.startswith("-") and len(t) > 1:
            c = t[1]
            if c in _SHORT_WITH_ARG:
                if len(t) > 2:
                    val = t[2:]
                else:
                    i += 1
                    val = toks[i]
                items.append(("-" + c, val))
            else:
                raise ValueError(f"unknown option {t}")
        else:
            pos.append(t)
        i += 1

    for name, val in items:
        if name in ("-x", "--profile"):
            o.profile_name = val
            prof, fl = P.get_profile(val)
            o.profile = prof
            o.flags |= fl
        elif name in ("-o", "--output"):
            o.output = val
        elif name == "--ideal":
            o.flags |= P.SQ_IDEAL
        elif name == "--full-contigs":
            o.flags |= P.SQ_FULL_CONTIG
            given.add("full_contigs")
        elif name in ("-n", "--nreads"):
            o.nreads = int(val)
            given.add("n")
        elif name in ("-q", "--fasta"):
            o.fasta = val
        elif name in ("-r", "--rlen"):
            o.rlen = max(int(val), 200)
        elif name in ("-K", "--batchsize"):
            o.batch = int(val)
        elif name in ("-t", "--threads"):
            o.threads = int(val)
        elif name in ("-c", "--paf"):
            o.paf = val
        elif name in ("-a", "--sam"):
            o.sam = val
        elif name in ("-f", "--coverage"):
            o.coverage = int(val)
        elif name == "--seed":
            o.seed = int(val)
        elif name == "--ideal-time":
            o.flags |= P.SQ_IDEAL_TIME
        elif name == "--ideal-amp":
            o.flags |= P.SQ_IDEAL_AMP
        elif name == "--dwell-mean":
            given.add("dwell_mean")
            o.profile = o.profile.replace(dwell_mean=float(val))
        elif name == "--kmer-model":
            o.model_file = val
        elif name == "--prefix":
            _yes_no(o, P.SQ_PREFIX, val)
        elif name == "--dwell-std":
            o.profile = o.profile.replace(dwell_std=float(val))
        elif name == "--amp-noise":
            o.amp_noise = float(val)
        elif name == "--paf-ref":
            o.flags |= P.SQ_PAF_REF
        elif name == "--digitisation":
            o.profile = o.profile.replace(digitisation=float(val))
        elif name == "--sample-rate":
            given.add("sample_rate")
            o.profile = o.profile.replace(sample_rate=float(val))
        elif name == "--range":
            o.profile = o.profile.replace(range=float(val))
        elif name == "--offset-mean":
            o.profile = o.profile.replace(offset_mean=float(val))
        elif name == "--offset-std":
            o.profile = o.profile.replace(offset_std=float(val))
        elif name == "--bps":
            given.add("bps")
            o.profile = o.profile.replace(bps=float(val))
        elif name == "--median-before-mean":
            o.profile = o.profile.replace(median_before_mean=float(val))
        elif name == "--median-before-std":
            o.profile = o.profile.replace(median_before_std=float(val))
        elif name == "--trans-count":
            o.trans_count = val
        elif name == "--trans-trunc":
            _yes_no(o, P.SQ_TRANS_TRUNC, val)
        elif name == "--cdna":
            o.flags |= P.SQ_CDNA
        elif name == "--ont-friendly":
            _yes_no(o, P.SQ_ONT, val)
        elif name == "--meth-freq":
            o.meth_freq = val
            o.flags |= P.SQ_METH
        elif name in ("-v", "--verbose", "--meth-model"):
            pass

    if len(pos) != 1:
        raise ValueError("exactly one reference FASTA is expected")
    o.ref = pos[0]
    if (o.flags & P.SQ_CDNA) and (o.flags & (P.SQ_TRANS_TRUNC | P.SQ_PREFIX)):
        raise ValueError("--trans-trunc / --prefix are not implemented for --cdna")
    if (o.flags & P.SQ_CDNA) and (o.flags & P.SQ_RNA):
        raise ValueError("--cdna is only valid with DNA profiles")

    p = o.profile
    if "dwell_mean" in given:  # src/sim.c:1035-1038
        p = p.replace(bps=_c_round(p.sample_rate / p.dwell_mean))
    if "sample_rate" in given or "bps" in given:  # src/sim.c:1040-1045
        p = p.replace(dwell_mean=_c_round(p.sample_rate / p.bps))
    o.profile = p
    return o


def resolve_nreads(o: SimOptions, num_ref: int, ref_sum: int) -> int:
    """src/sim.c:1049-1061"""
    if o.flags & P.SQ_FULL_CONTIG:
        return num_ref
    if o.coverage > 0:
        if o.flags & P.SQ_RNA:
            return num_ref * o.coverage
        return (ref_sum * o.coverage) // o.rlen
    return o.nreads
