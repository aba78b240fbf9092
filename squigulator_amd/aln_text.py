"""PAF / SAM emit of the signal-to-reference alignment of a simulated read.

The per-event dwell array (`ss`) comes straight from the signal generator; the
field layout follows the reference's formatter (src/format.c:53-122) and the
coordinates set up by the per-read worker (src/sim.c:576-596).
"""
from __future__ import annotations

from . import profiles as P

VERSION = "0.4.0-dirty"  # the reference version these goldens were cut from (src/version.h)

_COMP = {"A": "T", "a": "T", "C": "G", "c": "G", "G": "C", "g": "C", "T": "A", "t": "A"}


def reverse_complement(seq: str) -> str:
    """src/seq.h:78-112 (anything outside ACGTacgt complements to 'T')."""
    return "".join(_COMP.get(c, "T") for c in reversed(seq))


class Aln:
    """aln_t (src/format.h:8-27) filled as src/sim.c:576-596 does."""

    def __init__(self, flags, kmer_size, read_id, ref_name, ref_len, pos_st, pos_end, strand,
                 rlen, len_raw_signal, ss):
        rna = bool(flags & P.SQ_RNA)
        n_kmer = rlen - kmer_size + 1
        self.read_id = read_id
        self.len_raw_signal = len_raw_signal
        self.sig_start = 0
        self.sig_end = len_raw_signal
        self.strand = strand
        self.si_st_ref = pos_end - kmer_size + 1 if rna else pos_st
        self.si_end_ref = pos_st if rna else pos_end - kmer_size + 1
        if flags & P.SQ_PAF_REF:
            self.tid = ref_name
            self.tlen = (ref_len - kmer_size + 1) if not (flags & P.SQ_FULL_CONTIG) else n_kmer
            self.t_st, self.t_end = self.si_st_ref, self.si_end_ref
        else:
            self.tid = read_id
            self.tlen = n_kmer
            self.t_st, self.t_end = (n_kmer, 0) if rna else (0, n_kmer)
        self.ss = ss


def _ss_str(a: Aln) -> str:
    ss = a.ss[::-1] if a.t_st > a.t_end else a.ss
    return "".join(f"{int(v)}," for v in ss)


def paf_str(a: Aln) -> str:
    block = abs(a.t_end - a.t_st)
    return (f"{a.read_id}\t{a.len_raw_signal}\t{a.sig_start}\t{a.sig_end}\t{a.strand}\t"
            f"{a.tid}\t{a.tlen}\t{a.t_st}\t{a.t_end}\t{block}\t{block}\t255\t"
            f"sc:f:{1.0:f}\tsh:f:{0.0:f}\tss:Z:{_ss_str(a)}\n")


def sam_header(ref_names, ref_lengths) -> str:
    out = "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(ref_names, ref_lengths))
    return out + f"@PG\tID:squigulator\tPN:squigulator\tVN:{VERSION}\n"


def sam_str(a: Aln, seq: str, rname: str, ref_pos_st: int) -> str:
    flag = 0 if a.strand == "+" else 16
    body = seq if a.strand == "+" else reverse_complement(seq)
    return (f"{a.read_id}\t{flag}\t{rname}\t{ref_pos_st + 1}\t255\t{len(seq)}M\t*\t0\t0\t{body}\t*\t"
            f"si:Z:{a.sig_start},{a.sig_end},{a.si_st_ref},{a.si_end_ref}\tss:Z:{_ss_str(a)}\n")
