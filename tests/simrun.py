"""Run one reference-style command line through the ORACLE and format the outputs with the
product's text writers.  Test helper."""
from __future__ import annotations

import os

from squigulator_amd import aln_text, model, options, slow5_text
from squigulator_amd import profiles as P

import orc

INPUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")


def run_oracle(cmdline: str, model_override=None, nthreads=1):
    o = options.parse_args(cmdline)
    if model_override is not None:
        k, mean, stdv = model_override
    else:
        k = o.kmer_size_default
        mean, stdv = model.synthetic_model(k, meth=bool(o.meth_freq))
    orac = orc.Oracle(o.profile, o.flags, k, mean, stdv, o.seed, o.threads, o.rlen, o.amp_noise)
    ref = orac.load_ref(os.path.join(INPUTS, o.ref),
                        os.path.join(INPUTS, o.trans_count) if o.trans_count else None,
                        os.path.join(INPUTS, o.meth_freq) if o.meth_freq else None)
    names = [ref.names[i].decode() for i in range(ref.num_ref)]
    lengths = [ref.lengths[i] for i in range(ref.num_ref)]
    n = options.resolve_nreads(o, ref.num_ref, ref.sum)
    reads = orac.simulate(n, o.batch, want_ss=True, nthreads=nthreads)
    return o, k, names, lengths, reads, orac


def format_outputs(o, k, names, lengths, reads, mask_signal=False):
    out = {"slow5": [slow5_text.header(o.flags, o.profile.sample_rate)], "fasta": [], "paf": [],
           "sam": [aln_text.sam_header(names, lengths)]}
    for r in reads:
        rid = slow5_text.read_id(o.flags, r.read_number + 1, names[r.ref_idx], r.ref_pos_st, r.ref_pos_end, r.strand)
        out["slow5"].append(slow5_text.record(o.profile, o.flags, rid, r.offset, r.sig, r.median_before,
                                              r.read_number, r.start_time,
                                              sig_text="*" if mask_signal else None))
        seq = r.seq.decode()
        out["fasta"].append(f">{rid}\n{seq}\n")
        a = aln_text.Aln(o.flags, k, rid, names[r.ref_idx], r.ref_len, r.ref_pos_st, r.ref_pos_end, r.strand,
                         r.rlen, len(r.sig), r.ss)
        out["paf"].append(aln_text.paf_str(a))
        out["sam"].append(aln_text.sam_str(a, seq, names[r.ref_idx], r.ref_pos_st))
    return {k_: "".join(v) for k_, v in out.items()}


def mask_slow5_signal(text: str) -> str:
    lines = []
    for ln in text.splitlines(keepends=True):
        if ln.startswith(("#", "@")):
            lines.append(ln)
            continue
        c = ln.split("\t")
        c[7] = "*"
        lines.append("\t".join(c))
    return "".join(lines)
