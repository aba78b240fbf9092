"""ASCII SLOW5 emit for simulated reads (the text twin of the BLOW5 the reference writes).

Layout follows the reference's record/header setters (src/gensig.c:40-223) and
slow5lib's ASCII encoder (slow5lib/src/slow5.c:3824-3925; doubles through
slow5_double_to_str, slow5lib/src/slow5_misc.c:379-405: "%f" with trailing
zeros trimmed).  Header attributes are emitted in the sorted order slow5lib
uses (see test/slow5.exp:3-9).
"""
from __future__ import annotations

from . import profiles as P

_END_REASON = ("unknown", "partial", "mux_change", "unblock_mux_change",
               "data_service_unblock_mux_change", "signal_positive", "signal_negative")


def double_to_str(x: float) -> str:
    s = "%f" % x
    if "." in s:
        s = s.rstrip("0")
        if s.endswith("."):
            s = s[:-1]
            if s == "-0":
                s = "0"
    return s


def header(flags: int, sample_rate: float) -> str:
    rna = bool(flags & P.SQ_RNA)
    r10 = bool(flags & P.SQ_R10)
    ont = bool(flags & P.SQ_ONT)
    kit = ("sqk-rna004" if r10 else "sqk-rna002") if rna else ("sqk-lsk114" if r10 else "sqk-lsk109")
    attrs = {
        "asic_id": "asic_id_0",
        "exp_start_time": "2022-07-20T00:00:00Z",
        "experiment_type": "rna" if rna else "genomic_dna",
        "flow_cell_id": "FAN00000",
        "run_id": "run_0",
        "sample_frequency": str(int(sample_rate)),
        "sequencing_kit": kit,
    }
    lines = ["#slow5_version\t0.2.0", "#num_read_groups\t1"]
    lines += [f"@{k}\t{attrs[k]}" for k in sorted(attrs)]
    types = ["char*", "uint32_t", "double", "double", "double", "double", "uint64_t", "int16_t*",
             "char*", "double", "int32_t", "uint8_t", "uint64_t"]
    names = ["read_id", "read_group", "digitisation", "offset", "range", "sampling_rate",
             "len_raw_signal", "raw_signal", "channel_number", "median_before", "read_number",
             "start_mux", "start_time"]
    if ont:
        types.append("enum{" + ",".join(_END_REASON) + "}")
        names.append("end_reason")
    lines.append("#" + "\t".join(types))
    lines.append("#" + "\t".join(names))
    return "\n".join(lines) + "\n"


def read_id(flags: int, read_number_1based: int, ref_name: str, pos_st: int, pos_end: int, strand: str) -> str:
    """src/sim.c:566-570 / fake_uuid src/sim.c:498-504"""
    if flags & P.SQ_ONT:
        return "00000000-0000-0000-0000-%012d" % read_number_1based
    return "S1_%d!%s!%d!%d!%s" % (read_number_1based, ref_name, pos_st, pos_end, strand)


def record(profile, flags: int, rid: str, offset: float, sig, median_before: float,
           read_number: int, start_time: int, sig_text: str | None = None) -> str:
    if sig_text is None:
        sig_text = ",".join(map(str, sig.tolist() if hasattr(sig, "tolist") else sig))
    cols = [rid, "0", double_to_str(profile.digitisation), double_to_str(offset),
            double_to_str(profile.range), double_to_str(profile.sample_rate), str(len(sig)), sig_text,
            "0", double_to_str(median_before), str(read_number), "0", str(start_time)]
    if flags & P.SQ_ONT:
        cols.append("0")
    return "\t".join(cols) + "\n"
