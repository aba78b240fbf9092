"""Drive the HIP path (through the C ABI) over the same batches a reference command line implies."""
from __future__ import annotations

import numpy as np

from squigulator_amd import api, model, options


LAST_FALLBACK = 0


def run_hip_on_reads(cmdline, seqs, mode=api.MODE_EXACT, model_override=None, want_dwell=True):
    """seqs: the reads (bytes) in read order, as gen_read produced them.  Returns per-read dicts."""
    o = options.parse_args(cmdline)
    if model_override is not None:
        k, mean, stdv = model_override
    else:
        k = o.kmer_size_default
        mean, stdv = model.synthetic_model(k, meth=bool(o.meth_freq))
    gen = api.SignalGenerator(o.profile, o.flags, k, mean, stdv, o.seed, num_workers=o.threads,
                              amp_noise=o.amp_noise, mode=mode)
    global LAST_FALLBACK
    LAST_FALLBACK = 0
    out = []
    done, n = 0, len(seqs)
    start_time = 0
    while done < n:
        nb = min(o.batch, n - done)
        b = gen.stage(seqs[done:done + nb]).run().wait()
        LAST_FALLBACK += gen.timing()["fallback_samples"]
        sig = b.signal()
        dw = b.dwell() if want_dwell else None
        for i in range(nb):
            s = sig[b.sig_off[i]:b.sig_off[i + 1]]
            out.append(dict(sig=s, offset=b.offset[i], median=b.median_before[i], start_time=start_time,
                            ss=dw[b.ev_off[i]:b.ev_off[i + 1]] if want_dwell else None))
            start_time += len(s)
        b.free()
        done += nb
    gen.close()
    return out
