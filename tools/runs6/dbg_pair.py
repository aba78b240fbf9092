"""debug: the pair variant against the base library on the same reads; for every differing sample: where it sits"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from squigulator_amd import api, model, profiles

def run(lib, prof, fl, k, seqs, T, seed, workers=None):
    mean, stdv = model.synthetic_model(k)
    g = api.SignalGenerator(prof, fl, k, mean, stdv, seed, num_workers=T, mode=api.MODE_CERTIFIED, lib_path=lib)
    b = g.stage(seqs, workers).run().wait()
    out = (b.signal().copy(), np.array(b.sig_off), b.dwell().copy(), np.array(b.ev_off))
    b.free(); g.close()
    return out

A = os.path.join(ROOT, "tools", "var_a_base.so"); B = os.path.join(ROOT, "tools", "var_b_pair.so")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for name, pn, k, extra in (("r9", "dna-r9-prom", 6, 0), ("r10", "dna-r10-prom", 9, 0), ("rna004", "rna004-prom", 9, profiles.SQ_PREFIX)):
    prof, fl = profiles.get_profile(pn)
    fl |= extra
    for T in (1, 8):
        n = 64
        seqs = [bytes(rng.choice(list(b"ACGT"), size=int(m)).astype(np.uint8)) for m in rng.integers(250, 5000, size=n)]
        wk = (np.arange(n) * T // n).astype(np.int32)
        sa, oa, da, ea = run(A, prof, fl, k, seqs, T, 7, wk)
        sb, ob, db, eb = run(B, prof, fl, k, seqs, T, 7, wk)
        assert np.array_equal(oa, ob) and np.array_equal(da, db)
        bad = np.nonzero(sa != sb)[0]
        print(f"{name} T={T}: {len(sa)} samples, {len(bad)} differ")
        for x in bad[:12]:
            r = int(np.searchsorted(oa, x, side="right") - 1)
            pos = int(x - oa[r])
            dw = da[ea[r]:ea[r + 1]].astype(np.int64)
            if fl & profiles.SQ_RNA:
                gpos = int(oa[r + 1] - oa[r]) - 1 - pos          # generation index within the read
            else:
                gpos = pos
            cs = np.concatenate([[0], np.cumsum(dw)])
            e = int(np.searchsorted(cs, gpos, side="right") - 1)
            j = gpos - int(cs[e])
            print(f"   abs {int(x)} (parity {int(x) & 1}) read {r} gen-pos {gpos} event {e} (item {e // 256}, ev-in-item {e % 256}) j {j} of sps {int(dw[e])}; prev sps {int(dw[e-1]) if e else -1} next sps {int(dw[e+1]) if e + 1 < len(dw) else -1}; got {int(sb[x])} want {int(sa[x])}")
