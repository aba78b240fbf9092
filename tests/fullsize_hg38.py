"""Full-size check of BASELINE.json configs[2] (run by tests/test_config2_hg38.py in a process of its own: torch makes the
3.09 Gb genome in HBM and has to initialise the HIP runtime before the library is loaded)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def main():
    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import bench
    import orc
    from squigulator_amd import api, model, profiles
    seq, lens = bench.synthetic_genome_device(None, dev)
    torch.cuda.synchronize()
    assert sum(lens) == 3088269832 and len(lens) == 24
    off = np.concatenate([[0], np.cumsum(lens)])
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    K = 1024
    sigs = {}
    for mode in (api.MODE_CERTIFIED, api.MODE_EXACT):
        gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=mode)
        gen.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
        keep = []
        for bi in range(2):
            b = gen.sample(K).run().wait()
            keep.append((b.signal().copy(), np.array(b.sig_off), dict(b.sampled), b.reads() if (mode == api.MODE_CERTIFIED) else None,
                         b.dwell().copy(), np.array(b.ev_off), np.array(b.offset), np.array(b.median_before)))
            b.free()
        gen.close()
        sigs[mode] = keep
    for (s0, o0, *_), (s1, o1, *_) in zip(sigs[api.MODE_CERTIFIED], sigs[api.MODE_EXACT]):
        np.testing.assert_array_equal(o0, o1)
        np.testing.assert_array_equal(s0, s1)
    # coordinates and sequences against the genome in HBM
    sig, so, smp, reads, dw, eo, offs, meds = sigs[api.MODE_CERTIFIED][0]
    assert len(reads) == K and (smp["rlen"] >= 200).all()
    seen = set()
    for i in range(0, K, 16):
        ci, pos, rl = int(smp["ref_idx"][i]), int(smp["ref_pos"][i]), int(smp["rlen"][i])
        seen.add(ci)
        assert 0 <= ci < 24 and smp["ref_len"][i] == lens[ci] and pos + rl <= lens[ci]
        g = bytes(seq[off[ci] + pos: off[ci] + pos + rl].cpu().numpy())
        assert g.count(b"N") * 10 <= rl                                  # src/genread.c:261-266
        want = g if chr(smp["strand"][i]) == "+" else g.translate(_COMP)[::-1]
        same = sum(a == b_ for a, b_ in zip(want, reads[i]))
        assert same >= rl - g.count(b"N")                                 # identical but for the substituted Ns
        assert len(reads[i]) == rl and b"N" not in reads[i]
    assert len(seen) >= 12
    # lengths: a read's samples are the sum of its dwells
    for i in range(K):
        assert so[i + 1] - so[i] == dw[eo[i]:eo[i + 1]].sum()
    # the whole first batch against the oracle (one worker: every stream carries over from read to read)
    orac = orc.Oracle(prof, fl, 9, mean, stdv, 42, num_workers=1)
    want = orac.run_batch_seqs(reads, want_ss=True)
    orac.close()
    for i, w in enumerate(want):
        np.testing.assert_array_equal(sig[so[i]:so[i + 1]], w.sig, err_msg=f"read {i}")
        np.testing.assert_array_equal(dw[eo[i]:eo[i + 1]], w.ss)
        assert offs[i] == w.offset and meds[i] == w.median_before
    print("full-size ok: %d reads, %d samples compared with the oracle" % (K, int(so[K])))


if __name__ == "__main__":
    main()
