"""`-t 1` does not know the batch size: on the full-size genome (24 contigs with hg38's lengths, 3.09 Gb, made in HBM) the reads and signals of ONE
32768-read batch -- the size bench.py's headline workload runs -- are those of two consecutive 16384-read batches of a second context and, over their
first 2048 reads, of two 1024-read batches of a third (the size tests/fullsize_hg38.py compares with the oracle read by read): 4.3e9 int16, every
dwell, offset and median, bit for bit.  One worker walks its reads in order whatever -K cuts them into (src/sim.c:559-641), so this is a
size-independent property that crosses every per-batch boundary of the implementation: links, slices, the hand-out's carried stream states, the
first event pass that rides along with the previous batch's hand-out.  Run by tests/test_config2_hg38.py in a process of its own."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import bench
    from squigulator_amd import api, model, profiles
    seq, lens = bench.synthetic_genome_device(None, dev)
    torch.cuda.synchronize()
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)

    def ctx():
        g = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
        g.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
        return g

    def take(b):
        out = (b.signal(), np.array(b.sig_off), b.dwell().copy(), np.array(b.ev_off), np.array(b.offset), np.array(b.median_before),
               {k: (np.frombuffer(bytes(v), np.uint8) if isinstance(v, (bytes, bytearray)) else np.array(v)) for k, v in dict(b.sampled).items()})
        b.free()
        return out

    big = ctx()
    sig, so, dw, eo, offs, meds, smp = take(big.sample(32768).run().wait())
    big.close()
    assert len(so) == 32769 and so[-1] == len(sig) and len(sig) > 4.0e9
    n_cmp = 0
    for K, n_batches in ((16384, 2), (1024, 2)):
        g = ctx()
        staged = [g.sample(K) for _ in range(n_batches)]             # (staged ahead, as the bench's timed region: the successor's first pass rides along)
        r0 = 0
        for b in staged:
            s, o, d, e, of, md, sm = take(b.run().wait())
            a0, a1 = int(so[r0]), int(so[r0 + K])
            assert a1 - a0 == len(s), (K, r0)
            np.testing.assert_array_equal(o, so[r0:r0 + K + 1] - so[r0])
            assert np.array_equal(sig[a0:a1], s), f"-K {K}: the signals of reads {r0}..{r0 + K - 1} differ from the 32768-read batch's"
            e0, e1 = int(eo[r0]), int(eo[r0 + K])
            assert np.array_equal(dw[e0:e1], d) and np.array_equal(eo[r0:r0 + K + 1] - eo[r0], e)
            assert np.array_equal(offs[r0:r0 + K], of) and np.array_equal(meds[r0:r0 + K], md)
            for key in ("ref_idx", "ref_pos", "rlen", "strand"):
                assert np.array_equal(smp[key][r0:r0 + K], sm[key]), key
            n_cmp += len(s)
            r0 += K
        g.close()
    print("batch-size ok: %d reads in one batch, %d samples compared with smaller batches" % (32768, n_cmp))


if __name__ == "__main__":
    main()
