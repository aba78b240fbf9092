"""Cases for which the COMPILED REFERENCE path produced committed vectors (tools/make_refvec.py).

All use the synthetic pore model of squigulator_amd.model.synthetic_model(k) with k the profile's
default; command lines use the reference's option syntax.  Regimes: -t1, and -t T -K T (one read
per worker per batch => the reference's work stealing never fires).
"""
REFVEC_CASES = [
    ("r9_t1", "nCoV-2019.reference.fasta -x dna-r9-prom -n 8 --seed 42 -r 1000 -t1"),
    ("r9_ideal", "nCoV-2019.reference.fasta -x dna-r9-prom -n 3 --seed 42 -r 1000 -t1 --ideal"),
    ("r9_ideal_time", "nCoV-2019.reference.fasta -x dna-r9-prom -n 3 --seed 42 -r 1000 -t1 --ideal-time"),
    ("r9_ideal_amp", "nCoV-2019.reference.fasta -x dna-r9-prom -n 3 --seed 42 -r 1000 -t1 --ideal-amp"),
    ("r9_amp_noise", "nCoV-2019.reference.fasta -x dna-r9-prom -n 4 --seed 42 -r 1000 -t1 --amp-noise 0.5"),
    ("r9_prefix", "nCoV-2019.reference.fasta -x dna-r9-prom -n 4 --seed 42 -r 1000 -t1 --prefix=yes"),
    ("r9_tk16", "nCoV-2019.reference.fasta -x dna-r9-prom -n 40 --seed 42 -r 600 -t 16 -K 16"),
    ("r9min_t1", "nCoV-2019.reference.fasta -x dna-r9-min -n 4 --seed 7 -r 1000 -t1"),
    ("rna9_noprefix", "rnasequin_sequences_2.4.fa -x rna-r9-prom -n 3 --seed 42 -t1"),
    ("rna9_prefix", "rnasequin_sequences_2.4.fa -x rna-r9-prom -n 3 --seed 42 -t1 --prefix=yes"),
    ("rna004_noprefix", "rnasequin_sequences_2.4.fa -x rna004-prom -n 3 --seed 42 -t1"),
    ("rna004_prefix", "rnasequin_sequences_2.4.fa -x rna004-prom -n 3 --seed 42 -t1 --prefix=yes"),
    ("rna004_tk4", "rnasequin_sequences_2.4.fa -x rna004-prom -n 8 --seed 42 -t 4 -K 4 --prefix=yes"),
    ("r10_t1", "nCoV-2019.reference.fasta -x dna-r10-prom -n 5 --seed 42 -r 1000 -t1"),
    ("r10_tk8", "nCoV-2019.reference.fasta -x dna-r10-prom -n 16 --seed 42 -r 600 -t 8 -K 8"),
    ("cdna_tc", "rnasequin_sequences_2.4.fa -x dna-r10-min -n 3 --seed 3 -t1 --cdna --trans-count sequin_count.tsv"),
    # CpG methylation (--meth-freq): the 5-letter 5^k table; mfreq.tsv is the reference's own 5-line test file, mfreq_dense.tsv
    # a frequency for every fourth CpG of the genome (tools/make_methfreq.py)
    ("r9_meth", "nCoV-2019.reference.fasta -x dna-r9-prom -n 3 --seed 1 -r 4000 -t1 --meth-freq mfreq.tsv"),
    ("r9_meth_dense", "nCoV-2019.reference.fasta -x dna-r9-prom -n 6 --seed 5 -r 1500 -t1 --meth-freq mfreq_dense.tsv"),
    ("r9_meth_tk4", "nCoV-2019.reference.fasta -x dna-r9-prom -n 8 --seed 11 -r 1200 -t 4 -K 4 --meth-freq mfreq_dense.tsv"),
    ("r10_meth_dense", "nCoV-2019.reference.fasta -x dna-r10-prom -n 3 --seed 2 -r 1200 -t1 --meth-freq mfreq_dense.tsv"),
]
