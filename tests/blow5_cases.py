"""Command lines for which the compiled reference wrote golden BLOW5 files (tools/make_blow5_golden.py -> tests/golden/blow5)."""
BLOW5_CASES = [
    ("r9_t1", "nCoV-2019.reference.fasta -x dna-r9-prom -n 6 --seed 42 -r 800 -t1"),
    ("r10_t1", "nCoV-2019.reference.fasta -x dna-r10-prom -n 5 --seed 42 -r 700 -t1"),
    ("rna004_prefix", "rnasequin_sequences_2.4.fa -x rna004-prom -n 2 --seed 42 -t1 --prefix=yes"),
    ("rna9", "rnasequin_sequences_2.4.fa -x rna-r9-prom -n 2 --seed 5 -t1"),
    ("r9_ont", "nCoV-2019.reference.fasta -x dna-r9-min -n 4 --seed 42 -r 600 -t1 --ont-friendly yes"),
    ("r9_two_batches", "nCoV-2019.reference.fasta -x dna-r9-prom -n 7 --seed 9 -r 500 -t1 -K 3"),
]
