#!/bin/bash
# round 6, the record of the two-samples-per-lane loop (-DSQG_LEAN_PAIR=1): parity of the variant through SQG_LIB (the config tests and the parity files),
# twelve alternating repetitions of the timed region (a: the committed loop; b: the pair loop; d: the pair loop without its last phase, timing only),
# then three counter passes for a and b
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6e; mkdir -p $OUT
# the three variant libraries (development builds; hipcc cross-compiles them anywhere): a = committed loop, b = pair loop, d = pair loop without its last phase
python - <<'PY'
import os
from squigulator_amd import build
for out, extra in (("tools/var_a_base.so", []), ("tools/var_b_pair.so", ["-DSQG_LEAN_PAIR=1"]), ("tools/var_d_nopost.so", ["-DSQG_LEAN_PAIR=1", "-DSQG_PAIR_ABL=2"])):
    if not os.path.exists(out) or build.stamped_hash(out) != build.source_hash():
        build.build_variant(out, dev=True, extra=extra, verbose=False)
PY
SQG_LIB=$PWD/tools/var_b_pair.so timeout 1500 python -m pytest tests/test_00_configs.py tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py tests/test_sampler.py tests/test_long_reads.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.log
REPS=12 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
for v in a_base b_pair; do
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    echo "== $v: $C"; bash tools/pmc_quick.sh "$C" --lib $PWD/tools/var_$v.so --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 2>&1 | grep k_samples_lean
  done
done 2>&1 | tee $OUT/pmc.log
