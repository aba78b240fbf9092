#!/bin/bash
# rocprofv3 passes for the bench: kernel trace + stats, then PMC groups (each in its own pass)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
shift || true
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 $*"
# the stats pass runs the bench exactly as the driver does (defaults; CPU legs included): its kernel averages are the ones
# the bench line's roofline.kernel_ms has to agree with
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $GRAFT_REPO_ROOT/bench.py $* > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
ls -R $OUT | head -30
# HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots); the WRITE pass also runs the
# bench's store probe (k_store_probe writes a known 1 GiB per launch) to calibrate WRITE_SIZE units
BENCH2="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 $*"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o pmc4 -- $BENCH2 > $OUT/pmc4.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o pmc5 -- $BENCH2 > $OUT/pmc5.log 2>&1
# the L2 request path of the step's kernels (two TCP counters per pass: more aborts rocprofv3), and the same counters on
# tools/pmc_calib.hip's random 8-B look-ups in an L2-resident table (cal_rgather8): the request rate the chip sustains
timeout 900 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $OUT -o pmc6 -- $BENCH > $OUT/pmc6.log 2>&1
if [ -x $GRAFT_REPO_ROOT/tools/bin/pmc_calib ]; then
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o cal1 -- $GRAFT_REPO_ROOT/tools/bin/pmc_calib 2 > $OUT/cal1.log 2>&1
fi
# condense on the box (the raw traces of a default run -- timed steps + the streaming leg -- exceed what gpurun brings back):
# $OUT/summary/<name>_{summary.md,kernel_stats.csv,traffic.json,bench_under_rocprof.json}; the big CSVs are dropped
NAME=$(basename $OUT)
mkdir -p $OUT/summary
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py $OUT $OUT/summary/$NAME > /dev/null 2> $OUT/summary/summarize.err
python tools/make_traffic.py $OUT $OUT/summary/$NAME > /dev/null 2> $OUT/summary/traffic.err
cp $OUT/summary/traffic_latest.json $OUT/summary/${NAME}_traffic_latest.json 2>/dev/null
find $OUT -maxdepth 1 -name "*.csv" -size +2M -delete
du -sh $OUT
