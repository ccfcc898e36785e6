#!/bin/bash
# Round-end profile of the headline kernel: kernel stats + three separate PMC passes (HBM read, HBM write, SQ counters).
# usage (on the GPU box): tools/profile_headline.sh gpurun_out/prof_r01c ; then python tools/summarize_prof.py gpurun_out/prof_r01c <tag>
set -e
export TMPDIR=/tmp
OUT=$1
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify 0 --no-secondary --no-pruned"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
