#!/bin/bash
# Profile of the column-pruned score pass (2 M reads x 150 bp vs 2 kb): kernel stats + PMC passes of its three kernels.
# usage (on the GPU box): tools/profile_prune.sh gpurun_out/prof_prune_r02 [n_reads]
#   then: python tools/summarize_prune_prof.py gpurun_out/prof_prune_r02 <tag> [n_reads]
set -e
export TMPDIR=/tmp
export PYTHONPATH=$PWD
OUT=$1
N=${2:-2000000}
ARGS="tools/try_prune.py $N --score-only"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -- python3 $ARGS > $OUT/bench_a.txt 2> $OUT/a.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.txt 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.txt 2> $OUT/write.err
