#!/bin/bash
# Profile of the column-pruned score pass on a 25-letter alphabet (1 M protein reads x 150 vs 2,000 residues, 3 % substituted):
# kernel stats + PMC passes of prune_strip_kernel<24,WIDE>, prune_window_kernel<24,4,32,0,WIDE> and score_kernel_v2<..,WIDE>.
# usage (on the GPU box): tools/profile_protein.sh gpurun_out/prof_protein_r03 [n_reads] [subs]
#   then: python tools/summarize_prune_prof.py gpurun_out/prof_protein_r03 <tag> [n_reads] protein
set -e
export TMPDIR=/tmp
export PYTHONPATH=$PWD
OUT=$1
N=${2:-1000000}
SUBS=${3:-0.03}
ARGS="tools/bench_protein.py $N $SUBS"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -- python3 $ARGS > $OUT/bench_a.txt 2> $OUT/a.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.txt 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.txt 2> $OUT/write.err
