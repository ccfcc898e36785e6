#!/bin/bash
# Profile of the full-alignment path (config 3 shape, 1 M reads by default): kernel stats + PMC passes of the pass-2 kernel.
# usage (on the GPU box): tools/profile_align.sh gpurun_out/prof_align_r02 [n_reads]
#   then: python tools/summarize_align_prof.py gpurun_out/prof_align_r02 <tag>
set -e
export TMPDIR=/tmp
OUT=$1
N=${2:-1000000}
ARGS="tools/bench_align.py $N 1"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -- python3 $ARGS > $OUT/bench_a.txt 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_b -- python3 $ARGS > $OUT/bench_b.txt 2> $OUT/b.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.txt 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.txt 2> $OUT/write.err
