#!/bin/bash
# Profile of the seeded exact score pass: kernel stats + PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).
# usage (on the GPU box): tools/profile_seed.sh gpurun_out/prof_seed_r03 [n_reads] [extra try_seed.py args]
#   then: python tools/summarize_seed_prof.py gpurun_out/prof_seed_r03 <tag> [n_reads]
set -e
export TMPDIR=/tmp
export PYTHONPATH=$PWD
OUT=$1
N=${2:-10000000}
shift; shift || true
ARGS="tools/try_seed.py $N $@"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_a -- python3 $ARGS > $OUT/bench_a.txt 2> $OUT/a.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.txt 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.txt 2> $OUT/write.err
