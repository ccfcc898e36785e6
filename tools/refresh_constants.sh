#!/bin/bash
# Every rocprofv3 pass behind profiles/constants.json (the per-read instruction and HBM counts bench.py divides by measured times),
# against the library as built now. Kernel stats and PMC passes are separate runs, as MI355X_MICROARCH.md prescribes.
# usage (on the GPU box): tools/refresh_constants.sh gpurun_out/r04_const
#   then, with the directory merged back: python tools/refresh_constants.py gpurun_out/r04_const r04
set -e
export TMPDIR=/tmp
export PYTHONPATH=$PWD
OUT=$1
mkdir -p $OUT
python3 -c "from zoe_amd.build import fatbin_sha256; print(fatbin_sha256())" > $OUT/fatbin_sha256.txt
echo "band" && tools/profile_seed.sh $OUT/band 10000000
for e in ranges 3pass mixed align; do
    echo "entry $e"
    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_$e -- python3 tools/pmc_entry.py $e 1000000 > $OUT/pmc_$e.txt 2> $OUT/pmc_$e.err
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${e}_fetch -- python3 tools/pmc_entry.py $e 1000000 > $OUT/pmc_${e}_fetch.txt 2> $OUT/pmc_${e}_fetch.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${e}_write -- python3 tools/pmc_entry.py $e 1000000 > $OUT/pmc_${e}_write.txt 2> $OUT/pmc_${e}_write.err
done
echo "protein" && tools/profile_protein.sh $OUT/protein 1000000 0.03
echo "done"
