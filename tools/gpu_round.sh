#!/bin/bash
# One GPU-box session: bench line, rocprofv3 kernel stats of the same command, PMC passes for the HBM traffic.
# Run through gpurun from the repo root; everything is written under gpurun_out/<tag>/.
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo write done
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json k_trd k_coulomb_tei k_backtransform k_dgemm > $OUT/pmc_traffic.txt
cat $OUT/pmc_traffic.txt
# keep only the summaries (the per-dispatch CSVs are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT $OUT/stats
