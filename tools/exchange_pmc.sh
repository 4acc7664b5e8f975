#!/bin/bash
# HBM-side traffic of the exchange kernels (separate FETCH_SIZE / WRITE_SIZE passes, reduced by tools/pmc_traffic.py)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/expmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/tools/exchange_bench.py n2_pbe_nbf4230 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $R/tools/exchange_bench.py n2_pbe_nbf4230 > $OUT/write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json k_exl k_dgemm_tasklist > $OUT/pmc_traffic.txt
cat $OUT/pmc_traffic.txt
find $OUT -name "*counter_collection.csv" -size +5M -delete
