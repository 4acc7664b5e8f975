#!/bin/bash
# One GPU-box session: bench lines, rocprofv3 kernel stats of the same commands, PMC passes for the HBM traffic.
# Run through gpurun from the repo root:   bash tools/gpu_round.sh r03 [hybrid]
# Everything is written under gpurun_out/<tag>/; the summaries to keep are copied by hand into profiles/.
#   default workload (n2_pbe_nbf4230):   bench.json, stats/, pmc_traffic.json
#   hybrid (lif_pbe0_nbf6102), optional: bench_lif_pbe0.json, stats_lif_pbe0/, pmc_traffic_lif_pbe0.json
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
PY=python3   # (under rocprofv3 the program itself must follow "--": no env / bash -c hops)

one_workload () {   # $1 = workload, $2 = suffix of the output names
  W=$1; S=$2
  cd $REPO
  $PY bench.py --workload $W --steps 5 --warmup 1 > $OUT/bench$S.json 2> $OUT/bench$S.err
  cat $OUT/bench$S.json
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$S -o stats -- $PY $REPO/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats$S.log 2>&1 || echo "(rocprofv3 exit status $? -- its reports are written before the profiled process exits)"
  echo "stats $W done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch$S -o fetch -- $PY $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch$S.log 2>&1 || echo "(rocprofv3 exit status $?)"
  echo "fetch $W done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write$S -o write -- $PY $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write$S.log 2>&1 || echo "(rocprofv3 exit status $?)"
  echo "write $W done"
  cd $REPO
  $PY tools/pmc_traffic.py $OUT/pmc_fetch$S $OUT/pmc_write$S $OUT/pmc_traffic$S.json k_trd k_coulomb_tei k_backtransform k_dgemm k_exl > $OUT/pmc_traffic$S.txt
  cat $OUT/pmc_traffic$S.txt
}

one_workload n2_pbe_nbf4230 ""
if [ "$2" = "hybrid" ]; then one_workload lif_pbe0_nbf6102 _lif_pbe0; fi
# keep only the summaries (the per-dispatch CSVs are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT $OUT/stats
