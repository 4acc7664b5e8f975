#!/bin/bash
# FP64 matrix-core ceiling and what the GEMM tile kernel waits for (on the GPU box):  bash tools/gemm_pmc.sh
#   gpurun_out/gemm/fp64_rate.txt   register-only issue rates (tests/gpu_probe/fp64_rate.hip)
#   gpurun_out/gemm/gemm_bench.txt  the tile kernel alone (tools/gemm_bench.py)
#   gpurun_out/gemm/pmc_*.txt       SQ counters of the k_dgemm* kernels in tools/gemm_bench.py and in one bench step
OUT=$PWD/gpurun_out/gemm
REPO=$PWD
mkdir -p $OUT
tests/gpu_probe/fp64_rate > $OUT/fp64_rate.txt 2>&1 || exit 1
cat $OUT/fp64_rate.txt
python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_bench.txt || exit 1
cat $OUT/gemm_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || echo "(rocprofv3 -L: $?)"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64"
P2="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU"
for w in gemm bench; do
  if [ $w = gemm ]; then CMD="python3 $REPO/tools/gemm_bench.py"; else CMD="python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"; fi
  rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1_$w -o p1 -- $CMD > $OUT/p1_$w.log 2>&1 || echo "(rocprofv3 pass 1 $w: exit $?)"
  rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2_$w -o p2 -- $CMD > $OUT/p2_$w.log 2>&1 || echo "(rocprofv3 pass 2 $w: exit $?)"
  cd $REPO
  python3 tools/pmc_sq.py $OUT/p1_$w $OUT/p2_$w -- k_dgemm > $OUT/pmc_$w.txt 2>&1
  cat $OUT/pmc_$w.txt
  cd /tmp
done
find $OUT -name "*counter_collection.csv" -size +20M -delete
