set -e
R=$GRAFT_REPO_ROOT
python3 $R/tools/exchange_bench.py n2_pbe_nbf4230 groups > $R/gpurun_out/exchange_bench.txt 2>&1
python3 $R/tools/exchange_bench.py lif_pbe_nbf6102 > $R/gpurun_out/exchange_bench_lif.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exstats -o ex -- python3 $R/tools/exchange_bench.py n2_pbe_nbf4230 > /dev/null 2>&1
cp $(find $R/gpurun_out/exstats -name "*kernel_stats.csv") $R/gpurun_out/exchange_kernel_stats.csv
rm -rf $R/gpurun_out/exstats
cat $R/gpurun_out/exchange_bench.txt $R/gpurun_out/exchange_bench_lif.txt
