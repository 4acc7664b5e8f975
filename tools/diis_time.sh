cd $GRAFT_REPO_ROOT/gpurun_out
for v in 1 0; do
HELFEM_DIIS_LOWRANK=$v ../helfem_amd/bin/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 20 --mmax 1 --nelem 5 --nnodes 15 --method gga_x_pbe-gga_c_pbe --save "" > n2_diis_$v.log 2>&1
echo "LOWRANK=$v: $(grep -c Iteration n2_diis_$v.log) iterations; $(grep 'update done' n2_diis_$v.log | tail -2 | tr '\n' ' '); $(grep 'Total                 energy' n2_diis_$v.log)"
done
