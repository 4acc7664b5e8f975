B=helfem_amd/bin
run() { name=$1; shift; out=$("$@" --save "" 2>&1); rc=$?; e=$(echo "$out" | grep -E "^Total +energy:" | awk '{print $3}'); echo "$name rc=$rc E=$e $(echo "$out" | grep -E "Converged after|NOT converged" | head -1)"; if [ $rc -ne 0 ]; then echo "$out" | grep -v amdgpu | tail -2; fi; }
run H_atom_min   $B/atomic --Z H --lmax 0 --mmax 0 --nelem 1 --nnodes 3 --method HF --M 2
run H_atom       $B/atomic --Z H --lmax 0 --mmax 0 --nelem 5 --nnodes 15 --method HF --M 2
run Hep_LDA      $B/atomic --Z He --Q 1 --lmax 0 --mmax 0 --nelem 5 --nnodes 15 --method lda_x-lda_c_vwn --M 2
run He_sym0      $B/atomic --Z He --lmax 1 --mmax 1 --nelem 3 --nnodes 8 --method HF --symmetry 0
run Li_sym0_U    $B/atomic --Z Li --lmax 1 --mmax 1 --nelem 3 --nnodes 8 --method HF --symmetry 0 --M 2
run H2_sym0      $B/diatomic --Z1 H --Z2 H --Rbond 1.4 --lmax 4 --mmax 1 --nelem 2 --nnodes 8 --method HF --symmetry 0
run H2_long      $B/diatomic --Z1 H --Z2 H --Rbond 12.0 --lmax 12 --mmax 0 --nelem 4 --nnodes 10 --method HF
run H2_short     $B/diatomic --Z1 H --Z2 H --Rbond 0.2 --lmax 4 --mmax 0 --nelem 3 --nnodes 10 --method HF
run He_ghost     $B/diatomic --Z1 He --Z2 0 --Rbond 2.0 --lmax 6 --mmax 0 --nelem 3 --nnodes 10 --method HF
run HeH_nelb     $B/diatomic --Z1 He --Z2 H --Rbond 1.46 --lmax 4 --mmax 1 --nelem 2 --nnodes 8 --method HF --nela 2 --nelb 1
run N2_maxit2    $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 4 --mmax 1 --nelem 2 --nnodes 8 --method HF --maxit 2
run N2_lmaxlist  $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 6,4,2 --nelem 2 --nnodes 8 --method HF
run N2_grid1     $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 4 --mmax 1 --nelem 4 --nnodes 8 --method HF --grid 1
run N2_nquad     $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 4 --mmax 1 --nelem 2 --nnodes 8 --method gga_x_pbe-gga_c_pbe --nquad 60 --ldft 40 --mdft 15
run N2_angstrom  $B/diatomic --Z1 N --Z2 N --Rbond 1.0943 --angstrom 1 --lmax 4 --mmax 1 --nelem 2 --nnodes 8 --method HF
run Ne_dftthr0   $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --dftthr 0
run Ne_diis      $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --diisorder 10 --diiseps 1e-1 --diisthr 1e-2
run Ne_nodamp    $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 4 --nnodes 10 --method HF --dampfock 1.0
