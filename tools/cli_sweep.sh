#!/bin/bash
# A sweep of small calculations through the two executables: convergence and plausible energies for a range of systems,
# methods and options (run on a GPU box from the repo root; prints one line per case).
B=helfem_amd/bin
run() { name=$1; shift; out=$("$@" --save "" 2>&1); rc=$?; e=$(echo "$out" | grep -E "^Total +energy:" | awk '{print $3}'); it=$(echo "$out" | grep -c "Iteration"); conv=$(echo "$out" | grep -c "Converged after"); echo "$name rc=$rc E=$e iterations=$it converged=$conv"; if [ $rc -ne 0 ]; then echo "$out" | tail -3; fi; }
run CO_PBE      $B/diatomic --Z1 C --Z2 O --Rbond 2.132 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe
run HF_HF       $B/diatomic --Z1 H --Z2 F --Rbond 1.7328 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method HF
run LiH_LDA     $B/diatomic --Z1 Li --Z2 H --Rbond 3.015 --lmax 8 --mmax 1 --nelem 4 --nnodes 10 --method lda_x-lda_c_vwn
run Be2_PBE0    $B/diatomic --Z1 Be --Z2 Be --Rbond 4.65 --lmax 8 --mmax 1 --nelem 4 --nnodes 10 --method hyb_gga_xc_pbeh
run O2_UPBE     $B/diatomic --Z1 O --Z2 O --Rbond 2.282 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --M 3
run OH_UHF      $B/diatomic --Z1 O --Z2 H --Rbond 1.8324 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method HF --M 2
run N2p_ROHF    $B/diatomic --Z1 N --Z2 N --Rbond 2.11 --Q 1 --lmax 8 --mmax 2 --nelem 3 --nnodes 10 --method HF --M 2 --restricted 1
run H2p         $B/diatomic --Z1 H --Z2 H --Rbond 2.0 --Q 1 --lmax 8 --mmax 0 --nelem 4 --nnodes 10 --method HF --M 2
run HeH_TPSS    $B/diatomic --Z1 He --Z2 H --Q 1 --Rbond 1.46 --lmax 6 --mmax 1 --nelem 3 --nnodes 10 --method mgga_x_tpss-mgga_c_tpss
run N2_sym2     $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --symmetry 2
run N2_TF       $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --iguess 3
run N2_chol     $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 10 --mmax 2 --nelem 4 --nnodes 10 --method gga_x_pbe-gga_c_pbe --diag 0
run Ne_PBE      $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 5 --nnodes 15 --method gga_x_pbe-gga_c_pbe
run Ar_HF       $B/atomic --Z Ar --lmax 1 --mmax 1 --nelem 10 --nnodes 15 --method HF
run N_UPBE      $B/atomic --Z N --lmax 1 --mmax 1 --nelem 5 --nnodes 15 --method gga_x_pbe-gga_c_pbe --M 4
run C_UHF_mavg  $B/atomic --Z C --lmax 2 --mmax 2 --nelem 5 --nnodes 15 --method HF --M 3 --maverage 1
run Na_ROHF     $B/atomic --Z Na --lmax 1 --mmax 1 --nelem 8 --nnodes 15 --method HF --M 2 --restricted 1
run Ne_CAMLDA0  $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 5 --nnodes 15 --method hyb_lda_xc_cam_lda0
run Zn_LDA      $B/atomic --Z Zn --lmax 2 --mmax 2 --nelem 10 --nnodes 15 --method lda_x-lda_c_pw
run Kr_PBE_s2   $B/atomic --Z Kr --lmax 2 --mmax 2 --nelem 10 --nnodes 15 --method gga_x_pbe-gga_c_pbe --symmetry 2
run Rn_LDA_s2   $B/atomic --Z Rn --lmax 3 --mmax 3 --nelem 10 --nnodes 15 --method lda_x-lda_c_vwn --symmetry 2
run Xe_HF_s2    $B/atomic --Z Xe --lmax 2 --mmax 2 --nelem 10 --nnodes 15 --method HF --symmetry 2
run Ne_PBE_l6   $B/atomic --Z Ne --lmax 6 --mmax 6 --nelem 4 --nnodes 12 --method gga_x_pbe-gga_c_pbe
run Cu_UPBE     $B/atomic --Z Cu --lmax 2 --mmax 2 --nelem 10 --nnodes 15 --method gga_x_pbe-gga_c_pbe --M 2
run Ne_PBE_n25  $B/atomic --Z Ne --lmax 1 --mmax 1 --nelem 3 --nnodes 25 --method gga_x_pbe-gga_c_pbe
run CO_PBE_l36  $B/diatomic --Z1 C --Z2 O --Rbond 2.132 --lmax 36 --mmax 2 --nelem 3 --nnodes 10 --method gga_x_pbe-gga_c_pbe
run O2_UPBE_l30 $B/diatomic --Z1 O --Z2 O --Rbond 2.282 --lmax 30 --mmax 2 --nelem 3 --nnodes 10 --method gga_x_pbe-gga_c_pbe --M 3
run N2_PBE_e30  $B/diatomic --Z1 N --Z2 N --Rbond 2.068 --lmax 6 --mmax 1 --nelem 30 --nnodes 15 --method gga_x_pbe-gga_c_pbe
