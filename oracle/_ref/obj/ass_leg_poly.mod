﻿!mod$ v1 sum:47f5c4cda8f38357
!need$ 2c37ccdf5d34d40d n accuracy
!need$ 49f150a7136fb138 n input_output
!need$ 0e9501db05b6b31a n matrix_print
module ass_leg_poly
use accuracy,only:isp
use accuracy,only:selected_real_kind
use accuracy,only:int_sp
use accuracy,only:selected_int_kind
use accuracy,only:int_dp
use accuracy,only:idp
use accuracy,only:iqp
use input_output,only:inp
use input_output,only:iout
use input_output,only:rows_to_print
use input_output,only:columns_to_print
use input_output,only:eigenvectors_to_print
use input_output,only:print_parameter
use input_output,only:rowlab
use input_output,only:collab
use matrix_print,only:print_matrix
use matrix_print,only:print_matrix_d
use matrix_print,only:print_matrix_z
use matrix_print,only:print_triangle_matrix_d
use matrix_print,only:print_triangle_matrix_z
use matrix_print,only:print_vector_d
use matrix_print,only:print_vector_z
contains
subroutine p_lm(plm,x,m,l_max)
integer(4)::m
integer(4)::l_max
real(8)::plm(int(m,kind=8):int(l_max,kind=8))
real(8)::x
end
end
