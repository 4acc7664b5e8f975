﻿!mod$ v1 sum:9bc1909bfe4d016d
!need$ 03b3b16b9e776591 n associated_legendre_functions
!need$ 0bde2ac47243ead2 i iso_c_binding
module legendre_wrapper_m
use associated_legendre_functions,only:isp
use associated_legendre_functions,only:selected_real_kind
use associated_legendre_functions,only:int_sp
use associated_legendre_functions,only:selected_int_kind
use associated_legendre_functions,only:int_dp
use associated_legendre_functions,only:idp
use associated_legendre_functions,only:iqp
use associated_legendre_functions,only:inp
use associated_legendre_functions,only:iout
use associated_legendre_functions,only:rows_to_print
use associated_legendre_functions,only:columns_to_print
use associated_legendre_functions,only:eigenvectors_to_print
use associated_legendre_functions,only:print_parameter
use associated_legendre_functions,only:rowlab
use associated_legendre_functions,only:collab
use associated_legendre_functions,only:pi
use associated_legendre_functions,only:two_pi
use associated_legendre_functions,only:zero
use associated_legendre_functions,only:quarter
use associated_legendre_functions,only:half
use associated_legendre_functions,only:third
use associated_legendre_functions,only:fourth
use associated_legendre_functions,only:fifth
use associated_legendre_functions,only:sixth
use associated_legendre_functions,only:seventh
use associated_legendre_functions,only:eighth
use associated_legendre_functions,only:ninth
use associated_legendre_functions,only:tenth
use associated_legendre_functions,only:one
use associated_legendre_functions,only:two
use associated_legendre_functions,only:three
use associated_legendre_functions,only:four
use associated_legendre_functions,only:five
use associated_legendre_functions,only:six
use associated_legendre_functions,only:seven
use associated_legendre_functions,only:eight
use associated_legendre_functions,only:nine
use associated_legendre_functions,only:ten
use associated_legendre_functions,only:nrzero
use associated_legendre_functions,only:sqrt2
use associated_legendre_functions,only:sqrt
use associated_legendre_functions,only:a_fac
use associated_legendre_functions,only:b_fac
use associated_legendre_functions,only:int_zero
use associated_legendre_functions,only:int_one
use associated_legendre_functions,only:int_two
use associated_legendre_functions,only:int_three
use associated_legendre_functions,only:int_four
use associated_legendre_functions,only:int_five
use associated_legendre_functions,only:int_six
use associated_legendre_functions,only:int_seven
use associated_legendre_functions,only:int_eight
use associated_legendre_functions,only:int_nine
use associated_legendre_functions,only:int_ten
use associated_legendre_functions,only:int_eleven
use associated_legendre_functions,only:int_twelve
use associated_legendre_functions,only:int_thirteen
use associated_legendre_functions,only:int_fourteen
use associated_legendre_functions,only:int_fifteen
use associated_legendre_functions,only:int_sixteen
use associated_legendre_functions,only:int_seventeen
use associated_legendre_functions,only:int_eighteen
use associated_legendre_functions,only:int_nineteen
use associated_legendre_functions,only:int_twenty
use associated_legendre_functions,only:int_max
use associated_legendre_functions,only:hbar
use associated_legendre_functions,only:massau
use associated_legendre_functions,only:lenau
use associated_legendre_functions,only:timau
use associated_legendre_functions,only:efieldau
use associated_legendre_functions,only:electric_field_to_intensity
use associated_legendre_functions,only:peak_electric_field
use associated_legendre_functions,only:pmass
use associated_legendre_functions,only:massn2p
use associated_legendre_functions,only:au_in_ev
use associated_legendre_functions,only:x
use associated_legendre_functions,only:y
use associated_legendre_functions,only:m_max
use associated_legendre_functions,only:m_min
use associated_legendre_functions,only:l_max
use associated_legendre_functions,only:n_points
use associated_legendre_functions,only:normalized
use associated_legendre_functions,only:derivative
use associated_legendre_functions,only:print_functions
use associated_legendre_functions,only:print_wronskian
use associated_legendre_functions,only:print_norms
use associated_legendre_functions,only:print_factors
use associated_legendre_functions,only:input_values
use associated_legendre_functions,only:test_wron
use associated_legendre_functions,only:norm
use associated_legendre_functions,only:arg
use associated_legendre_functions,only:scale_factor
use associated_legendre_functions,only:log_factor
use associated_legendre_functions,only:wron
use associated_legendre_functions,only:factor
use associated_legendre_functions,only:l
use associated_legendre_functions,only:m
use associated_legendre_functions,only:m_sign
use associated_legendre_functions,only:s_fac
use associated_legendre_functions,only:smallest
use associated_legendre_functions,only:tiny
use associated_legendre_functions,only:biggest
use associated_legendre_functions,only:huge
use associated_legendre_functions,only:eps
use associated_legendre_functions,only:upper
use associated_legendre_functions,only:lower
use associated_legendre_functions,only:step
use associated_legendre_functions,only:row_label
use associated_legendre_functions,only:col_label
use associated_legendre_functions,only:title
use associated_legendre_functions,only:control
use associated_legendre_functions,only:recur
use associated_legendre_functions,only:directive
use associated_legendre_functions,only:xi
use associated_legendre_functions,only:eta
use associated_legendre_functions,only:reg_l
use associated_legendre_functions,only:reg_m
use associated_legendre_functions,only:reg_lm
use associated_legendre_functions,only:irreg_l
use associated_legendre_functions,only:irreg_m
use associated_legendre_functions,only:irreg_lm
use associated_legendre_functions,only:up
use associated_legendre_functions,only:down_a
use associated_legendre_functions,only:down_b
use associated_legendre_functions,only:down
use associated_legendre_functions,only:cf_legendre
use associated_legendre_functions,only:coefficients
use associated_legendre_functions,only:legendre_functions
use associated_legendre_functions,only:normalization
use associated_legendre_functions,only:leg
use associated_legendre_functions,only:factorials
use associated_legendre_functions,only:wronskian
use associated_legendre_functions,only:normalization_factors
use associated_legendre_functions,only:print_norm_factors
use associated_legendre_functions,only:renormalize
use associated_legendre_functions,only:lorder
use associated_legendre_functions,only:morder
use associated_legendre_functions,only:mabs
use associated_legendre_functions,only:meo
use associated_legendre_functions,only:a
use associated_legendre_functions,only:r_int
use associated_legendre_functions,only:radius_moeq
use associated_legendre_functions,only:point
use associated_legendre_functions,only:a_p
use associated_legendre_functions,only:x_i
use associated_legendre_functions,only:eta_i
use associated_legendre_functions,only:rho_i
use associated_legendre_functions,only:varphi
use associated_legendre_functions,only:r
use associated_legendre_functions,only:dr
use associated_legendre_functions,only:xi_small
use associated_legendre_functions,only:xi_large
use associated_legendre_functions,only:facm
use associated_legendre_functions,only:vardm
use associated_legendre_functions,only:dl21
use associated_legendre_functions,only:temp
use associated_legendre_functions,only:csum_real
use associated_legendre_functions,only:csum_imag
use associated_legendre_functions,only:varphi_diff
use associated_legendre_functions,only:ctemp_real
use associated_legendre_functions,only:ctemp_imag
use associated_legendre_functions,only:rsqr
use associated_legendre_functions,only:r_12
use associated_legendre_functions,only:r_12_invs
use associated_legendre_functions,only:print_matrix
use associated_legendre_functions,only:print_matrix_d
use associated_legendre_functions,only:print_matrix_z
use associated_legendre_functions,only:print_triangle_matrix_d
use associated_legendre_functions,only:print_triangle_matrix_z
use associated_legendre_functions,only:print_vector_d
use associated_legendre_functions,only:print_vector_z
use associated_legendre_functions,only:continued_fractions
use associated_legendre_functions,only:continued_fraction_legendre
use associated_legendre_functions,only:lentz_thompson$matrix_print$print_matrix_d
use associated_legendre_functions,only:legendre
use associated_legendre_functions,only:legendre_recursion
use associated_legendre_functions,only:initialize
use associated_legendre_functions,only:initialize_regular_l
use associated_legendre_functions,only:initialize_regular_lm
use associated_legendre_functions,only:initialize_irregular_l
use associated_legendre_functions,only:initialize_irregular_lm
use associated_legendre_functions,only:upward_regular_legendre_recursion_l
use associated_legendre_functions,only:upward_regular_legendre_recursion_lm
use associated_legendre_functions,only:upward_irregular_legendre_recursion_lm
use associated_legendre_functions,only:downward_irregular_legendre_recursion_lm_a
use associated_legendre_functions,only:downward_irregular_legendre_recursion_lm_b
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
contains
subroutine calculate_plm_array(plm,xi)
real(8),intent(out)::plm(:,:)
real(8),intent(in)::xi
end
subroutine calculate_normalized_plm_array(plm,xi)
real(8),intent(out)::plm(:,:)
real(8),intent(in)::xi
end
subroutine calculate_qlm_array(qlm,xi)
real(8),intent(out)::qlm(:,:)
real(8),intent(in)::xi
end
function calculate_plm(l,m,xi) result(r)
integer(4),intent(in)::l
integer(4),intent(in)::m
real(8),intent(in)::xi
real(8)::r
end
function calculate_normalized_plm(l,m,xi) result(r)
integer(4),intent(in)::l
integer(4),intent(in)::m
real(8),intent(in)::xi
real(8)::r
end
function calculate_qlm(l,m,xi) result(i)
integer(4),intent(in)::l
integer(4),intent(in)::m
real(8),intent(in)::xi
real(8)::i
end
subroutine calc_plm_arr(r,lmax,mmax,xi) bind(c,name="calc_Plm_arr")
integer(4),value::lmax
integer(4),value::mmax
real(8)::r(1_8:int((lmax+1_4)*(mmax+1_4),kind=8))
real(8),value::xi
end
subroutine calc_norm_plm_arr(r,lmax,mmax,xi) bind(c,name="calc_norm_Plm_arr")
integer(4),value::lmax
integer(4),value::mmax
real(8)::r(1_8:int((lmax+1_4)*(mmax+1_4),kind=8))
real(8),value::xi
end
subroutine calc_qlm_arr(i,lmax,mmax,xi) bind(c,name="calc_Qlm_arr")
integer(4),value::lmax
integer(4),value::mmax
real(8)::i(1_8:int((lmax+1_4)*(mmax+1_4),kind=8))
real(8),value::xi
end
function calc_plm_val(l,m,xi) bind(c,name="calc_Plm_val") result(r)
integer(4),value::l
integer(4),value::m
real(8),value::xi
real(8)::r
end
function calc_norm_plm_val(l,m,xi) bind(c,name="calc_norm_Plm_val") result(r)
integer(4),value::l
integer(4),value::m
real(8),value::xi
real(8)::r
end
function calc_qlm_val(l,m,xi) bind(c,name="calc_Qlm_val") result(i)
integer(4),value::l
integer(4),value::m
real(8),value::xi
real(8)::i
end
end
