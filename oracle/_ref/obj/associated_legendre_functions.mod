﻿!mod$ v1 sum:03b3b16b9e776591
!need$ 89c30a47e5895f6f n auxilliary_subroutines
!need$ 6270394355f35b59 n prolate_functions
!need$ 13a61343b5675ca9 n lentz_thompson
!need$ 0e9501db05b6b31a n matrix_print
module associated_legendre_functions
use auxilliary_subroutines,only:isp
use auxilliary_subroutines,only:selected_real_kind
use auxilliary_subroutines,only:int_sp
use auxilliary_subroutines,only:selected_int_kind
use auxilliary_subroutines,only:int_dp
use auxilliary_subroutines,only:idp
use auxilliary_subroutines,only:iqp
use auxilliary_subroutines,only:inp
use auxilliary_subroutines,only:iout
use auxilliary_subroutines,only:rows_to_print
use auxilliary_subroutines,only:columns_to_print
use auxilliary_subroutines,only:eigenvectors_to_print
use auxilliary_subroutines,only:print_parameter
use auxilliary_subroutines,only:rowlab
use auxilliary_subroutines,only:collab
use auxilliary_subroutines,only:pi
use auxilliary_subroutines,only:two_pi
use auxilliary_subroutines,only:zero
use auxilliary_subroutines,only:quarter
use auxilliary_subroutines,only:half
use auxilliary_subroutines,only:third
use auxilliary_subroutines,only:fourth
use auxilliary_subroutines,only:fifth
use auxilliary_subroutines,only:sixth
use auxilliary_subroutines,only:seventh
use auxilliary_subroutines,only:eighth
use auxilliary_subroutines,only:ninth
use auxilliary_subroutines,only:tenth
use auxilliary_subroutines,only:one
use auxilliary_subroutines,only:two
use auxilliary_subroutines,only:three
use auxilliary_subroutines,only:four
use auxilliary_subroutines,only:five
use auxilliary_subroutines,only:six
use auxilliary_subroutines,only:seven
use auxilliary_subroutines,only:eight
use auxilliary_subroutines,only:nine
use auxilliary_subroutines,only:ten
use auxilliary_subroutines,only:nrzero
use auxilliary_subroutines,only:sqrt2
use auxilliary_subroutines,only:sqrt
use auxilliary_subroutines,only:a_fac
use auxilliary_subroutines,only:b_fac
use auxilliary_subroutines,only:int_zero
use auxilliary_subroutines,only:int_one
use auxilliary_subroutines,only:int_two
use auxilliary_subroutines,only:int_three
use auxilliary_subroutines,only:int_four
use auxilliary_subroutines,only:int_five
use auxilliary_subroutines,only:int_six
use auxilliary_subroutines,only:int_seven
use auxilliary_subroutines,only:int_eight
use auxilliary_subroutines,only:int_nine
use auxilliary_subroutines,only:int_ten
use auxilliary_subroutines,only:int_eleven
use auxilliary_subroutines,only:int_twelve
use auxilliary_subroutines,only:int_thirteen
use auxilliary_subroutines,only:int_fourteen
use auxilliary_subroutines,only:int_fifteen
use auxilliary_subroutines,only:int_sixteen
use auxilliary_subroutines,only:int_seventeen
use auxilliary_subroutines,only:int_eighteen
use auxilliary_subroutines,only:int_nineteen
use auxilliary_subroutines,only:int_twenty
use auxilliary_subroutines,only:int_max
use auxilliary_subroutines,only:hbar
use auxilliary_subroutines,only:massau
use auxilliary_subroutines,only:lenau
use auxilliary_subroutines,only:timau
use auxilliary_subroutines,only:efieldau
use auxilliary_subroutines,only:electric_field_to_intensity
use auxilliary_subroutines,only:peak_electric_field
use auxilliary_subroutines,only:pmass
use auxilliary_subroutines,only:massn2p
use auxilliary_subroutines,only:au_in_ev
use auxilliary_subroutines,only:x
use auxilliary_subroutines,only:y
use auxilliary_subroutines,only:m_max
use auxilliary_subroutines,only:m_min
use auxilliary_subroutines,only:l_max
use auxilliary_subroutines,only:n_points
use auxilliary_subroutines,only:normalized
use auxilliary_subroutines,only:derivative
use auxilliary_subroutines,only:print_functions
use auxilliary_subroutines,only:print_wronskian
use auxilliary_subroutines,only:print_norms
use auxilliary_subroutines,only:print_factors
use auxilliary_subroutines,only:input_values
use auxilliary_subroutines,only:test_wron
use auxilliary_subroutines,only:norm
use auxilliary_subroutines,only:arg
use auxilliary_subroutines,only:scale_factor
use auxilliary_subroutines,only:log_factor
use auxilliary_subroutines,only:wron
use auxilliary_subroutines,only:factor
use auxilliary_subroutines,only:l
use auxilliary_subroutines,only:m
use auxilliary_subroutines,only:m_sign
use auxilliary_subroutines,only:s_fac
use auxilliary_subroutines,only:smallest
use auxilliary_subroutines,only:tiny
use auxilliary_subroutines,only:biggest
use auxilliary_subroutines,only:huge
use auxilliary_subroutines,only:eps
use auxilliary_subroutines,only:upper
use auxilliary_subroutines,only:lower
use auxilliary_subroutines,only:step
use auxilliary_subroutines,only:row_label
use auxilliary_subroutines,only:col_label
use auxilliary_subroutines,only:title
use auxilliary_subroutines,only:control
use auxilliary_subroutines,only:recur
use auxilliary_subroutines,only:directive
use auxilliary_subroutines,only:xi
use auxilliary_subroutines,only:eta
use auxilliary_subroutines,only:reg_l
use auxilliary_subroutines,only:reg_m
use auxilliary_subroutines,only:reg_lm
use auxilliary_subroutines,only:irreg_l
use auxilliary_subroutines,only:irreg_m
use auxilliary_subroutines,only:irreg_lm
use auxilliary_subroutines,only:up
use auxilliary_subroutines,only:down_a
use auxilliary_subroutines,only:down_b
use auxilliary_subroutines,only:down
use auxilliary_subroutines,only:cf_legendre
use auxilliary_subroutines,only:coefficients
use auxilliary_subroutines,only:legendre_functions
use auxilliary_subroutines,only:normalization
use auxilliary_subroutines,only:leg
use auxilliary_subroutines,only:factorials
use auxilliary_subroutines,only:wronskian
use auxilliary_subroutines,only:normalization_factors
use auxilliary_subroutines,only:print_norm_factors
use auxilliary_subroutines,only:renormalize
use prolate_functions,only:lorder
use prolate_functions,only:morder
use prolate_functions,only:mabs
use prolate_functions,only:meo
use prolate_functions,only:a
use prolate_functions,only:r_int
use prolate_functions,only:radius_moeq
use prolate_functions,only:point
use prolate_functions,only:a_p
use prolate_functions,only:x_i
use prolate_functions,only:eta_i
use prolate_functions,only:rho_i
use prolate_functions,only:varphi
use prolate_functions,only:r
use prolate_functions,only:dr
use prolate_functions,only:xi_small
use prolate_functions,only:xi_large
use prolate_functions,only:facm
use prolate_functions,only:vardm
use prolate_functions,only:dl21
use prolate_functions,only:temp
use prolate_functions,only:csum_real
use prolate_functions,only:csum_imag
use prolate_functions,only:varphi_diff
use prolate_functions,only:ctemp_real
use prolate_functions,only:ctemp_imag
use prolate_functions,only:rsqr
use prolate_functions,only:r_12
use prolate_functions,only:r_12_invs
use lentz_thompson,only:print_matrix
use lentz_thompson,only:print_matrix_d
use lentz_thompson,only:print_matrix_z
use lentz_thompson,only:print_triangle_matrix_d
use lentz_thompson,only:print_triangle_matrix_z
use lentz_thompson,only:print_vector_d
use lentz_thompson,only:print_vector_z
use lentz_thompson,only:continued_fractions
use lentz_thompson,only:continued_fraction_legendre
use matrix_print,only:lentz_thompson$matrix_print$print_matrix_d=>print_matrix_d
interface legendre
procedure::legendre
end interface
interface legendre_recursion
procedure::upward_regular_legendre_recursion_l
procedure::upward_regular_legendre_recursion_lm
procedure::upward_irregular_legendre_recursion_lm
procedure::downward_irregular_legendre_recursion_lm_a
procedure::downward_irregular_legendre_recursion_lm_b
end interface
interface initialize
procedure::initialize_regular_l
procedure::initialize_regular_lm
procedure::initialize_irregular_l
procedure::initialize_irregular_lm
end interface
contains
subroutine legendre(r_lm,i_lm,normalized)
type(reg_lm),optional::r_lm
type(irreg_lm),optional::i_lm
logical(4),optional::normalized
end
subroutine initialize_regular_l(r_l,normalized)
type(reg_l)::r_l
logical(4),optional::normalized
end
subroutine initialize_regular_lm(r_lm,normalized)
type(reg_lm)::r_lm
logical(4),optional::normalized
end
subroutine initialize_irregular_l(i_l)
type(irreg_l)::i_l
end
subroutine initialize_irregular_lm(i_lm)
type(irreg_lm)::i_lm
end
subroutine upward_regular_legendre_recursion_l(r_l,normalized)
type(reg_l)::r_l
logical(4),optional::normalized
end
subroutine upward_regular_legendre_recursion_lm(r_lm,normalized)
type(reg_lm)::r_lm
logical(4),optional::normalized
end
subroutine upward_irregular_legendre_recursion_lm(i_lm,u)
type(irreg_lm)::i_lm
type(up)::u
end
subroutine downward_irregular_legendre_recursion_lm_a(i_lm,d,a)
type(irreg_lm)::i_lm
type(down)::d
type(down_a)::a
end
subroutine downward_irregular_legendre_recursion_lm_b(i_lm,d,b)
type(irreg_lm)::i_lm
type(down)::d
type(down_b)::b
end
end
