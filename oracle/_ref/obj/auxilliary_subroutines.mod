﻿!mod$ v1 sum:89c30a47e5895f6f
!need$ 99f7e6bc3f7597b4 n special_functions
!need$ 2c37ccdf5d34d40d n accuracy
!need$ 40fa78096c51d7cb n data_module
module auxilliary_subroutines
use accuracy,only:isp
use accuracy,only:selected_real_kind
use accuracy,only:int_sp
use accuracy,only:selected_int_kind
use accuracy,only:int_dp
use accuracy,only:idp
use accuracy,only:iqp
use data_module,only:inp
use data_module,only:iout
use data_module,only:rows_to_print
use data_module,only:columns_to_print
use data_module,only:eigenvectors_to_print
use data_module,only:print_parameter
use data_module,only:rowlab
use data_module,only:collab
use data_module,only:pi
use data_module,only:two_pi
use data_module,only:zero
use data_module,only:quarter
use data_module,only:half
use data_module,only:third
use data_module,only:fourth
use data_module,only:fifth
use data_module,only:sixth
use data_module,only:seventh
use data_module,only:eighth
use data_module,only:ninth
use data_module,only:tenth
use data_module,only:one
use data_module,only:two
use data_module,only:three
use data_module,only:four
use data_module,only:five
use data_module,only:six
use data_module,only:seven
use data_module,only:eight
use data_module,only:nine
use data_module,only:ten
use data_module,only:nrzero
use data_module,only:sqrt2
use data_module,only:sqrt
use data_module,only:a_fac
use data_module,only:b_fac
use data_module,only:int_zero
use data_module,only:int_one
use data_module,only:int_two
use data_module,only:int_three
use data_module,only:int_four
use data_module,only:int_five
use data_module,only:int_six
use data_module,only:int_seven
use data_module,only:int_eight
use data_module,only:int_nine
use data_module,only:int_ten
use data_module,only:int_eleven
use data_module,only:int_twelve
use data_module,only:int_thirteen
use data_module,only:int_fourteen
use data_module,only:int_fifteen
use data_module,only:int_sixteen
use data_module,only:int_seventeen
use data_module,only:int_eighteen
use data_module,only:int_nineteen
use data_module,only:int_twenty
use data_module,only:int_max
use data_module,only:hbar
use data_module,only:massau
use data_module,only:lenau
use data_module,only:timau
use data_module,only:efieldau
use data_module,only:electric_field_to_intensity
use data_module,only:peak_electric_field
use data_module,only:pmass
use data_module,only:massn2p
use data_module,only:au_in_ev
use special_functions,only:x
use special_functions,only:y
use special_functions,only:m_max
use special_functions,only:m_min
use special_functions,only:l_max
use special_functions,only:n_points
use special_functions,only:normalized
use special_functions,only:derivative
use special_functions,only:print_functions
use special_functions,only:print_wronskian
use special_functions,only:print_norms
use special_functions,only:print_factors
use special_functions,only:input_values
use special_functions,only:test_wron
use special_functions,only:norm
use special_functions,only:arg
use special_functions,only:scale_factor
use special_functions,only:log_factor
use special_functions,only:wron
use special_functions,only:factor
use special_functions,only:l
use special_functions,only:m
use special_functions,only:m_sign
use special_functions,only:s_fac
use special_functions,only:smallest
use special_functions,only:tiny
use special_functions,only:biggest
use special_functions,only:huge
use special_functions,only:eps
use special_functions,only:upper
use special_functions,only:lower
use special_functions,only:step
use special_functions,only:row_label
use special_functions,only:col_label
use special_functions,only:title
use special_functions,only:control
use special_functions,only:recur
use special_functions,only:directive
use special_functions,only:xi
use special_functions,only:eta
use special_functions,only:reg_l
use special_functions,only:reg_m
use special_functions,only:reg_lm
use special_functions,only:irreg_l
use special_functions,only:irreg_m
use special_functions,only:irreg_lm
use special_functions,only:up
use special_functions,only:down_a
use special_functions,only:down_b
use special_functions,only:down
use special_functions,only:cf_legendre
use special_functions,only:coefficients
use special_functions,only:legendre_functions
use special_functions,only:normalization
use special_functions,only:leg
contains
subroutine factorials()
end
subroutine wronskian(r_lm,i_lm,nrmlm)
type(reg_lm)::r_lm
type(irreg_lm)::i_lm
type(normalization)::nrmlm(0_8:int(m_max,kind=8))
end
subroutine normalization_factors(nrmlm)
type(normalization)::nrmlm(0_8:int(m_max,kind=8))
end
subroutine print_norm_factors(nrmlm)
type(normalization)::nrmlm(0_8:int(m_max,kind=8))
end
subroutine renormalize(f_lm,nrmlm)
real(8)::f_lm(0_8:int(l_max,kind=8),int(int_zero,kind=8):int(m_max,kind=8))
type(normalization)::nrmlm(0_8:int(m_max,kind=8))
end
end
