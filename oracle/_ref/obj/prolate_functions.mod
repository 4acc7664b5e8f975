﻿!mod$ v1 sum:6270394355f35b59
!need$ 40fa78096c51d7cb n data_module
!need$ 2c37ccdf5d34d40d n accuracy
module prolate_functions
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
integer(4)::lorder
integer(4)::morder
integer(4)::mabs
integer(4)::meo
real(8)::a
real(8)::r_int
real(8)::radius_moeq
real(8)::point(1_8:3_8,1_8:2_8)
real(8)::a_p(1_8:3_8)
real(8)::x_i(1_8:2_8)
real(8)::eta_i(1_8:2_8)
real(8)::rho_i(1_8:2_8)
real(8)::varphi(1_8:2_8)
real(8)::r(1_8:2_8)
real(8)::dr(1_8:3_8)
real(8)::xi_small
real(8)::xi_large
real(8)::facm
real(8)::vardm
real(8)::dl21
real(8)::temp
real(8)::csum_real
real(8)::csum_imag
real(8)::varphi_diff
real(8)::ctemp_real
real(8)::ctemp_imag
real(8)::rsqr
real(8)::r_12
real(8)::r_12_invs
end
