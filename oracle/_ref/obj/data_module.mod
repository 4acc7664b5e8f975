﻿!mod$ v1 sum:40fa78096c51d7cb
!need$ 2c37ccdf5d34d40d n accuracy
!need$ 49f150a7136fb138 n input_output
module data_module
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
real(8)::pi
real(8)::two_pi
real(8)::zero
real(8)::quarter
real(8)::half
real(8)::third
real(8)::fourth
real(8)::fifth
real(8)::sixth
real(8)::seventh
real(8)::eighth
real(8)::ninth
real(8)::tenth
real(8)::one
real(8)::two
real(8)::three
real(8)::four
real(8)::five
real(8)::six
real(8)::seven
real(8)::eight
real(8)::nine
real(8)::ten
real(8)::nrzero
real(8)::sqrt2
intrinsic::sqrt
real(8)::a_fac
real(8)::b_fac
integer(4)::int_zero
integer(4)::int_one
integer(4)::int_two
integer(4)::int_three
integer(4)::int_four
integer(4)::int_five
integer(4)::int_six
integer(4)::int_seven
integer(4)::int_eight
integer(4)::int_nine
integer(4)::int_ten
integer(4)::int_eleven
integer(4)::int_twelve
integer(4)::int_thirteen
integer(4)::int_fourteen
integer(4)::int_fifteen
integer(4)::int_sixteen
integer(4)::int_seventeen
integer(4)::int_eighteen
integer(4)::int_nineteen
integer(4)::int_twenty
integer(4)::int_max
real(8)::hbar
real(8)::massau
real(8)::lenau
real(8)::timau
real(8)::efieldau
real(8)::electric_field_to_intensity
real(8)::peak_electric_field
real(8)::pmass
real(8)::massn2p
real(8)::au_in_ev
end
