﻿!mod$ v1 sum:99f7e6bc3f7597b4
!need$ 40fa78096c51d7cb n data_module
!need$ 2c37ccdf5d34d40d n accuracy
module special_functions
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
real(8),allocatable::x(:)
real(8),allocatable::y(:)
integer(4)::m_max
integer(4)::m_min
integer(4)::l_max
integer(4)::n_points
logical(4)::normalized
logical(4)::derivative
logical(4)::print_functions
logical(4)::print_wronskian
logical(4)::print_norms
logical(4)::print_factors
logical(4)::input_values
logical(4)::test_wron
real(8)::norm
real(8)::arg
real(8)::scale_factor
real(8)::log_factor
real(8)::wron
real(8),allocatable::factor(:)
integer(4)::l
integer(4)::m
integer(4)::m_sign
integer(4)::s_fac
real(8)::smallest
intrinsic::tiny
real(8)::biggest
intrinsic::huge
real(8)::eps
real(8)::upper
real(8)::lower
real(8)::step
character(8_4,1),allocatable::row_label(:)
character(8_4,1),allocatable::col_label(:)
character(64_4,1)::title
character(24_4,1)::control
character(24_4,1)::recur
character(16_4,1)::directive
type::xi
real(8),allocatable::f_small(:,:)
real(8),allocatable::f_large(:,:)
end type
type::eta
real(8),allocatable::f_1(:,:)
real(8),allocatable::f_2(:,:)
end type
type::reg_l
real(8),allocatable::f(:)
real(8),allocatable::df(:)
end type
type::reg_m
real(8),allocatable::f(:)
real(8),allocatable::df(:)
end type
type::reg_lm
real(8),allocatable::f(:,:)
real(8),allocatable::df(:,:)
type(xi)::xi
type(eta)::eta
end type
type::irreg_l
real(8),allocatable::f(:)
real(8),allocatable::df(:)
end type
type::irreg_m
real(8),allocatable::f(:)
real(8),allocatable::df(:)
end type
type::irreg_lm
real(8),allocatable::f(:,:)
real(8),allocatable::df(:,:)
type(xi)::xi
type(eta)::eta
end type
type::up
character(24_4,1)::dir
end type
type::down_a
character(24_4,1)::dir
end type
type::down_b
character(24_4,1)::dir
end type
type::down
type(down_a)::a
type(down_b)::b
end type
type::cf_legendre
character(24_4,1)::dir
end type
type::coefficients
real(8),allocatable::a(:,:)
real(8),allocatable::b(:,:)
end type
type::legendre_functions
type(reg_l)::r_l
type(reg_m)::r_m
type(reg_lm)::r_lm
type(irreg_l)::i_l
type(irreg_m)::i_m
type(irreg_lm)::i_lm
type(up)::u
type(down)::d
type(coefficients)::c_q
end type
type::normalization
real(8),allocatable::leg_fac(:)
real(8),allocatable::norm(:)
integer(4)::maxlm
end type
type(legendre_functions)::leg
end
