﻿!mod$ v1 sum:2c37ccdf5d34d40d
module accuracy
integer(4),parameter::isp=4_4
intrinsic::selected_real_kind
integer(4),parameter::int_sp=4_4
intrinsic::selected_int_kind
integer(4),parameter::int_dp=8_4
integer(4),parameter::idp=8_4
integer(4),parameter::iqp=16_4
end
