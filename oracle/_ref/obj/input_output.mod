﻿!mod$ v1 sum:49f150a7136fb138
module input_output
integer(4)::inp
integer(4)::iout
integer(4)::rows_to_print
integer(4)::columns_to_print
integer(4)::eigenvectors_to_print
logical(4)::print_parameter
character(8_4,1),allocatable::rowlab(:)
character(8_4,1),allocatable::collab(:)
end
