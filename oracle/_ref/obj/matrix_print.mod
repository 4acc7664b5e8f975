﻿!mod$ v1 sum:0e9501db05b6b31a
module matrix_print
interface print_matrix
procedure::print_matrix_d
procedure::print_matrix_z
procedure::print_triangle_matrix_d
procedure::print_triangle_matrix_z
procedure::print_vector_d
procedure::print_vector_z
end interface
contains
subroutine print_matrix_d(a,n,m,iout,frmt,title,collab,rowlab)
real(8)::a(:,:)
integer(4)::n
integer(4)::m
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
character(*,1),optional::rowlab(:)
end
subroutine print_matrix_z(a,n,m,iout,frmt,title,collab,rowlab)
complex(8)::a(:,:)
integer(4)::n
integer(4)::m
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
character(*,1),optional::rowlab(:)
end
subroutine print_triangle_matrix_d(a,n,iout,frmt,title,collab)
real(8)::a(:)
integer(4)::n
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
end
subroutine print_triangle_matrix_z(a,n,iout,frmt,title,collab)
complex(8)::a(:)
integer(4)::n
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
end
subroutine print_vector_d(a,iout,frmt,title,collab)
real(8)::a(:)
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
end
subroutine print_vector_z(a,iout,frmt,title,collab)
complex(8)::a(:)
integer(4)::iout
character(*,1),optional::frmt
character(*,1),optional::title
character(*,1),optional::collab(:)
end
end
