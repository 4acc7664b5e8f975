// Functional name / id handling of the drivers (reference: src/general/dftfuncs.cpp:64-118 parse_xc_func,
// :388-428 exact_exchange, :464-520 is_range_separated).  libxc is absent; the ids below are libxc's
// and only the functionals implemented in hip/xc_device.h are accepted.
#pragma once
#include <string>

namespace helfem {
/// "HF" -> (-1,0); "none" -> (0,0); "x-c" keyword pair or numeric ids
void parse_xc_func(int &x_func, int &c_func, const std::string &method);
/// fraction of exact exchange: 1 for HF, 0 for the pure functionals available here
double exact_exchange(int x_func);
/// range separation of the exchange functional: omega, fraction alpha of full-range and beta of short-range exact
/// exchange (reference: range_separation, dftfuncs.cpp:505); omega = 0 for everything but the range-separated hybrids
void range_separation(int x_func, double &omega, double &alpha, double &beta);
/// which screened kernel the functional uses (reference: is_range_separated, dftfuncs.cpp:464)
void is_range_separated(int x_func, bool &erf, bool &yukawa);
const char *xc_func_name(int func_id);
}  // namespace helfem
