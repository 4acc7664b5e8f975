#include "dftfuncs.h"
#include <cctype>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <strings.h>

namespace helfem {

namespace {
struct Known {
  const char *name;
  int id;
};
const Known known[] = {{"lda_x", 1}, {"lda_c_vwn", 7}, {"lda_c_vwn_rpa", 8}, {"lda_c_pw", 12}, {"lda_c_pw_mod", 13},
                       {"gga_x_pbe", 101}, {"gga_c_pbe", 130}, {"gga_x_b88", 106}, {"gga_c_lyp", 131},
                       {"hyb_gga_xc_b3lyp", 402},  // 0.08 lda_x + 0.72 gga_x_b88 + 0.19 lda_c_vwn_rpa + 0.81 gga_c_lyp + 0.20 exact exchange
                       {"hyb_gga_xc_pbeh", 406},  // PBE0: 0.75 gga_x_pbe + gga_c_pbe + 0.25 exact exchange
                       {"mgga_x_tpss", 202}, {"mgga_c_tpss", 231}, {"lda_x_erf", 546}, {"lda_x_yukawa", 641},
                       {"hyb_lda_xc_cam_lda0", 178}};  // CAM-LDA0: erfc range separation, omega = 1/3

int find_func(const std::string &name) {
  if (name.empty()) throw std::runtime_error("empty functional name\n");
  if (isdigit(name[0])) return atoi(name.c_str());
  if (!strcasecmp(name.c_str(), "none")) return 0;
  if (!strcasecmp(name.c_str(), "hyb_x_hf") || !strcasecmp(name.c_str(), "HF")) return -1;
  for (const Known &k : known)
    if (!strcasecmp(name.c_str(), k.name)) return k.id;
  std::ostringstream oss;
  oss << "\nError: functional " << name << " is not available in this build!\n";
  throw std::runtime_error(oss.str());
}
}  // namespace

void parse_xc_func(int &x_func, int &c_func, const std::string &xc) {
  x_func = 0;
  c_func = 0;
  size_t dpos = xc.find('-', 0);
  if (dpos != std::string::npos) {
    x_func = find_func(xc.substr(0, dpos));
    c_func = find_func(xc.substr(dpos + 1));
  } else
    x_func = find_func(xc);
}

// fraction of exact exchange (libxc xc_hyb_exx_coef; dftfuncs.cpp:134-160 of the reference)
double exact_exchange(int x_func) {
  return x_func == -1 ? 1.0 : (x_func == 406 ? 0.25 : (x_func == 402 ? 0.20 : (x_func == 178 ? 0.5 : 0.0)));
}

// libxc's xc_hyb_cam_coef / hyb_type of the range-separated hybrids available here (dftfuncs.cpp:464-570 of the
// reference): K = alpha K[1/r12] + beta K[screened kernel]
void range_separation(int x_func, double &omega, double &alpha, double &beta) {
  omega = 0.0;
  alpha = exact_exchange(x_func);
  beta = 0.0;
  if (x_func == 178) {  // hyb_lda_xc_cam_lda0
    omega = 1.0 / 3.0;
    alpha = 0.5;
    beta = -0.25;
  }
}
void is_range_separated(int x_func, bool &erf, bool &yukawa) {
  erf = (x_func == 178);
  yukawa = false;
}

const char *xc_func_name(int id) {
  if (id == -1) return "HF";
  if (id == 0) return "none";
  for (const Known &k : known)
    if (k.id == id) return k.name;
  return "unknown";
}

}  // namespace helfem
