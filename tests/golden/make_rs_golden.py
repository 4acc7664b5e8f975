"""Known answers for the range-separated exchange path (atomic program), computed with mpmath at 40 digits and
written to tests/golden/rs_special.json.  Nothing here comes from the reference's sources; the reference has no
test vectors for this path, these are independent mathematical anchors:

  * modified spherical Bessel functions i_L(x), k_L(x) in the convention of libhelfem/src/utils.cpp:47-70
    (k_L = GSL k_l / (pi/2), i.e. k_0 = exp(-x)/x);
  * Phi_n(Xi, xi) of the Legendre expansion erfc(mu r12)/r12 = mu sum_n Phi_n(mu r, mu r') P_n(cos gamma), by direct
    numerical integration of the definition (no use of the closed forms of Angyan et al.);
  * attenuation functions F(a) of the short-range LDA exchange, erf and Yukawa (closed forms at 40 digits);
  * self-interaction of the hydrogen 1s density rho = exp(-2r)/pi with the screened interactions,
    J_w = int int rho rho w(r12) = (2/pi) int_0^inf k^2 w~(k)/(4 pi) (1 + k^2/4)^-4 dk:
      Yukawa  w~ = 4 pi/(k^2 + lambda^2),  erfc  w~ = 4 pi (1 - exp(-k^2/(4 mu^2)))/k^2.

Run:  python tests/golden/make_rs_golden.py
"""
import json
import os

import mpmath as mp

mp.mp.dps = 40
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rs_special.json")


def s(x):
    return mp.nstr(x, 20)


bessel = []
for L in (0, 1, 2, 3, 5, 8, 12):
    for x in ("1e-6", "1e-3", "0.01", "0.3", "1.0", "2.5", "7.0", "15.0", "29.0", "31.0", "45.0", "120.0"):
        xv = mp.mpf(x)
        i_ref = mp.sqrt(mp.pi / (2 * xv)) * mp.besseli(L + mp.mpf(1) / 2, xv)
        k_ref = mp.sqrt(mp.pi / (2 * xv)) * mp.besselk(L + mp.mpf(1) / 2, xv) * 2 / mp.pi
        bessel.append({"L": L, "x": float(xv), "il": s(i_ref), "kl": s(k_ref)})


def phi_num(n, Xi, xi):
    def f(t):
        r = mp.sqrt(Xi ** 2 + xi ** 2 - 2 * Xi * xi * t)
        return mp.erfc(r) / r * mp.legendre(n, t)
    return (2 * n + 1) / mp.mpf(2) * mp.quad(f, [-1, 0, 1])


phi = []
for n in range(0, 7):
    for Xi, xi in (("0.1", "0.05"), ("0.45", "0.35"), ("0.6", "0.39"), ("0.6", "0.41"), ("1.5", "0.39"), ("1.5", "1.2"),
                   ("3.0", "0.3"), ("3.0", "2.9"), ("0.49", "0.48"), ("0.51", "0.45"), ("5", "4.5"), ("2.0", "0.01"),
                   ("0.3", "1e-4")):
        phi.append({"n": n, "Xi": float(mp.mpf(Xi)), "xi": float(mp.mpf(xi)),
                    "phi": s(phi_num(n, mp.mpf(float(mp.mpf(Xi))), mp.mpf(float(mp.mpf(xi)))))})


def F_erf(a):
    e = mp.e ** (-1 / (4 * a * a))
    return 1 - mp.mpf(8) / 3 * a * (mp.sqrt(mp.pi) * mp.erf(1 / (2 * a)) + (2 * a - 4 * a ** 3) * e - 3 * a + 4 * a ** 3)


def F_yuk(a):
    return 1 - mp.mpf(8) / 3 * a * (mp.atan(1 / a) + a / 4 - a / 4 * (a * a + 3) * mp.log(1 + 1 / (a * a)))


mp.mp.dps = 80
att = []
for a in ("1e-3", "0.05", "0.3", "1.0", "1.9", "2.1", "3.9", "4.1", "10", "100", "2000"):
    av = mp.mpf(float(mp.mpf(a)))
    att.append({"a": float(av), "erf": s(F_erf(av)), "yukawa": s(F_yuk(av))})
mp.mp.dps = 40

h1s = []
for lam in ("0.4", "1.0"):
    lv = mp.mpf(float(mp.mpf(lam)))
    Jy = (2 / mp.pi) * mp.quad(lambda k: k ** 2 / (k ** 2 + lv ** 2) * (1 + k ** 2 / 4) ** -4, [0, 1, 10, mp.inf])
    Je = (2 / mp.pi) * mp.quad(lambda k: (1 - mp.e ** (-k ** 2 / (4 * lv ** 2))) * (1 + k ** 2 / 4) ** -4, [0, 1, 10, mp.inf])
    h1s.append({"omega": float(lv), "J_yukawa": s(Jy), "J_erfc": s(Je)})

json.dump({"bessel": bessel, "phi": phi, "attenuation": att, "h1s": h1s}, open(OUT, "w"), indent=0)
print("wrote", OUT)
