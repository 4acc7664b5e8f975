#!/bin/bash
# Builds the one part of the reference that compiles from its own sources in this image:
# the Fortran Legendre-function library (src/legendre/*.f90, the list of
# /root/reference/src/CMakeLists.txt:24-30 minus legendretable.cpp which needs Armadillo).
# Sources are compiled where they lie; outputs go ONLY to oracle/_ref/ (git-ignored).
# The rest of the reference's hot path includes <armadillo>/<xc.h>/<gsl/...>, none of which
# exist in this image, so it is unbuildable here.
#
# Used only by tests / the fixture generator (tests/golden/make_legendre_golden.py) as a
# checker for the product's own P_L^M / Q_L^M implementation.  Never linked into the product.
set -e
REF=${HELFEM_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src/legendre" ]; then
  echo "build_ref: reference tree not present, skipping (prebuilt files are used if any)"; exit 0
fi
FC=${FC:-/opt/rocm/bin/amdflang}
if ! command -v "$FC" >/dev/null 2>&1; then echo "build_ref: no Fortran compiler, skipping"; exit 0; fi
mkdir -p "$OUT/obj"
cd "$OUT/obj"
SRCS="accuracy input_output itoc Matrix_Print Data_Module Special_Functions Auxilliary_Subroutines Prolate_Functions Lentz_Thompson Associated_Legendre_Functions Ass_Leg_Poly Legendre_Wrapper"
OBJS=""
for s in $SRCS; do
  "$FC" -O2 -fPIC -c "$REF/src/legendre/$s.f90" -o "$s.o"
  OBJS="$OBJS $s.o"
done
"$FC" -shared -o "$OUT/libref_legendre.so" $OBJS
echo "build_ref: built $OUT/libref_legendre.so"
