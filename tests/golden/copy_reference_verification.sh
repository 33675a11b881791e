#!/bin/sh
# Regenerates tests/golden/reference_verification/ (data files only) from the reference tree.
set -e
REF=${1:-/root/reference}
OUT=$(dirname "$0")/reference_verification
mkdir -p "$OUT"
cp "$REF/Exec/hydro_tests/Sedov/Verification/spherical_sedov.dat" "$OUT/"
for f in sod test2 test3; do cp "$REF/Exec/hydro_tests/Sod/Verification/$f-exact.out" "$OUT/"; done
