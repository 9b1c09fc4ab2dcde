#!/bin/bash
# Development: another build of the library beside the product's, e.g.
#   scripts/build_variant.sh phiprofile "-DPHI_PROFILE"   ->  spatialcore_amd/libvar_phiprofile.so
# (use with SC_LIB=... scripts/generator_probe.py or scripts/bench_variant.py)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/sc_var_$name; mkdir -p $out
rm -f $out/*.o
for f in sc_api sc_moran sc_graph sc_perm sc_permgen sc_comm sc_lee; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $extra -c $root/spatialcore_amd/csrc/$f.hip -o $out/$f.o &
done
wait
for f in sc_api sc_moran sc_graph sc_perm sc_permgen sc_comm sc_lee; do test -f $out/$f.o || { echo "build_variant: $f.hip did not compile"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/spatialcore_amd/libvar_$name.so $out/*.o -ldl -pthread
echo built spatialcore_amd/libvar_$name.so
