#!/bin/bash
# Development: another build of the library beside the product's, e.g.
#   scripts/build_variant.sh phiprofile "-DPHI_PROFILE"   ->  spatialcore_amd/libvar_phiprofile.so
# (use with SC_LIB=... scripts/generator_probe.py or scripts/bench_variant.py)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/sc_var_$name; mkdir -p $out
for f in sc_api sc_moran sc_graph sc_perm sc_permgen sc_comm sc_lee; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $extra -c $root/spatialcore_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/spatialcore_amd/libvar_$name.so $out/*.o -ldl
echo built spatialcore_amd/libvar_$name.so
