#!/bin/bash
# Prebuild library variants of ONE source file (default h2_gemm.hip) for A/B runs inside one gpurun call:
#   bash tools/build_variants.sh [-f file.hip] tag1="-DH2_DBG=1" tag2="-DH2_ABL=2" ...   ->  build_tmp/lib_<tag>.so
# (-DMPL_LAB is passed for the varied file: the H2_* switches are compile errors in the product build, csrc/h2_phase.hpp)
# The other objects are compiled once (build_tmp/obj).  On the GPU box: cp build_tmp/lib_<tag>.so openmpl_amd/lib/libmpl_hip.so
# (the source hash stamp of the default build stays valid, so cabi.load() does not rebuild).
set -e
cd "$(dirname "$0")/.."
F=h2_gemm.hip
if [ "$1" = "-f" ]; then F=$2; shift 2; fi
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
mkdir -p build_tmp/obj
SRCS=$(python -c "from openmpl_amd import build; print(' '.join(build.SOURCES))")
for s in $SRCS; do
  o=build_tmp/obj/${s%.hip}.o
  if [ "$s" != "$F" ] && { [ ! -f $o ] || [ openmpl_amd/csrc/$s -nt $o ] || [ openmpl_amd/csrc/common.hpp -nt $o ] || [ openmpl_amd/csrc/gemm_common.hpp -nt $o ] || [ openmpl_amd/csrc/h2_phase.hpp -nt $o ] || [ include/mpl_hip.h -nt $o ]; }; then
    $CC -c openmpl_amd/csrc/$s -o $o &
  fi
done
wait
n=0
for v in "$@"; do
  tag=${v%%=*}; flags=${v#*=}
  ( $CC -DMPL_LAB $flags -c openmpl_amd/csrc/$F -o build_tmp/obj/${F%.hip}.$tag.o
    objs=""
    for s in $SRCS; do if [ "$s" = "$F" ]; then objs="$objs build_tmp/obj/${F%.hip}.$tag.o"; else objs="$objs build_tmp/obj/${s%.hip}.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/lib_$tag.so $objs
    echo "built build_tmp/lib_$tag.so ($flags)" ) &
  n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
done
wait
