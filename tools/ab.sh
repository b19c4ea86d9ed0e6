#!/bin/bash
# A/B of prebuilt library variants inside ONE gpurun call (boxes differ by a few per cent, so only same-call numbers compare):
#   bash tools/build_variants.sh [-f file.hip] base="" x="-DH2_WT_AUX=0" ...        (here, no GPU: -> build_tmp/lib_<tag>.so)
#   gpurun -- 'bash tools/ab.sh "<command>" [-r rounds] base x ...'                  (on the GPU box)
# Installs build_tmp/lib_<tag>.so as the library (the source-hash stamp stays valid, so nothing rebuilds), runs <command>,
# prefixes every output line with the tag; `rounds` alternations (default 2).  The installed library is restored at the end.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
CMD=$1; shift
R=2
if [ "$1" = "-r" ]; then R=$2; shift 2; fi
L=openmpl_amd/lib/libmpl_hip.so
cp $L /tmp/libmpl_hip.keep.so
for r in $(seq 1 $R); do
  for tag in "$@"; do
    cp build_tmp/lib_$tag.so $L
    timeout 600 bash -c "$CMD" 2>&1 | sed "s/^/[$tag] /"
  done
done
cp /tmp/libmpl_hip.keep.so $L
