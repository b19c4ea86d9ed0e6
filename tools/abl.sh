# bench-only: loop-time ablations of the x3 k loop (tools/x3_phase.py stamps), built on the GPU box
export TMPDIR=/tmp
for f in 0 1 2 3 5 7; do
  export MPL_HIPCC_FLAGS="-DX3_ABL=$f"   # exported: the measuring process hashes the flags too (build.source_hash)
  python -m openmpl_amd.build --force > /dev/null 2>&1
  echo "ABL=$f"; MPL_X3_LAUNCHES=1 timeout 120 python tools/x3_phase.py 2>&1 | tail -3 | cut -c1-200
done
