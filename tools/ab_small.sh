# A/B of prebuilt library variants on the small-batch shapes inside ONE gpurun call:  bash tools/ab_small.sh build_tmp/lib_A.so build_tmp/lib_B.so
export TMPDIR=/tmp
L=openmpl_amd/lib/libmpl_hip.so
for v in "$@" "$1"; do
  cp "$v" $L
  echo "== $v"
  python -m pytest tests/test_h2_gpu.py -q -k small_batch 2>&1 | grep -E "passed|failed"
  python tools/small_batch.py 2>&1 | grep "V=" | sed 's/| team.*//'
done
