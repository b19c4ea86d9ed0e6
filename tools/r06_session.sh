#!/bin/bash
# The round-6 profile sessions (one gpurun call): every BASELINE configuration + the shapes the review named.
#   gpurun -- 'bash tools/r06_session.sh [names...]'      then here: python tools/make_profiles.py r06 [<name>]
# (at most THREE sessions per gpurun call: it merges 64 MiB back, nine sessions lost everything once)
cd "$(dirname "$0")/.."
ALL="cmu_v8_bf16_l2 cmu_v8_bf16_l12 cmu_v8_fp32_l12 full v2_b256 v2_b1 kptok_v31 v31"
for n in ${@:-$ALL}; do
  case $n in
    head) bash tools/gpu_profile_session.sh r06 ;;
    cmu_v8_bf16_l2) bash tools/gpu_profile_session.sh r06 $n "--precision bf16 --views 8 --depth 2" ;;
    cmu_v8_bf16_l12) bash tools/gpu_profile_session.sh r06 $n "--precision bf16 --views 8 --depth 12" ;;
    cmu_v8_fp32_l12) bash tools/gpu_profile_session.sh r06 $n "--views 8 --depth 12" ;;
    full) bash tools/gpu_profile_session.sh r06 $n "--flagset full" ;;
    v2_b256) bash tools/gpu_profile_session.sh r06 $n "--views 2 --batch 256" ;;
    v2_b1) bash tools/gpu_profile_session.sh r06 $n "--views 2 --batch 1" ;;
    kptok_v31) bash tools/gpu_profile_session.sh r06 $n "--flagset kptok --views 31 --batch 256" ;;
    v31) bash tools/gpu_profile_session.sh r06 $n "--views 31 --batch 256" ;;
  esac > /dev/null 2>&1
  d=gpurun_out/prof_r06_$n; [ "$n" = head ] && d=gpurun_out/prof_r06
  echo "$n: $(cut -c1-160 $d/bench.json 2>/dev/null | head -1) [$(du -sh $d 2>/dev/null | cut -f1)]"
done
