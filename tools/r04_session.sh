set -x
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04e_tests.log; cat gpurun_out/r04e_tests.log
bash tools/gpu_profile_session.sh r04 > /dev/null 2>&1
bash tools/gpu_profile_session.sh r04 full "--flagset full" > /dev/null 2>&1
bash tools/gpu_profile_session.sh r04 cmu_v8_bf16_l2 "--precision bf16 --views 8 --depth 2" > /dev/null 2>&1
bash tools/gpu_profile_session.sh r04 cmu_v8_bf16_l12 "--precision bf16 --views 8 --depth 12" > /dev/null 2>&1
bash tools/gpu_profile_session.sh r04 cmu_v8_fp32_l12 "--views 8 --depth 12" > /dev/null 2>&1
ls gpurun_out/prof_r04*; du -sh gpurun_out
