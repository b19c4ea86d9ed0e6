#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s5; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kp -o t -- python3 $R/tools/kptok_prof.py > $O/kp.log 2>&1
cat $O/kp/*/t_kernel_stats.csv | head -12
cd $R; timeout 600 python -m pytest tests/test_h2_gpu.py tests/test_failures_gpu.py tests/test_metrics.py -x -q 2>&1 | tail -3
timeout 300 python tools/batch_curve.py 2>&1 | grep -v amdgpu.ids
