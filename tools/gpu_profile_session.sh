#!/bin/bash
# One GPU session for the tracked profiles/ of a round:
#     bash tools/gpu_profile_session.sh r03                         headline workload (python bench.py)
#     bash tools/gpu_profile_session.sh r03 cmu_v8_bf16 "--precision bf16 --views 8 --depth 2"    another workload
#   1. bench line (N=1)                                             -> gpurun_out/prof_$TAG[_$SUF]/bench.json
#   2. rocprofv3 kernel trace + stats of the SAME command           -> .../trace/
#   3. rocprofv3 PMC passes (each its own run, --kernel-trace only) over the same command: the kernels profiled are
#      the ones the forward runs (h2_stack_kernel, spt3_kernel, fuse_head_kernel, ...)
# tools/make_profiles.py $TAG [$SUF] turns the outputs into profiles/$TAG_*.
TAG=${1:-r03}
SUF=$2
ARGS=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG${SUF:+_$SUF}
mkdir -p $O; export TMPDIR=/tmp
cd $R
python -c "from openmpl_amd import build; print(build.source_hash())" > $O/srchash.txt
if [ -z "$SUF" ]; then
  python bench.py > $O/bench.log 2>&1
else
  python bench.py $ARGS --no-extra --no-cpu-baseline > $O/bench.log 2>&1
fi
grep '^{"metric"' $O/bench.log | tail -1 > $O/bench.json
cd /tmp
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra $ARGS"
echo "$CMD" | sed "s#$R/##" > $O/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a -o p -- $CMD > $O/pmc_a.log 2>&1
$P --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU TCC_HIT_sum TCC_MISS_sum -d $O/pmc_b -o p -- $CMD > $O/pmc_b.log 2>&1
$P --pmc FETCH_SIZE -d $O/pmc_c -o p -- $CMD > $O/pmc_c.log 2>&1
$P --pmc WRITE_SIZE -d $O/pmc_d -o p -- $CMD > $O/pmc_d.log 2>&1
# gpurun merges at most 64 MiB back: keep what tools/make_profiles.py reads, drop the raw traces
find $O -type f ! -name 't_kernel_stats.csv' ! -name 'p_counter_collection.csv' ! -name 'bench.json' ! -name 'cmd.txt' ! -name 'srchash.txt' ! -name 'bench.log' -delete
du -sh $O
