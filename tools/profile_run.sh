#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command, then the PMC passes (tools/pmc_run.sh); summaries land in gpurun_out/.
# usage (GPU box, repo root): bash tools/profile_run.sh <tag>
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_trace.log
f=$(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_trace
cd $R && bash tools/pmc_run.sh ${TAG}_pmc 262144 > gpurun_out/${TAG}_pmc.log 2>&1
rm -rf gpurun_out/${TAG}_pmc_SQ_WAVES gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_SQ_INSTS_SALU gpurun_out/${TAG}_pmc_SQ_INSTS_VALU_MFMA_MOPS_I8
