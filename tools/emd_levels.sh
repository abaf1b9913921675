#!/bin/bash
# gpurun -- bash tools/emd_levels.sh TAG [B n]...   per-launch EMD table (evaluations, time, gaps) for three regimes
set -o pipefail
TAG=${1:-x}; B=${2:-64}; N=${3:-2048}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/emdlv_$TAG
mkdir -p $O
for reg in rigid indep near; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$reg -- python3 $R/tools/emd_levels.py run $B $N $reg $O/counts_$reg.json > $O/run_$reg.log 2>&1 || { tail -5 $O/run_$reg.log; exit 1; }
  python3 $R/tools/emd_levels.py parse $O/$reg $O/counts_$reg.json > $O/table_$reg.txt || exit 1
  cat $O/table_$reg.txt
  rm -f $O/$reg/*/*.db
done
