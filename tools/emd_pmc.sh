#!/bin/bash
# gpurun -- bash tools/emd_pmc.sh TAG B N regime : SQ / LDS counters per launch of one fused EMD call
set -o pipefail
TAG=${1:-x}; B=${2:-64}; N=${3:-2048}; REG=${4:-rigid}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/emdpmc_$TAG
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/emd_levels.py run $B $N $REG $O/counts.json > $O/run_a.log 2>&1 || { tail -5 $O/run_a.log; exit 1; }
python3 $R/tools/emd_levels.py pmc $O/a > $O/table_a.txt && cut -c1-200 $O/table_a.txt
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/emd_levels.py run $B $N $REG $O/counts.json > $O/run_b.log 2>&1 || { tail -5 $O/run_b.log; exit 1; }
python3 $R/tools/emd_levels.py pmc $O/b > $O/table_b.txt && cut -c1-200 $O/table_b.txt
rm -f $O/*/*/*.db
