#!/bin/bash
# LDS counters of the matrix-core kernels and the overlap picture of the two-stream step (profiles/r4_lds_counters.txt,
# r4_trace_busy.txt):  gpurun -- bash tools/collect_lds_and_overlap.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r4x; mkdir -p $O && cd $R
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/lds -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lds.log 2>&1 || { tail -5 $O/pmc_lds.log; exit 1; }
python3 tools/pmc_kernel_table.py $O/lds "attn_fwd_kernel|attn_bwd|attn_proj|sa_level_stream_kernel|outproj_maxpts_kernel|pool_|df_wgrad" > $O/lds_table.txt
cut -c1-230 $O/lds_table.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
python3 tools/trace_busy.py $O/kt 10 > $O/trace_busy.txt 2>&1
cat $O/trace_busy.txt
rm -rf $O/kt $O/lds/*/*.db
