cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python3 tools/bench_knn_group.py --reps 3 > $O/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p2 -- python3 tools/bench_knn_group.py --reps 3 > $O/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p3 -- python3 tools/bench_knn_group.py --reps 3 > $O/p3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/p4 -- python3 tools/bench_knn_group.py --reps 3 > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do echo "== $p"; python3 tools/pmc_kernel_table.py $O/$p "knn_select|group_fwd|knn_group_pad|knn32" 2>&1 | cut -c1-400; done > $O/tables.txt
cat $O/tables.txt
rm -rf $O/p1/*/*.db $O/p2/*/*.db $O/p3/*/*.db $O/p4/*/*.db 2>/dev/null
du -sh $O
