#!/bin/bash
# gpurun -- bash tools/listsum_pmc.sh : L2 / fabric read counters of the list-sum kernel (sa_point_l1_bwd_kernel) in a real step
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/listsum_pmc; mkdir -p $O && cd $R
rocprofv3 --list-avail 2>/dev/null | grep -E "TCC_HIT|TCC_MISS|TCC_EA0?_RDREQ|TCP_TCC_READ_REQ|TCC_REQ|TCC_READ|TCC_EA0?_WRREQ|TCC_BUBBLE" | head -40 > $O/avail.txt
cat $O/avail.txt | cut -c1-160 | head -30
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; continue; }
  python3 tools/pmc_kernel_table.py $O/$tag "sa_point_l1_bwd_kernel|pool_dgrad_kernel" > $O/$tag.txt
  cut -c1-220 $O/$tag.txt
  rm -f $O/$tag/*/*.db
done
