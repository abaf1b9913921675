set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ks_r5; mkdir -p $O && cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/ks.log 2>&1 || { tail -5 $O/ks.log; exit 1; }
f=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f $O/kernel_stats.csv
head -40 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/ks
