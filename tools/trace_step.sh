set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_r6; mkdir -p $O && cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 tools/trace_busy.py $O/kt 10 > $O/trace_busy.txt 2>&1
cat $O/trace_busy.txt
rm -rf $O/kt
