#!/bin/bash
# Everything under profiles/ for one build, on the GPU box:  gpurun -- bash tools/collect_profiles.sh r4
# (1) bench line incl. cpu_baseline  (2) rocprofv3 --kernel-trace --stats of the same command  (3) FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes -> HBM bytes per kernel  (4) SQ counters of the named matrix-core kernel and of the stage kernel.
set -o pipefail
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O && cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done" ; tail -2 $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 bench.py --no-cpu-baseline --no-other-workloads > $O/bench_ks.json 2> $O/bench_ks.err || exit 1
python3 -c "import json, bench; json.dump({'build_id': bench.build_id(), 'batch': 64, 'points': 2048, 'attn': 'f32', 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-workloads'}, open('$O/kernel_stats.meta.json', 'w'))" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -o ks1 -- python3 bench.py --one-stream --no-cpu-baseline --no-other-workloads > $O/bench_ks1.json 2> $O/bench_ks1.err || exit 1
cp $O/kernel_stats.meta.json $O/kernel_stats_one_stream.meta.json
sed -i 's/bench.py --no-cpu-baseline/bench.py --one-stream --no-cpu-baseline/' $O/kernel_stats_one_stream.meta.json
echo "kernel stats done (two streams: ks/, one stream: ks1/)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_traffic.json > $O/pmc_traffic.txt || exit 1
cat $O/pmc_traffic.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_grbm.log 2>&1 || exit 1
python3 tools/pmc_kernel_table.py $O/sq "sa_level_stream_kernel|outproj_maxpts_kernel|point_mlp3|attn_fwd_kernel|attn_bwd|attn_proj|ws_gemm_kernel|knn_select_kernel|gemm_kernel<128|emdf_|pool_|stem_" > $O/sq_table.txt
python3 tools/pmc_kernel_table.py $O/grbm "sa_level_stream_kernel|outproj_maxpts_kernel|attn_fwd_kernel|attn_bwd|ws_gemm_kernel|knn_select_kernel" > $O/grbm_table.txt
cat $O/sq_table.txt | cut -c1-250
rm -f $O/*/*/*.db $O/*/*.db
du -sh $O
