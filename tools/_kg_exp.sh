mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_point_ops.py -m gpu -x -q > gpurun_out/r2i/pointops.log 2>&1; echo "rc=$?" >> gpurun_out/r2i/pointops.log
for blk in 512 1024; do PZN_KG_BLOCKS=$blk python tools/bench_knn_group.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('blocks $blk', {k:v['ms'] for k,v in d.items() if k.startswith('knn_group_N') or k.startswith('knn_N')})" >> gpurun_out/r2i/variants.txt; done
cat gpurun_out/r2i/variants.txt
tail -4 gpurun_out/r2i/pointops.log
