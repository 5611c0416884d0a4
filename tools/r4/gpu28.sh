#!/bin/bash
# the smaller block-size groups on a side stream (MI_LTE_GROUP_STREAMS=0: one stream), same library, same box
cd /root/repo
o=gpurun_out/r04x; mkdir -p $o
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_pipeline_gpu.py tests/test_turbo_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_chain.txt
for gs in 0 1 0 1; do
  echo "== MI_LTE_GROUP_STREAMS=$gs"
  MI_LTE_GROUP_STREAMS=$gs timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
done | tee $o/group_streams.txt
MI_LTE_FUZZ_SEED=71 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k uplink 2>&1 | grep -v "^ERROR: DCI" | tail -2 | tee $o/pytest_ul_seed71.txt
