#!/bin/bash
# round 3, call 2: (a) the full suite with the scaled-up fuzz; (b) parity of the turbo variants (pair-interleaved perm, vote loop);
# (c) per-kernel times of every variant on the chain workload, base first and last
cd /root/repo
mkdir -p gpurun_out/r03b
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q --durations=6 -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -14 > gpurun_out/r03b/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" >> gpurun_out/r03b/pytest_gpu.txt
tail -14 gpurun_out/r03b/pytest_gpu.txt
cp gpurun_out/fuzz_report.json gpurun_out/r03b/ 2>/dev/null
cp openlte_amd/libmi_lte.so /tmp/lib_keep.so
cp _ko/lib_PPVN4.so openlte_amd/libmi_lte.so
echo "== parity with lib_PPVN4"
timeout 600 python -m pytest tests/test_turbo_gpu.py tests/test_chain_gpu.py tests/test_fuzz_gpu.py tests/test_uplink_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r03b/pytest_ppvn4.txt
cp /tmp/lib_keep.so openlte_amd/libmi_lte.so
AB_TIMEOUT=120 bash tools/ab/run_variants.sh chain --steps 8 --warmup 2 2>&1 | tee gpurun_out/r03b/variants_chain.txt
