#!/bin/bash
# the 8 x 16 x 16 plan of the 2048-point transform: parity first, then against the 8 x 8 x 8 x 4 plan (lib_fft_old) on the same box
cd /root/repo
o=gpurun_out/r04s; mkdir -p $o
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_sync_gpu.py tests/test_uplink_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -8 | tee $o/pytest_frontend.txt
AB_TIMEOUT=150 tools/ab/run_variants.sh frontend --steps 20 --warmup 5 2>&1 | tee $o/variants_frontend.txt
AB_TIMEOUT=150 tools/ab/run_variants.sh uplink --steps 10 --warmup 3 2>&1 | tee $o/variants_uplink.txt
