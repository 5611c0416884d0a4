set -u
mkdir -p gpurun_out/r02e
python -m pytest tests/test_bcjr_gpu.py tests/test_chain_gpu.py tests/test_turbo_gpu.py -m gpu -q 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02e/pytest.txt
tail -15 gpurun_out/r02e/pytest.txt
bash tools/ab/run_variants.sh turbo --decoder bcjr > gpurun_out/r02e/ab_bcjr.txt 2>&1
cat gpurun_out/r02e/ab_bcjr.txt
python tools/ab/bench_kernels.py chain --no-cpu-baseline > gpurun_out/r02e/chain.txt 2>&1; cat gpurun_out/r02e/chain.txt
