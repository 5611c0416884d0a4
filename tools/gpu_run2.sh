set -u
mkdir -p gpurun_out/r02b
python -m pytest tests/test_chain_gpu.py tests/test_frontend_gpu.py -m gpu -q 2>&1 | grep -v "^ERROR: DCI" | tail -15 > gpurun_out/r02b/pytest.txt
tail -3 gpurun_out/r02b/pytest.txt
python tools/ab/bench_kernels.py chain --no-cpu-baseline --ce full > gpurun_out/r02b/ab.txt 2>&1
bash tools/ab/run_variants.sh chain >> gpurun_out/r02b/ab.txt 2>&1
cat gpurun_out/r02b/ab.txt
