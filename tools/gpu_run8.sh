set -u
mkdir -p gpurun_out/r02h
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02h/pytest.txt
tail -4 gpurun_out/r02h/pytest.txt
python tools/ab/bench_kernels.py turbo --decoder bcjr --no-cpu-baseline > gpurun_out/r02h/bcjr.txt 2>&1; cat gpurun_out/r02h/bcjr.txt
