set -u
mkdir -p gpurun_out/r02g
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" > gpurun_out/r02g/pytest.txt
tail -5 gpurun_out/r02g/pytest.txt
python tools/ab/bench_kernels.py chain --no-cpu-baseline > gpurun_out/r02g/chain.txt 2>&1; cat gpurun_out/r02g/chain.txt
python tools/ab/bench_kernels.py chain --no-cpu-baseline --no-kernel-events > gpurun_out/r02g/chain_noev.txt 2>&1; tail -2 gpurun_out/r02g/chain_noev.txt
python tools/ab/bench_kernels.py uplink --no-cpu-baseline > gpurun_out/r02g/uplink.txt 2>&1; cat gpurun_out/r02g/uplink.txt
python tools/ab/bench_kernels.py chain --decoder bcjr --no-cpu-baseline > gpurun_out/r02g/chain_bcjr.txt 2>&1; cat gpurun_out/r02g/chain_bcjr.txt
