set -u
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" | tail -25 > gpurun_out/r02a/pytest.txt
python tools/diag_qpsk.py > gpurun_out/r02a/diag_qpsk.txt 2>&1
python bench.py > gpurun_out/r02a/bench_chain.json 2> gpurun_out/r02a/bench_chain.err
python bench.py --workload turbo --decoder bcjr > gpurun_out/r02a/bench_turbo_bcjr.json 2> gpurun_out/r02a/bench_turbo_bcjr.err
bash tools/profile_bench.sh r02a_turbo_bcjr --workload turbo --decoder bcjr > gpurun_out/r02a/prof.log 2>&1
nproc > gpurun_out/r02a/host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r02a/host.txt; rocm-smi --showmeminfo vram 2>/dev/null | tail -5 >> gpurun_out/r02a/host.txt
tail -5 gpurun_out/r02a/pytest.txt; cat gpurun_out/r02a/diag_qpsk.txt | tail -12
