#!/bin/bash
# round 4, last call: whole GPU suite + smoke + the default line on the final tree, then six more seeds of the differential tests
cd /root/repo
o=gpurun_out/r04w; mkdir -p $o; rm -f $o/bench_kernels.txt
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee $o/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a $o/pytest_gpu.txt
cp gpurun_out/fuzz_report.json $o/fuzz_report.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; tail -c 200 $o/bench_default.json; tail -3 $o/bench_default.err
timeout 300 python bench.py --workload sync > gpurun_out/r04_bench_sync.json 2>/dev/null
rm -rf gpurun_out/fuzz_soak
bash tools/r3/fuzz_soak.sh 87 92
