#!/bin/bash
# round 4, closing call: the whole GPU suite + smoke on the final tree
cd /root/repo
o=gpurun_out/r04w; mkdir -p $o
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee $o/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a $o/pytest_gpu.txt
cp gpurun_out/fuzz_report.json $o/fuzz_report.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/smoke.txt
