#!/bin/bash
# end of round 2: the full GPU suite once more, and the chain-in-BCJR-mode profile with the fused rate un-matching (tag r02j)
cd /root/repo
mkdir -p gpurun_out/r02j
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee gpurun_out/r02j/pytest.txt
bash tools/profile_bench.sh r02j_chain_bcjr --workload chain --decoder bcjr > /dev/null 2>&1
ls gpurun_out/prof_r02j_chain_bcjr/trace | head -3
