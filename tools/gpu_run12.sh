#!/bin/bash
# chain-only profile refresh (tag r02i): kernel trace + HBM counters + SQ counters
cd /root/repo
bash tools/profile_bench.sh r02i_chain --workload chain > /dev/null 2>&1
bash tools/pmc_sq.sh r02ia_chain --workload chain 2>&1 | grep "k_" > gpurun_out/sq_chain_r02i.txt
SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh r02ib_chain --workload chain 2>&1 | grep "k_" >> gpurun_out/sq_chain_r02i.txt
wc -l gpurun_out/sq_chain_r02i.txt
