#!/bin/bash
# round 4, call 10: the profile set (kernel trace + HBM counters + SQ counters) of the chain, the W2 two-port front end, W3 in BCJR mode and the uplink
cd /root/repo
TAG=r04
for w in "chain --workload chain" "frontend2 --workload frontend2" "turbo_bcjr --workload turbo --decoder bcjr" "turbo_bcjr_early --workload turbo --decoder bcjr_early" "uplink --workload uplink" "turbo --workload turbo"; do
  set -- $w; t=$1; shift
  timeout 600 bash tools/profile_bench.sh ${TAG}_$t "$@" > /dev/null 2>&1
done
timeout 300 bash tools/pmc_sq.sh ${TAG}a_chain --workload chain 2>&1 | grep "k_" > gpurun_out/sq_chain_r04.txt
SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" timeout 300 bash tools/pmc_sq.sh ${TAG}b_chain --workload chain 2>&1 | grep "k_" >> gpurun_out/sq_chain_r04.txt
timeout 300 bash tools/pmc_sq.sh ${TAG}a_turbo_bcjr --workload turbo --decoder bcjr 2>&1 | grep "k_" > gpurun_out/sq_turbo_bcjr_r04.txt
timeout 300 bash tools/pmc_sq.sh ${TAG}a_uplink --workload uplink 2>&1 | grep "k_" > gpurun_out/sq_uplink_r04.txt
ls gpurun_out | grep r04
