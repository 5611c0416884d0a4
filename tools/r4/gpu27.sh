#!/bin/bash
# round 4, call 27: the profile set of the workloads the new 2048-point transform is part of
cd /root/repo
TAG=r04
for w in "chain --workload chain" "frontend2 --workload frontend2" "uplink --workload uplink"; do
  set -- $w; t=$1; shift
  rm -rf gpurun_out/prof_${TAG}_$t
  timeout 600 bash tools/profile_bench.sh ${TAG}_$t "$@" > /dev/null 2>&1
done
timeout 300 bash tools/pmc_sq.sh ${TAG}a_chain --workload chain 2>&1 | grep "k_" > gpurun_out/sq_chain_r04.txt
SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" timeout 300 bash tools/pmc_sq.sh ${TAG}b_chain --workload chain 2>&1 | grep "k_" >> gpurun_out/sq_chain_r04.txt
timeout 300 bash tools/pmc_sq.sh ${TAG}a_uplink --workload uplink 2>&1 | grep "k_" > gpurun_out/sq_uplink_r04.txt
timeout 300 python bench.py --workload uplink > gpurun_out/r04_bench_uplink.json 2>/dev/null
timeout 300 python bench.py --workload frontend > gpurun_out/r04_bench_frontend.json 2>/dev/null
timeout 120 tools/ubench/store_rate 2>&1 | tee gpurun_out/r04v/store_rate.txt
