#!/bin/bash
# On the GPU box: the profile set DESIGN.md 6 cites, under gpurun_out/ (tag = $1): kernel trace + HBM counters for chain / uplink / turbo (REF and
# BCJR), SQ instruction counters for chain / turbo-BCJR, and one bench line per workload.
set -u
TAG=${1:-r02h}
bash tools/profile_bench.sh ${TAG}_chain --workload chain > /dev/null 2>&1
bash tools/profile_bench.sh ${TAG}_uplink --workload uplink > /dev/null 2>&1
bash tools/profile_bench.sh ${TAG}_turbo --workload turbo > /dev/null 2>&1
bash tools/profile_bench.sh ${TAG}_turbo_bcjr --workload turbo --decoder bcjr > /dev/null 2>&1
bash tools/profile_bench.sh ${TAG}_chain_bcjr --workload chain --decoder bcjr > /dev/null 2>&1
bash tools/pmc_sq.sh ${TAG}a_chain --workload chain 2>&1 | grep "k_" > gpurun_out/sq_chain.txt
SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh ${TAG}b_chain --workload chain 2>&1 | grep "k_" >> gpurun_out/sq_chain.txt
bash tools/pmc_sq.sh ${TAG}a_turbo_bcjr --workload turbo --decoder bcjr 2>&1 | grep "k_" > gpurun_out/sq_turbo_bcjr.txt
mkdir -p gpurun_out/bench_$TAG
for w in uplink turbo frontend control sync; do python bench.py --workload $w > gpurun_out/bench_$TAG/$w.json 2> gpurun_out/bench_$TAG/$w.err; done
python bench.py --workload turbo --decoder bcjr > gpurun_out/bench_$TAG/turbo_bcjr.json 2>/dev/null
python bench.py --workload chain --decoder bcjr --no-cpu-baseline > gpurun_out/bench_$TAG/chain_bcjr.json 2>/dev/null
python bench.py --ce full --no-cpu-baseline --no-turbo-leg --no-host-leg > gpurun_out/bench_$TAG/chain_full_ce.json 2>/dev/null
python bench.py > gpurun_out/bench_$TAG/chain.json 2> gpurun_out/bench_$TAG/chain.err
ls gpurun_out/bench_$TAG
