#!/bin/bash
# tools/profile_round.sh <tag>: the round's profile set on the GPU box (under gpurun_out/; tools/summarize_profile.py and tools/sq_json.py turn it
# into profiles/<tag>_*, profiles/pmc_traffic_*.json and profiles/sq_counters_*.json -- run them here, the box has the data):
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes for chain / chain-mixed / frontend / uplink, SQ counters for chain, chain-mixed and uplink,
# one bench line per workload.
set -u
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for w in chain chain-mixed frontend uplink; do
  n=${w//-/_}
  bash tools/profile_bench.sh ${TAG}_$n --workload $w > /dev/null 2>&1
  python tools/summarize_profile.py gpurun_out/prof_${TAG}_$n ${TAG}_$n $n > /dev/null 2>&1
done
for w in chain chain-mixed uplink; do
  n=${w//-/_}
  bash tools/pmc_sq.sh ${TAG}a_$n --workload $w 2>&1 | grep "k_" > gpurun_out/sq_${TAG}_$n.txt
  SQ_COUNTERS="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" bash tools/pmc_sq.sh ${TAG}b_$n --workload $w 2>&1 | grep "k_" >> gpurun_out/sq_${TAG}_$n.txt
  steps=3; [ "$n" = chain_mixed ] && steps=6   # (the mixed workload's own closing check runs three more steps with a plan assignment each)
  python tools/sq_json.py gpurun_out/sq_${TAG}_$n.txt $n gpurun_out/pmc_${TAG}a_$n/bench.json $steps > /dev/null
  cp gpurun_out/sq_${TAG}_$n.txt profiles/${TAG}_${n}_sq_counters.txt
done
python tools/trace_by_shape.py gpurun_out/prof_${TAG}_chain_mixed/trace | grep -v k_rm_rank > profiles/${TAG}_chain_mixed_by_shape.txt
mkdir -p gpurun_out/bench_$TAG gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/pmc_traffic_*.json profiles/sq_counters_*.json gpurun_out/profiles_$TAG/ 2>/dev/null
for w in chain-mixed uplink turbo frontend frontend2 control sync; do python bench.py --workload $w > gpurun_out/bench_$TAG/${w//-/_}.json 2> gpurun_out/bench_$TAG/${w//-/_}.err; done
python bench.py > gpurun_out/bench_$TAG/chain.json 2> gpurun_out/bench_$TAG/chain.err
ls gpurun_out/bench_$TAG gpurun_out/profiles_$TAG
