#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-host-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o k -- python bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o k -- python bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
ls -la $OUT/trace/* | head
