#!/bin/bash
# tools/gpu.sh <name> '<command>': run one command on the MI355X box through gpurun, log to gpurun_out/<name>.log
# (replaces the numbered one-shot scripts of earlier rounds; see git history for those)
name=$1; shift
mkdir -p gpurun_out
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1500} -- "mkdir -p gpurun_out; ( $* ) > gpurun_out/$name.log 2>&1; tail -c 3000 gpurun_out/$name.log"
