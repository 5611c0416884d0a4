#!/bin/bash
cd /root/repo
o=gpurun_out/r04c; mkdir -p $o
timeout 300 tools/ubench/issue_mix 2>&1 | tee $o/ubench_issue_mix.txt
