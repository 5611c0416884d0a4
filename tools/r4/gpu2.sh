#!/bin/bash
# round 4, call 2: the issue-rate table with more opcodes, wall-time based; the SWAR form of the trellis step against the packed one
cd /root/repo
o=gpurun_out/r04b; mkdir -p $o
timeout 600 tools/ubench/issue_rate 2>&1 | tee $o/ubench_issue_rate.txt
