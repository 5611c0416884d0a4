#!/bin/bash
cd /root/repo
o=gpurun_out/r04g; mkdir -p $o
timeout 300 tools/ubench/lds_rate 2>&1 | tee $o/ubench_lds_rate.txt
