#!/bin/bash
cd /root/repo
o=gpurun_out/r04v; mkdir -p $o
timeout 120 tools/ubench/store_rate 2>&1 | tee $o/store_rate.txt
