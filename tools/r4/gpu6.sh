#!/bin/bash
# round 4, call 6: whole GPU suite, the default bench line, the strong-scaling form on one device (oversubscribed), libfftw3f on the box?
cd /root/repo
o=gpurun_out/r04f; mkdir -p $o
(ldconfig -p | grep -i fftw || echo "no libfftw3 on the GPU box (ldconfig -p)") | tee $o/fftw_on_gpu_box.txt
ls /sys/devices/system/node/ 2>/dev/null | tr '\n' ' ' | tee $o/numa.txt; echo; for d in /sys/bus/pci/devices/*; do if [ -f $d/numa_node ] && grep -q 0x1002 $d/vendor 2>/dev/null; then echo "$d class $(cat $d/class) numa $(cat $d/numa_node)"; fi; done | head -20 | tee -a $o/numa.txt
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -6 | tee $o/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a $o/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; tail -c 600 $o/bench_default.json; tail -3 $o/bench_default.err
timeout 300 python bench.py --strong --gpus 2 --oversubscribe --steps 2 --warmup 1 --units 8000 > $o/bench_strong_oversubscribed_2.json 2> $o/strong.err; tail -c 1500 $o/bench_strong_oversubscribed_2.json; tail -3 $o/strong.err
