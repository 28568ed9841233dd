#!/bin/bash
mkdir -p gpurun_out/rows
python benchmarks/rules_stamps.py 3 2>&1 | grep -v Warning | tee gpurun_out/rows/rules_stamps.txt
python benchmarks/insitu_rules_timing.py 2>&1 | grep variant | tee gpurun_out/rows/insitu_with_planes.txt
