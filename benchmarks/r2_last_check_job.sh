#!/bin/bash
# what the driver runs at round end: GPU tests, smoke(), the default bench line
mkdir -p gpurun_out/last
python -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/last/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/last/smoke.log
python bench.py --gpus 1 --steps 10 --warmup 1 2>gpurun_out/last/bench.err | tail -1 > gpurun_out/last/bench.json
python -c "
import json; d=json.loads(open('gpurun_out/last/bench.json').read()); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'], d['roofline_c3']['frac'])"
