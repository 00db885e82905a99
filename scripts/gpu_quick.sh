#!/bin/bash
# dev helper: GPU parity tests + a short bench (run through scripts/gpurun_retry.sh)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --streams 64 --frames 100 --steps 2 --warmup 1 --no-extras 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: print(l.strip()); continue
    print('value',j['value'],'MS/s  ms/step',j['ms_per_step'],'kernel_ms',j['kernel_ms'],'check',j['check'])
"
