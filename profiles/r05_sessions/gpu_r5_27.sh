#!/bin/bash
# session 27: what k_trim's general instance costs beside the exact one (same text, same adapter)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python bench.py --steps 3 --warmup 2 --cpu-baseline 0 --pmc 0 --cli-path 0 --read-sets 0 --two-in-flight 0 --min-seconds 0.2 2>gpurun_out/r5_27.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['fastq_trim_path']); print({k:v for k,v in d['kernels'].items() if 'trim' in k})"
