#!/bin/bash
# session 30: the CLI's route with -gff at the bench sample's size: where its time goes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python bench.py --steps 3 --warmup 2 --cpu-baseline 0 --pmc 0 --read-sets 0 --two-in-flight 0 --min-seconds 0.2 2>gpurun_out/r5_30.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cli_path']; print(json.dumps(c.get('gff_libraries_resident'))); print(json.dumps(c.get('libraries_resident')))" | tee gpurun_out/r5_30_gff_cli.txt
tail -3 gpurun_out/r5_30.err
