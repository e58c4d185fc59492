#!/bin/bash
# session 21: k_isotype without per-thread arrays: parity (golden GFF files, smoke's oracle check, the CLI's -gff runs), then timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_21_smoke.txt 2>&1; tail -2 gpurun_out/r5_21_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gff or isomir or isotype" > gpurun_out/r5_21_tests.txt 2>&1
tail -3 gpurun_out/r5_21_tests.txt
for cfg in "cur:" "w2:MIRGE_NATIVE_SO=build_var/iso_w2.so" "old:MIRGE_ISO_FAST=0" "cur:" "w2:MIRGE_NATIVE_SO=build_var/iso_w2.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 3 --warmup 2 --cpu-baseline 0 --pmc 0 --cli-path 0 --read-sets 0 --two-in-flight 0 --min-seconds 0.2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gff_typing']
print('$name', {k:g.get(k) for k in ('mirna_reads','isomir_records','ms_call','k_isotype_ms','error')}, g.get('roofline',{}).get('frac'))" >> gpurun_out/r5_21_ab.txt
done
cat gpurun_out/r5_21_ab.txt
