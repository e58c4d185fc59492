#!/bin/bash
# session 40: k_part_dedup with its shares' counts and offsets loaded together (one barrier, no serial prefix): parity, then A/B on C3 and the Zipf sample
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size_c3 or one_call" > gpurun_out/r5_40_tests.txt 2>&1; tail -2 gpurun_out/r5_40_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 cur= before=build_var/dedup_before.so > gpurun_out/r5_40_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_40_ab_c3.txt
timeout 1500 python tools/ab_multi.py --rounds 4 --bench-args "--pool 600000" cur= before=build_var/dedup_before.so > gpurun_out/r5_40_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_40_ab_zipf.txt
