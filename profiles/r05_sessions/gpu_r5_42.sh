#!/bin/bash
# session 42: the unique reads' length histogram by a kernel of its own on the side stream instead of inside k_part_dedup: parity, then A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size or one_call or unpack or baking or properties_at_scale or umi_route_at_scale or sorted_order" > gpurun_out/r5_42_tests.txt 2>&1; tail -2 gpurun_out/r5_42_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 cur= before=build_var/hist_in_dedup.so > gpurun_out/r5_42_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_42_ab_c3.txt
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--pool 600000" cur= before=build_var/hist_in_dedup.so > gpurun_out/r5_42_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_42_ab_zipf.txt
