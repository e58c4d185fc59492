#!/bin/bash
# session 25: the words of a wide read selected instead of indexed (no per-thread scratch in the 32-255-nt groups' kernels)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "vs_oracle or vs_bruteforce or low_complexity or edge_cases or golden_cascade or random_cascade or one_call or longer or 255 or exact_passes or properties_at_scale or full_size_c3" > gpurun_out/r5_25_tests.txt 2>&1
tail -3 gpurun_out/r5_25_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 cur= wide_index=build_var/wide_index.so > gpurun_out/r5_25_ab.txt 2>&1
tail -4 gpurun_out/r5_25_ab.txt
