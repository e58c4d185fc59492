#!/bin/bash
# round 5, GPU session 10: soak of the GPU fuzzers on the final tree (cascades incl. the exact-pass family, collapses, texts, trimming
# incl. anchored / linked adapters, UMI options, join + CSVs, variant tally) + the new table-cache test
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( MIRGE_FUZZ_SEEDS=160 timeout 3300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r05_fuzz_soak.txt 2>&1
tail -3 gpurun_out/r05_fuzz_soak.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "whole_read_tables or exact_passes or documented_linked or backend_bowtie or weighted or bench_single" 2>&1 | tail -4 ) > gpurun_out/r5_10_tests.txt 2>&1
tail -3 gpurun_out/r5_10_tests.txt
