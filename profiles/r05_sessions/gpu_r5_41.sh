#!/bin/bash
# session 41: what k_part_dedup's waves wait for -- timing builds (results wrong by construction): 1 no wait for the global cursor,
# 2 no length histogram, 3 no records read or inserted
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python tools/ab_multi.py --rounds 2 cur= nocursor=build_var/dedup_exp1.so nohist=build_var/dedup_exp2.so norecords=build_var/dedup_exp3.so > gpurun_out/r5_41_ab.txt 2>&1
tail -6 gpurun_out/r5_41_ab.txt
