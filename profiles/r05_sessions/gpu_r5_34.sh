#!/bin/bash
# session 34: C2 and C4's per-rank shape, the round's last tree against the tree this session started from (5bb8d1e), interleaved
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--workload c2" cur= start=build_var/start_of_session.so > gpurun_out/r5_34_ab_c2.txt 2>&1
tail -3 gpurun_out/r5_34_ab_c2.txt
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--workload c4 --reads 20000000" cur= start=build_var/start_of_session.so > gpurun_out/r5_34_ab_c4.txt 2>&1
tail -3 gpurun_out/r5_34_ab_c4.txt
