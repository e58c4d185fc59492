set -x
timeout 1200 python -m pytest tests -m gpu -x -q -k "oracle or golden or fuzz or one_call or edge or join or granule or low_complex or staged or cli_end" 2>&1 | tail -5
KT_ONLY=1 bash profiles/collect.sh r03t > gpurun_out/collect.log 2>&1
python profiles/timeline.py r03t 2>&1 | tail -45 > gpurun_out/timeline_new.txt
rm -rf gpurun_out/prof_r03t
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --pmc 0 --cli-path 0 > gpurun_out/bench_new.json 2>gpurun_out/bench_new.err
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --pmc 0 --cli-path 0 > gpurun_out/bench_new2.json 2>>gpurun_out/bench_new.err
