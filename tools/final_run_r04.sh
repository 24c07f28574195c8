mkdir -p gpurun_out/r04
bash tools/profile_r04.sh r04 "pbs ep ep2 lvl2 ks cb unf" > gpurun_out/r04_profile_final.log 2>&1
python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
python bench.py --no-cpu-baseline > gpurun_out/r04/bench_final_b.json 2>/dev/null
python tools/bench_configs.py > gpurun_out/r04/configs_final.jsonl 2> gpurun_out/r04/configs_final.err
python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu_final2.txt 2>&1
tail -3 gpurun_out/r04/pytest_gpu_final2.txt
cut -c1-300 gpurun_out/r04/bench_final.json
