# round 5: the wave-uniform store fast path in k_edge_x6 / k_ffn_x6 / k_proj_x6 — block at core dims and config 4, this build
python bench.py --dims core --no-secondary --no-cpu-baseline --no-c-abi --steps 10 --warmup 3 > gpurun_out/fp_core.json 2>gpurun_out/fp_core.err
python -c "
import json
l=json.loads(open('gpurun_out/fp_core.json').read().strip().splitlines()[-1]); print('core dims', l['ms_per_step'], l['roofline']['all_kernels_us'])"
python bench.py --model c4 --no-cpu-baseline --no-c-abi --steps 5 --warmup 2 > gpurun_out/fp_c4.json 2>gpurun_out/fp_c4.err
python -c "
import json
l=json.loads(open('gpurun_out/fp_c4.json').read().strip().splitlines()[-1]); print('c4', l['ms_per_step'], l['config']['timing']); print(l['kernel_us_one_forward'])"
python tools/experiments/core_replay_time.py 2>&1 | tail -3
