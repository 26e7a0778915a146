for v in "" en_NO_SUMS en_all; do
  if [ -z "$v" ]; then lib=graphnets.jl_amd/libgnx.so; else lib=graphnets.jl_amd/libgnx_$v.so; fi
  GNX_LIB_PATH=$PWD/$lib python bench.py --dims core --no-secondary --no-cpu-baseline --no-c-abi --steps 10 --warmup 3 > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
  python -c "
import json
l=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('$v', l['ms_per_step'], l['roofline']['all_kernels_us']['k_rows_gemm_edge'])"
done
