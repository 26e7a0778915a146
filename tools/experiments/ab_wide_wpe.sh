for round in 1 2; do
  for tag in main wpe2; do
    if [ "$tag" = main ]; then unset GNX_LIB_PATH; else export GNX_LIB_PATH=graphnets.jl_amd/libgnx_$tag.so; fi
    bw=$(python tools/experiments/bw_time.py 2>/dev/null | grep -E "fwd\+bwd|bw_dx_ff2|bw_dx_ff1|bw_ff1_recompute|k_rows_gemm_node" | tr '\n' ' ' | sed 's/  */ /g')
    c4=$(python bench.py --model c4 --no-cpu-baseline --no-c-abi --full-line --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_us_one_forward'].get('k_rows_gemm_node'))")
    echo "round $round $tag: $bw | c4 $c4"
  done
done
