mkdir -p gpurun_out/r05i
for NAME in default if2k if8k if16k; do
  if [ "$NAME" = default ]; then unset CMI_GPU_LIBRARY; else export CMI_GPU_LIBRARY=$PWD/cmacionize_amd/variants/libcmi_gpu_$NAME.so; fi
  echo "== $NAME"
  python3 tools/run_config.py lexington 256 1e8 7 2>&1 | tail -n 1 | cut -c1-140
  python3 tools/run_config.py diffuse 256 1e8 9 2>&1 | tail -n 1 | cut -c1-140
done > gpurun_out/r05i/items.txt 2>&1
unset CMI_GPU_LIBRARY
for T in tile_refill_threshold=32 tile_refill_threshold=56 tile_refill_threshold=40; do
  echo "== $T"
  python3 tools/run_config.py lexington 256 1e8 7 $T 2>&1 | tail -n 1 | cut -c1-140
  python3 tools/run_config.py diffuse 256 1e8 9 $T 2>&1 | tail -n 1 | cut -c1-140
done > gpurun_out/r05i/refill.txt 2>&1
echo done
