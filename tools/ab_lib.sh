# bench.py --no-cpu with the default library and with alternative builds (NDLQR_LIBRARY)
python bench.py --no-cpu 2>/dev/null > gpurun_out/ab_0.json
for lib in "$@"; do NDLQR_LIBRARY=$PWD/$lib python bench.py --no-cpu 2>/dev/null > gpurun_out/ab_$(basename $lib .so).json; done
