for m in 0 1 2; do NDLQR_UPPER=$m python bench.py --no-cpu 2>/dev/null > gpurun_out/bench_up$m.json; done
