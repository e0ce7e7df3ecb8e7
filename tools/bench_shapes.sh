# bench.py --no-cpu over the other BASELINE shapes (developer tool, GPU box)
python bench.py --no-cpu --nx 12 --nu 4 --horizon 1024 --batch 512 2>/dev/null > gpurun_out/shape_c4.json
python bench.py --no-cpu --nx 6 --nu 3 --horizon 256 --batch 1024 2>/dev/null > gpurun_out/shape_c2b.json
python bench.py --no-cpu --nx 6 --nu 3 --horizon 256 --batch 1 --steps 200 2>/dev/null > gpurun_out/shape_c2.json
python bench.py --no-cpu --nx 4 --nu 2 --horizon 256 --batch 1024 2>/dev/null > gpurun_out/shape_42.json
python bench.py --no-cpu --nx 8 --nu 4 --horizon 256 --batch 1024 2>/dev/null > gpurun_out/shape_84.json
python bench.py --no-cpu --nx 13 --nu 4 --horizon 256 --batch 1024 2>/dev/null > gpurun_out/shape_134.json
python bench.py --no-cpu --nx 12 --nu 4 --horizon 256 --batch 1024 --flags 1 2>/dev/null > gpurun_out/shape_strict.json
python bench.py --no-cpu --nx 12 --nu 4 --horizon 256 --batch 1024 --flags 8 2>/dev/null > gpurun_out/shape_keep.json
