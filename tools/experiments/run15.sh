cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-2500
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; tail -1 $O/bench_gpus2.json | cut -c1-900; tail -3 $O/bench_gpus2.err
timeout 600 python bench.py --gpus 2 --parallel views --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_views2.json 2> $O/bench_views2.err; tail -1 $O/bench_views2.json | cut -c1-600; tail -3 $O/bench_views2.err
timeout 600 python bench.py --config multiviewx --arch resnet50 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -1 $O/bench_cfg3.json | cut -c1-900; tail -3 $O/bench_cfg3.err
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_msda_gpu.py --deselect tests/test_fullsize_gpu.py 2>&1 | tail -8 | tee $O/pytest_rest.txt
