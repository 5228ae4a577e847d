( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json; tail -4 gpurun_out/bench_default.err
