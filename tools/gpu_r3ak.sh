#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python bench.py ) > gpurun_out/r3_bench_default.json 2> gpurun_out/r3ak_bench.err
cut -c1-300 gpurun_out/r3_bench_default.json
