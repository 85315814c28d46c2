#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of the default bench command plus separate PMC passes
# (HBM traffic of the dominant kernels) -> gpurun_out/prof_<tag>/ ; summarise with tools/summarize_profiles.py.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "trace done" > $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python tools/gemm_probe.py both 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python tools/gemm_probe.py both 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_split -- python tools/split_probe.py 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_split -- python tools/split_probe.py 3 > /dev/null 2>&1
echo "gemm pmc done" >> $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_conv -- python tools/conv2d_probe.py 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_conv -- python tools/conv2d_probe.py 3 > /dev/null 2>&1
echo "conv pmc done" >> $OUT/progress.txt
ls -R $OUT | head -40
