#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of the default bench command plus separate PMC passes
# (HBM traffic of the dominant kernels) -> gpurun_out/prof_<tag>/ ; summarise with tools/summarize_profiles.py.
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
if [ "${TRACE_ONLY:-0}" = "1" ]; then rm -rf $OUT/trace $OUT/trace_bf16; else rm -rf $OUT; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the HEADLINE loop only (one geometry per kernel: the averages of the trace are the per-launch times of the bench line), then
# the same for BASELINE configs[2] (--dtype bf16)
# the step forms are PINNED to what 'tune' picks outside the profiler (critic step replayed, generator step launched eagerly with its
# forward one batch ahead): under rocprofv3 the timing comparison comes out differently, and the trace and the bench line must time ONE program
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline --graph-critic on --graph-generator off"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $LEGS --steps 100 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "trace done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bf16 -- python3 bench.py --dtype bf16 $LEGS --steps 100 --warmup 20 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
echo "bf16 trace done" >> $OUT/progress.txt
if [ "${TRACE_ONLY:-0}" = "1" ]; then
  python3 tools/summarize_profiles.py $TAG $OUT/summary > $OUT/summary.log 2>&1
  rm -f $OUT/*/runc/*_kernel_trace.csv $OUT/*/runc/*_domain_stats.csv
  exit 0
fi
# HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own (TCC slots), kernel trace only beside them
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_split -- python3 tools/split_probe.py 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_split -- python3 tools/split_probe.py 3 > /dev/null 2>&1
echo "conv1d pmc done" >> $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_dense -- python3 tools/dense_split_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_dense -- python3 tools/dense_split_probe.py > /dev/null 2>&1
echo "dense pmc done" >> $OUT/progress.txt
# the frequency-domain context Conv1D (tools/conv1d_fft_probe.py: forward with and without a kernel update, weight gradient)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c1fft -- python3 tools/conv1d_fft_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c1fft -- python3 tools/conv1d_fft_probe.py > /dev/null 2>&1
echo "conv1d fft pmc done" >> $OUT/progress.txt
export C2M_ONLY=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_conv -- python3 tools/conv2d_mfma_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_conv -- python3 tools/conv2d_mfma_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq_conv -- python3 tools/conv2d_mfma_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/pmc_lds_conv -- python3 tools/conv2d_mfma_probe.py > /dev/null 2>&1
unset C2M_ONLY
echo "conv pmc done" >> $OUT/progress.txt
# the fused conv-stack kernels (csrc/conv2d_chain.hip): SQ / LDS counters and HBM traffic, summarised by the script itself
bash tools/chain_pmc.sh > $OUT/chain_pmc.log 2>&1
cp gpurun_out/chain_pmc/summary.json $OUT/chain_summary.json
echo "chain pmc done" >> $OUT/progress.txt
# summarise on the box (the raw traces are too large to travel back) into gpurun_out/prof_<tag>/summary/
python3 tools/summarize_profiles.py $TAG $OUT/summary > $OUT/summary.log 2>&1
rm -f $OUT/*/runc/*_kernel_trace.csv $OUT/*/runc/*_counter_collection.csv $OUT/*/runc/*_domain_stats.csv
ls -R $OUT | head -40
