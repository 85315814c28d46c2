#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the conv2d bf16x6 probe kernels (two separate --pmc passes), summarised per kernel.
set -u
OUT=gpurun_out/probe_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- ./tools/conv2d_bf16x6_probe > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p2 -- ./tools/conv2d_bf16x6_probe > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ('p1', 'p2'):
    for f in glob.glob('gpurun_out/probe_pmc/%s/*/*_counter_collection.csv' % p):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in acc.items():
            print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
