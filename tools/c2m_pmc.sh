#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS counters of the conv2d_mfma kernels (separate --pmc passes), summarised per kernel.
set -u
OUT=gpurun_out/c2m_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export C2M_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -- python3 tools/conv2d_mfma_probe.py > $OUT/p0.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 tools/conv2d_mfma_probe.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --output-format csv -d $OUT/p2 -- python3 tools/conv2d_mfma_probe.py > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p3 -- python3 tools/conv2d_mfma_probe.py > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ('p0', 'p1', 'p2', 'p3'):
    for f in glob.glob('gpurun_out/c2m_pmc/%s/*/*_counter_collection.csv' % p):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in acc.items():
            if 'c2m' in k:
                print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
    for f in glob.glob('gpurun_out/c2m_pmc/%s/*/*_kernel_stats.csv' % p):
        for r in csv.DictReader(open(f)):
            if 'c2m' in r['Name']:
                print(p, r['Name'][:70], 'calls', r['Calls'], 'avg ns', r['AverageNs'])
PY
tail -3 $OUT/p1.log $OUT/p2.log $OUT/p3.log
