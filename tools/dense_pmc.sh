#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the dense.hip kernels (separate --pmc passes) -> gpurun_out/dense_pmc/
OUT=gpurun_out/dense_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 tools/dense_split_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 tools/dense_split_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('sq', 'lds'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('gpurun_out/dense_pmc/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            if 'dns::' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k in sorted(agg):
        print(d, k)
        for c, v in sorted(agg[k].items()): print('    {:28s} {:14.0f} per launch'.format(c, v / max(1, cnt[(k, c)])))
PY
rm -rf $OUT/sq $OUT/lds
