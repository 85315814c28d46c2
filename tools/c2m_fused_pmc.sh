#!/bin/bash
# LDS bank conflicts of the fused backward kernels (run on the GPU box): C2M_ONLY probe under rocprofv3 --pmc
set -u
OUT=gpurun_out/c2m_fused_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export C2M_ONLY=1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/lds -- python3 tools/conv2d_mfma_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob('gpurun_out/c2m_fused_pmc/lds/*/*_counter_collection.csv') + glob.glob('gpurun_out/c2m_fused_pmc/lds/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-60:]
        acc.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'c2m' in k:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print(k, 'conflict %.1f %%' % (100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_LDS_IDX_ACTIVE', 1))), m)
PY
rm -rf $OUT/lds
