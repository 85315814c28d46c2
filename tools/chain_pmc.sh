#!/bin/bash
# Run ON THE GPU BOX: SQ / LDS counters of the conv2d_chain kernels (separate --pmc passes), summarised per kernel.
set -u
OUT=gpurun_out/chain_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CHAIN_PROBE_SHORT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -- python3 tools/chain_probe.py > $OUT/p0.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 tools/chain_probe.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --output-format csv -d $OUT/p2 -- python3 tools/chain_probe.py > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/p3 -- python3 tools/chain_probe.py > $OUT/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- python3 tools/chain_probe.py > $OUT/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -- python3 tools/chain_probe.py > $OUT/p5.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for p in ('p0', 'p1', 'p2', 'p3', 'p4', 'p5'):
    for f in glob.glob('gpurun_out/chain_pmc/%s/*/*_counter_collection.csv' % p):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in acc.items():
            if 'c2c' in k:
                res[k].update({c: round(sum(v) / len(v)) for c, v in d.items()})
    for f in glob.glob('gpurun_out/chain_pmc/%s/*/*_kernel_stats.csv' % p):
        for r in csv.DictReader(open(f)):
            if 'c2c' in r['Name']:
                res[r['Name'].split('(')[0][:70]].update({'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3})
for k, d in res.items():
    print(k, json.dumps(d))
json.dump(res, open('gpurun_out/chain_pmc/summary.json', 'w'), indent=1)
PY
tail -2 $OUT/p1.log
