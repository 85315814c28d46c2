"""Summarise gpurun_out/prof_<tag>/ (made by tools/profile_round.sh) into profiles/<tag>_*: the rocprofv3
--kernel-trace --stats table of the default bench command, and HBM traffic per launch of the dominant kernels from
the separate FETCH_SIZE / WRITE_SIZE PMC passes (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half the
bytes of a wide coalesced read -> x2; counters are in KiB)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src = os.path.join('gpurun_out', 'prof_' + tag)
outdir = sys.argv[2] if len(sys.argv) > 2 else 'profiles'
os.makedirs(outdir, exist_ok=True)


def short(name):
    name = name.replace('void ', '').replace('ptts::', '')
    return name.split('(')[0][:70]


rows = list(csv.DictReader(open(max(glob.glob(src + '/trace/runc/*_kernel_stats.csv'), key=os.path.getmtime))))
total = sum(float(r['TotalDurationNs']) for r in rows)
lines = ['# rocprofv3 --kernel-trace --stats -- python bench.py (headline loop only: --no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline)   (round %s)' % tag,
         '# total kernel time %.1f ms' % (total / 1e6),
         '%8s %12s %11s %11s %11s %6s  %s' % ('calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%', 'kernel')]
for r in rows[:40]:
    lines.append('%8s %12.1f %11.1f %11.1f %11.1f %6.2f  %s' % (r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3,
                 float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage']), short(r['Name'])))
open(outdir + '/%s_kernel_stats.txt' % tag, 'w').write('\n'.join(lines) + '\n')
bench = open(src + '/bench.json').read().strip()
open(outdir + '/%s_bench_under_rocprof.json' % tag, 'w').write(bench + '\n')
# the same for the configs[2] run (--dtype bf16)
bf = glob.glob(src + '/trace_bf16/runc/*_kernel_stats.csv')
if bf:
    rows16 = list(csv.DictReader(open(max(bf, key=os.path.getmtime))))
    total16 = sum(float(r['TotalDurationNs']) for r in rows16)
    l16 = ['# rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 (headline loop only)   (round %s)' % tag,
           '# total kernel time %.1f ms' % (total16 / 1e6),
           '%8s %12s %11s %11s %11s %6s  %s' % ('calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%', 'kernel')]
    for r in rows16[:40]:
        l16.append('%8s %12.1f %11.1f %11.1f %11.1f %6.2f  %s' % (r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3,
                   float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage']), short(r['Name'])))
    open(outdir + '/%s_kernel_stats_bf16.txt' % tag, 'w').write('\n'.join(l16) + '\n')
    if os.path.exists(src + '/bench_bf16.json'):
        open(outdir + '/%s_bench_bf16_under_rocprof.json' % tag, 'w').write(open(src + '/bench_bf16.json').read().strip() + '\n')
if os.path.exists(src + '/chain_summary.json'):
    cs = json.load(open(src + '/chain_summary.json'))
    for k, d in cs.items():
        # counters are in KiB.  No x2 on FETCH_SIZE here: the forward kernel pins it -- its only reads are the fp32 spectrum (13.3 MB at
        # B = 128: 12 918 KiB counted) and its writes the 8 maps (215.8 MB: 210 800 KiB counted); the maps of the other kernels arrive by
        # global_load_lds_dwordx4, 16 bytes per lane like the dwordx4 loads the guide's correction is about, and are counted in full too
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            d['hbm_bytes_per_launch'] = (d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
            d['hbm_read_bytes'], d['hbm_write_bytes'] = d['FETCH_SIZE'] * 1024, d['WRITE_SIZE'] * 1024
            d['hbm_bytes_per_launch_with_guide_x2_on_fetch'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
    cs['_note'] = ('tools/chain_probe.py at [B,400,65], L = 8, averaged over its launches (B = 64 / 128 / 192, mean 128).  Algorithmic bytes at B = 128: '
                   'forward 13.3 MB read + 215.8 MB written; backward-data 242.4 MB read + 13.3 MB (g0) + 216.2 MB (gamma maps) written; '
                   'backward 242.4 MB read, partial rows written (its 88 MB of writes are the scratch slots of the dynamically indexed half of the dW accumulators); '
                   'second order 13.3 + 2 x 215.8 MB read + 26.6 MB written.')
    json.dump(cs, open(outdir + '/%s_conv2d_chain_counters.json' % tag, 'w'), indent=1, sort_keys=True)


def pmc(dirname, counter):
    f = sorted(glob.glob(src + '/' + dirname + '/runc/*_counter_collection.csv'), key=os.path.getmtime)[-1:]
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != counter:
            continue
        out.setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']))
    return out


traffic = {}
for fdir, wdir in (('pmc_fetch', 'pmc_write'), ('pmc_fetch_conv', 'pmc_write_conv'), ('pmc_fetch_split', 'pmc_write_split'), ('pmc_fetch_dense', 'pmc_write_dense')):
    if not glob.glob(src + '/' + fdir + '/runc/*_counter_collection.csv'):
        continue
    fe, wr = pmc(fdir, 'FETCH_SIZE'), pmc(wdir, 'WRITE_SIZE')
    for k in fe:
        if not ('gemm' in k or 'conv2d' in k or 'split3' in k or 'wgrad' in k or 'c2m' in k or 'dense' in k) or k in traffic:
            continue
        f_kib = sum(fe[k]) / len(fe[k])
        w_kib = sum(wr.get(k, [0.0])) / max(1, len(wr.get(k, [0.0])))
        traffic[k] = {'fetch_bytes_raw': f_kib * 1024, 'fetch_bytes_corrected_x2': 2 * f_kib * 1024, 'write_bytes': w_kib * 1024,
                      'hbm_bytes_per_launch': (2 * f_kib + w_kib) * 1024, 'launches_averaged': len(fe[k])}
# round 4: HBM bytes of the critic's whole Conv2D stack per critic step (bench.py's `roofline.traffic`): the per-launch bytes of the
# stack's kernels at [64,400,65,4] (tools/conv2d_mfma_probe.py) x their launches per critic step, a launch over the stacked 2B batch
# counted twice; the two 1 -> 4 / 4 -> 1 stencil layers at their algorithmic bytes (no counter pass of their own)
def _k(prefix):
    return next((traffic[k]['hbm_bytes_per_launch'] for k in traffic if k.startswith(prefix)), None)
parts = {'forward 4->4 (7 layers: real + fake stacked 2B, x^ at B)': (_k('c2m::fwd_ws_kernel<1, false'), 7 * (2 + 1)),
         'backward-data of the gradient penalty (7 layers at B)': (_k('c2m::fwd_ws_kernel<0, true'), 7),
         'fused dx + dW + dbias (7 layers at 2B)': (_k('c2m::bwd_ws_kernel<1'), 7 * 2),
         'fused masked forward + dW of the second-order sweep (7 layers at B)': (_k('c2m::bwd_ws_kernel<2'), 7)}
if all(v[0] is not None for v in parts.values()):
    A1 = 64 * 400 * 65 * 4.0
    stencil = (A1 + 4 * A1) * 3 + (2 * 4 * A1 + A1) * 2 + (4 * A1 + A1) + (2 * 4 * A1 + A1)      # 1 -> 4 layer: forward x 3B, dW (+ dx for x^) ...
    tot = sum(v[0] * v[1] for v in parts.values()) + stencil
    traffic['critic_conv2d_stack_per_critic_step'] = {
        'hbm_bytes': tot, 'algorithmic_bytes': (149.0 * 4 + 3) * 64 * 400 * 65 * 4,
        'parts': {k: {'hbm_bytes_per_launch_at_B64': v[0], 'units_of_B_per_critic_step': v[1]} for k, v in parts.items()},
        'first_layer_1to4_and_its_backward_algorithmic': stencil,
        'note': 'sum over the stack\'s kernels of (FETCH_SIZE x 2 + WRITE_SIZE per launch at B = 64) x (launches per critic step, a 2B launch counted twice)'}
json.dump(traffic, open(outdir + '/%s_traffic.json' % tag, 'w'), indent=1, sort_keys=True)
# the frequency-domain Conv1D's kernels (tools/conv1d_fft_probe.py), in a file of their own: the batched products share their template
# names with the Dense layers' launches of other shapes.  Raw and x2-corrected fetch bytes both given (see the chain counters' note).
if glob.glob(src + '/pmc_fetch_c1fft/runc/*_counter_collection.csv'):
    fe, wr = pmc('pmc_fetch_c1fft', 'FETCH_SIZE'), pmc('pmc_write_c1fft', 'WRITE_SIZE')
    c1 = {}
    for k in fe:
        if not any(t in k for t in ('dense_bf16x6', 'split3_dense', 'wdft', 'dft_mirror', 'transpose_batched', 'wgrad_inverse', 'wgrad_partials_sum', 'gemm_bf16x6', 'wgrad_bf16x6')):
            continue
        f_kib = sum(fe[k]) / len(fe[k])
        w_kib = sum(wr.get(k, [0.0])) / max(1, len(wr.get(k, [0.0])))
        c1[k] = {'fetch_bytes_raw': f_kib * 1024, 'fetch_bytes_corrected_x2': 2 * f_kib * 1024, 'write_bytes': w_kib * 1024, 'launches_averaged': len(fe[k])}
    c1['_note'] = ('tools/conv1d_fft_probe.py at B = 64, T = 400, Cin = 601, N = 256, KW = 21: dense_bf16x6_kernel<0, false, MT, 3> with MT = 8 are the '
                   'per-frequency products (forward [128 x 1216].[1216 x 256] and correlation [256 x 128].[128 x 1216], averaged together), MT = 7 the DFT of x / dy, '
                   'MT = 5 the inverse DFT; algorithmic bytes of the forward per-frequency launch: 394 MB of kernel planes + 131 MB of X^ + 27 MB written.')
    json.dump(c1, open(outdir + '/%s_conv1d_freq_traffic.json' % tag, 'w'), indent=1, sort_keys=True)
# SQ / LDS counters of the conv2d_mfma kernels (averages per launch)
sq = {}
for d in ('pmc_sq_conv', 'pmc_lds_conv'):
    for f in glob.glob(src + '/' + d + '/runc/*_counter_collection.csv'):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if 'c2m' not in k:
                continue
            acc.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for k, dct in acc.items():
            sq.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in dct.items()})
if sq:
    json.dump(sq, open(outdir + '/%s_conv2d_mfma_counters.json' % tag, 'w'), indent=1, sort_keys=True)
print(open(outdir + '/%s_kernel_stats.txt' % tag).read()[:3500])
print(json.dumps(traffic, indent=1)[:3000])
