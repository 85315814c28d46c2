"""HIP-event time per C-ABI entry point (and per GEMM shape) for one critic step and one generator step at config 2."""
import sys, os, io, contextlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip, _hip

class A: batch = 64; frames = 400; ctx = 601
cfg = bench.make_cfg(A)
dev = backend_hip.device()
voc = vocoders.VocoderPML(16000, 0.005, 65, 20)
with contextlib.redirect_stdout(io.StringIO()):
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(601, voc, cfg)
    crit = networks_critic.Critic(voc, 601, cfg)
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
X, Y = bench.synthetic(64, 400, 601, 86, 65, 123, dev)
for name, fn in (('critic', lambda: opt.critic_step(X, Y)), ('generator', lambda: opt.generator_step(X, Y))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    tot = collections.Counter(); cnt = collections.Counter()
    reps = 3
    for _ in range(reps):
        with _hip.KernelTimer() as kt:
            fn()
        for n, tag, d in kt.durations_ms():
            if n == 'ptts_colstats' and tag:
                n = 'colstats rows{} C{} mode{}'.format(*tag)
            key = n if n != 'ptts_gemm' else 'gemm M{} N{} K{} tA{} tB{}{}'.format(*tag[:5], ' conv' if tag[5] else '')
            tot[key] += d / reps; cnt[key] += 1.0 / reps
    print('==', name, 'step: sum of C-ABI calls {:.2f} ms'.format(sum(tot.values())))
    for k, v in tot.most_common(45):
        print('  {:<46} {:7.3f} ms  x{:<5.0f} avg {:7.1f} us'.format(k, v, cnt[k], v / cnt[k] * 1e3))
