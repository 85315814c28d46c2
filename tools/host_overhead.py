import sys, os, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip
class A: batch = 64; frames = 400; ctx = 601
cfg = bench.make_cfg(A); cfg.train_wgan_parallel_streams = True
dev = backend_hip.device()
voc = vocoders.VocoderPML(16000, 0.005, 65, 20)
with contextlib.redirect_stdout(io.StringIO()):
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(601, voc, cfg); crit = networks_critic.Critic(voc, 601, cfg)
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit); opt.prepare()
X, Y = bench.synthetic(64, 400, 601, 86, 65, 123, dev)
for name, fn in (('critic', lambda: opt.critic_step(X, Y)), ('generator', lambda: opt.generator_step(X, Y))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): fn()
    th = (time.perf_counter() - t) / 5
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t) / 5
    print('{:<10} host enqueue {:.2f} ms per step, wall incl. GPU {:.2f} ms'.format(name, th * 1e3, tg * 1e3))
