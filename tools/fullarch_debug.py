"""Per-tensor gradient errors of the B = 16 full-architecture critic step against the fp64 oracle, for the context-Conv1D variants
(frequency domain / time domain forward and weight gradient, deterministic mode): which part of an error is the kernel's arithmetic
and which is LeakyReLU masks flipping between fp32 and fp64."""
import io, contextlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import percival_oracle as O
import bench
from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, ops, _hip

T, CTX, SPEC, NM = 400, 601, 65, 20
class A: batch = 16; frames = T; ctx = CTX
cfg = bench.make_cfg(A); cfg.train_wgan_critic_LSWGANtransidx = 30.0
voc = vocoders.VocoderPML(16000, 0.005, SPEC, NM)
a = O.Arch(CTX, SPEC, NM, 256, 1, 21, 8, 4, 5, 5)
gw = O.random_weights(O.generator_weight_shapes(a), seed=11)
cw = O.random_weights(O.critic_weight_shapes(a), seed=12)
with contextlib.redirect_stdout(io.StringIO()):
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(CTX, voc, cfg)
    crit = networks_critic.Critic(voc, CTX, cfg)
    mod.kerasmodel.set_weights([w.numpy() for w in gw]); crit.model.set_weights([w.numpy() for w in cw])
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit); opt.prepare()
Bq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(42)
X = torch.rand(Bq, T, CTX, generator=g, dtype=torch.float64) * 2 - 1
Y = torch.randn(Bq, T, 86, generator=g, dtype=torch.float64)
Y[:, :, 1 + SPEC:] = torch.rand(Bq, T, NM, generator=g, dtype=torch.float64)
al = torch.rand(Bq, generator=g, dtype=torch.float64)
cwr = [w.detach().clone().requires_grad_(True) for w in cw]
total, parts = O.critic_step_loss(cwr, gw, a, X, Y, al)
grads = torch.autograd.grad(total, cwr, retain_graph=True)
f32 = lambda t: t.to(torch.float32).cuda().contiguous()
Xd, Yd, ald = f32(X), f32(Y), f32(al)

def run(label):
    ops.clear_caches()
    opt.critic_opti.zero_grad()
    with ops.deferred_weight_grads():
        tot, p = opt.critic_loss(Xd, Yd, ald, training=True)
        tot.backward()
    torch.cuda.synchronize()
    errs = []
    for prm, g_ in zip(opt.critic_opti.flat.params, grads):
        e = float((prm.grad.detach().cpu().double() - g_).norm()) / max(float(g_.norm()), 1e-30)
        errs.append((tuple(g_.shape), e))
    print(label, 'loss err', abs(float(tot) - float(total)) / abs(float(total)))
    print('   ', ' '.join('{}:{:.1e}'.format('x'.join(map(str, s)), e) for s, e in errs))
    return [prm.grad.detach().clone() for prm in opt.critic_opti.flat.params]


# per loss part: the total's context-branch gradient is the small difference of the valid and fake parts
pg = {k: torch.autograd.grad(parts[k], cwr, retain_graph=True, allow_unused=True) for k in ('valid', 'fake', 'gp')}
for ki, k in enumerate(('valid', 'fake', 'gp')):
    ops.clear_caches(); opt.critic_opti.zero_grad()
    with ops.deferred_weight_grads():
        tot, p = opt.critic_loss(Xd, Yd, ald, training=True)
        p[ki].backward()
    torch.cuda.synchronize()
    out = []
    for prm, g_ in zip(opt.critic_opti.flat.params, pg[k]):
        if g_ is None:
            out.append('{}:None({:.1e})'.format('x'.join(map(str, prm.shape)), float(prm.grad.norm()))); continue
        out.append('{}:{:.1e}'.format('x'.join(map(str, g_.shape)), float((prm.grad.detach().cpu().double() - g_).norm()) / max(float(g_.norm()), 1e-30)))
    print('part', k, ' '.join(out))
print('cancellation: |g_total| / (|g_valid| + |g_fake| + |g_gp|) per tensor:',
      ' '.join('{:.1e}'.format(float(gt.norm()) / max(1e-30, sum(float(pg[k][i].norm()) for k in pg if pg[k][i] is not None))) for i, gt in enumerate(grads)))
# the oracle itself in fp32 against fp64: the conditioning of the quantity
cw32 = [w.detach().float().requires_grad_(True) for w in cw]
t32, _ = O.critic_step_loss(cw32, [w.float() for w in gw], a, X.float(), Y.float(), al.float())
g32 = torch.autograd.grad(t32, cw32)
print('fp32 ORACLE vs fp64 oracle:', ' '.join('{:.1e}'.format(float((x.double() - y).norm()) / max(float(y.norm()), 1e-30)) for x, y in zip(g32, grads)))
sys.exit(0)
g0 = run('default')
ops._C1FFT.wgrad_enabled = False
g1 = run('fft fwd, time wgrad')
ops._C1FFT.wgrad_enabled = True; ops._C1FFT.enabled = False
g2 = run('time fwd+wgrad')
ops.conv1d_split(False)
g3 = run('fp32 mfma conv1d')
ops.conv1d_split(True); ops._C1FFT.enabled = True
ops.deterministic(True)
g4 = run('deterministic')
ops.deterministic(False)
rel = lambda x, y: float((x - y).norm() / y.norm())
print('conv1d dW device variants vs default:', [rel(gx[16], g0[16]) for gx in (g1, g2, g3, g4)])
