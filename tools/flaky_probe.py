import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import io, contextlib
import torch
import test_model_gpu as tm
with contextlib.redirect_stdout(io.StringIO()):
    cfg, voc, mod, crit, a, gw, cw, X, Y, al = tm.build('default')
    from percivaltts_amd import optimizertts_wgan
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
Xd, Yd = tm.f32(X), tm.f32(Y)
ref = None
for it in range(12):
    opt.gen_opti.zero_grad()
    for p in opt.critic_opti.flat.params: p.requires_grad_(False)
    l, _ = opt.generator_loss(Xd, Yd, training=True)
    l.backward()
    for p in opt.critic_opti.flat.params: p.requires_grad_(True)
    g = opt.gen_opti.flat.grad.clone()
    if ref is None:
        ref = g
        continue
    d = (g - ref).abs()
    bad = (d > 1e-6 * ref.abs().max()).nonzero().flatten()
    if len(bad):
        off = 0; hits = []
        for i, p in enumerate(opt.gen_opti.flat.params):
            n = p.numel()
            sel = bad[(bad >= off) & (bad < off + n)] - off
            if len(sel):
                hits.append((i, tuple(p.shape), len(sel)))
            off += n
        print('iter', it, 'loss', float(l), 'affected params:', hits[-6:], '... total', len(hits))
    else:
        print('iter', it, 'identical to first (max diff', float(d.max()), ')')
