"""CPU tests of the host-side mirror of the reference API (no kernels are launched)."""
import os

import numpy as np
import pytest
import torch

import percivaltts_amd
from percivaltts_amd import vocoders, modeltts_common, networks_critic, data, optimizertts, optimizertts_wgan
from oracle import percival_oracle as O


def small_cfg(**kw):
    cfg = percivaltts_amd.configuration()
    cfg.arch_hiddenwidth = 4
    cfg.train_batch_size = 2
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_configuration_bag():
    a, b = percivaltts_amd.configuration(), percivaltts_amd.configuration()
    a.x, b.x = 1, 1
    assert a == b
    b.y = 2
    assert a != b
    a.merge(b)
    assert a == b and a.y == 2
    a.train_batch_size, a.id_valid_start = 5, 1032
    assert a.id_train_nb() == 1030            # percivaltts.py:68-70
    assert percivaltts_amd.is_int('12') and percivaltts_amd.is_int('-3') and percivaltts_amd.is_int('4.0')
    assert not percivaltts_amd.is_int('4.5') and not percivaltts_amd.is_int('abc')
    assert percivaltts_amd.proc_memresident() != -1           # tests/test_smoke.py:47


def test_vocoder_sizes():
    v = vocoders.VocoderPML(16000, 0.005, 129, 33)
    assert (v.featuressize(), v.specsize(), v.noisesize(), v.vuvsize(), v.f0size()) == (163, 129, 33, 0, 1)   # run.py:89
    w = vocoders.VocoderWORLD(16000, 0.005, 65, 17, mlpg_wins=[[-0.5, 0.0, 0.5], [1.0, -2.0, 1.0]])
    assert w.featuressizeraw() == 84 and w.featuressize() == 252     # tests/test_smoke_tensorflowkeras.py:165,174
    with pytest.raises(NotImplementedError):
        v.synthesis(None)


def test_count_params_known_answers():
    voc = vocoders.VocoderPML(16000, 0.005, 65, 17)
    m = modeltts_common.Generic(425, voc, layertypes=['FC', 'FC', 'FC'], cfgarch=small_cfg())
    assert m.count_params() == 2195          # /root/reference/tests/test_smoke_tensorflowkeras.py:53
    cfg = small_cfg(arch_hiddenwidth=2, arch_ctx_nbcnnlayers=2, arch_ctx_winlen=3, arch_gen_nbcnnlayers=2,
                    arch_gen_nbfilters=2, arch_gen_winlen=3, arch_spec_freqlen=3)
    g = modeltts_common.DCNNF0SpecNoiseFeatures(425, voc, cfg)
    c = networks_critic.Critic(voc, 425, cfg)
    assert g.count_params() == 3034 and c.model.count_params() == 2917
    # weight order and shapes are the oracle's (creation order of the reference's layers)
    a = O.Arch(425, 65, 17, 2, 2, 3, 2, 2, 3, 3)
    assert [tuple(t.shape) for _, t in g.kerasmodel.weights()] == [tuple(s) for s in O.generator_weight_shapes(a)]
    assert [tuple(t.shape) for _, t in c.model.weights()] == [tuple(s) for s in O.critic_weight_shapes(a)]


def test_generic_layer_grammar_and_errors():
    voc = vocoders.VocoderPML(16000, 0.005, 65, 17)
    m = modeltts_common.Generic(425, voc, layertypes=[['CNN1D', 6, 5], 'FC', ['FC', 3], 'BLSTM', 'GRU', 'BGRU', ['RND', 2]], cfgarch=small_cfg())
    assert m.count_params() > 0
    with pytest.raises(ValueError):
        modeltts_common.Generic(425, voc, layertypes=['NOPE'], cfgarch=small_cfg())
    wv = vocoders.VocoderWORLD(16000, 0.005, 65, 17)
    mw = modeltts_common.Generic(425, wv, layertypes=['FC'], cfgarch=small_cfg())
    assert mw.kerasmodel.outputs[0].shape == (84,)


def test_save_load_roundtrip(tmp_path):
    voc = vocoders.VocoderPML(16000, 0.005, 65, 17)
    cfg = small_cfg(dummyattribute=-1)
    m = modeltts_common.Generic(425, voc, layertypes=['FC', 'FC'], cfgarch=cfg)
    f = str(tmp_path / 'smokymodelparams.pkl')
    m.save(f, cfg=cfg, extras={'cost_val': 67.43})
    for ext in ('.arch.json', '.weights.npz', '.cfgextras.pkl'):
        assert os.path.exists(f + ext)
    before = m.kerasmodel.get_weights()
    with torch.no_grad():
        for p in m.kerasmodel.parameters():
            p.add_(1.0)
    cfg_loaded, extras_loaded = m.load(f)
    assert cfg_loaded == cfg and extras_loaded == {'cost_val': 67.43}      # tests/test_smoke_tensorflowkeras.py:77-79
    for a, b in zip(before, m.kerasmodel.get_weights()):
        np.testing.assert_array_equal(a, b)


def test_randomize_hyper():
    cfg = small_cfg()
    cfg.train_hypers = [('train_learningrate_log10', -6.0, -2.0), ('train_adam_beta1', 0.8, 1.0), ('train_batch_size', 1, 4)]
    c1, s1 = optimizertts.OptimizerTTS.randomize_hyper(cfg)
    c2, s2 = optimizertts.OptimizerTTS.randomize_hyper(cfg)
    assert c1 != c2 and s1 != s2 and isinstance(c1.train_batch_size, (int, np.integer))     # :103
    cfg.train_hypers = []
    c3, s3 = optimizertts.OptimizerTTS.randomize_hyper(cfg)
    assert s3 == ''


def test_wgan_defaults_weights_and_schedule():
    voc = vocoders.VocoderPML(16000, 0.005, 65, 17)
    cfg = small_cfg(arch_hiddenwidth=2, arch_ctx_nbcnnlayers=1, arch_ctx_winlen=3, arch_gen_nbcnnlayers=2,
                    arch_gen_nbfilters=2, arch_gen_winlen=3, arch_spec_freqlen=3)
    g = modeltts_common.DCNNF0SpecNoiseFeatures(425, voc, cfg)
    c = networks_critic.Critic(voc, 425, cfg)
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, g, errtype='WLSWGAN', critic=c)
    assert opt.cfg.train_wgan_pg_lambda == 10 and opt.cfg.train_wgan_LScoef == 0.25
    assert opt.cfg.train_wgan_critic_learningrate_log10 == -4 and opt.cfg.train_wgan_gen_adam_beta2 == 0.9
    opt.cfg.train_wgan_critic_LSWGANtransidx = 30.0
    w_ls, ww = opt._wls_weights()
    w_ref, ww_ref = O.wls_weights(65, 17, 0, 0.25, 30.0)
    np.testing.assert_allclose(w_ls, w_ref, rtol=1e-12)
    assert abs(ww - ww_ref) < 1e-12
    # the build's own frequency warping: increasing in the cut-off, inside the band range
    i1, i2 = optimizertts_wgan.freq2fwspecidx(2000, 16000, 65), optimizertts_wgan.freq2fwspecidx(4000, 16000, 65)
    assert 0 < i1 < i2 < 65

    # schedule of train_on_batch (optimizertts_wgan.py:225-231) with the device steps stubbed out
    calls = []
    opt.world = 1
    opt.critic_step = lambda X, Y, alpha=None: calls.append('c') or torch.zeros(())
    opt.generator_step = lambda X, Y: calls.append('g') or torch.zeros(())
    opt.cfg.train_wgan_hipgraph = False
    for gen_updates, expect in ((0, 10), (24, 10), (25, 5), (499, 5), (500, 10), (501, 5)):
        opt.generator_updates = gen_updates
        fired = []
        for batchid in range(20):
            opt.generator_updates = gen_updates
            _, lg = opt.device_step(batchid, None, None)
            fired.append(lg is not None)
        assert [i for i, f in enumerate(fired) if f] == list(range(0, 20, expect))


def test_data_batching(tmp_path):
    rng = np.random.RandomState(0)
    fids = ['a', 'b', 'c']
    lens = [37, 52, 44]
    for sub, dim in (('lab', 7), ('cmp', 5), ('w', 1)):
        os.makedirs(str(tmp_path / sub))
        for fid, n in zip(fids, lens):
            arr = rng.rand(n + (1 if sub == 'lab' else 0), dim).astype(np.float32)
            if sub == 'w':
                arr[:] = 1.0
                arr[:3] = 0.0
            arr.tofile(str(tmp_path / sub / (fid + '.' + sub)))
    indir = str(tmp_path / 'lab' / '*.lab') + ':(-1,7)'
    outdir = str(tmp_path / 'cmp' / '*.cmp') + ':(-1,5)'
    wdir = str(tmp_path / 'w' / '*.w') + ':(-1,1)'
    assert data.getpathandshape(indir) == (str(tmp_path / 'lab' / '*.lab'), (-1, 7)) and data.getlastdim(outdir) == 5
    X = data.load(indir, fids)
    Y = data.load(outdir, fids)
    assert X[0].shape == (38, 7) and Y[0].shape == (37, 5)
    X, Y = data.croplen([X, Y])
    assert X[0].shape[0] == Y[0].shape[0] == 37
    Xb, Yb, Wb = data.load_inoutset(indir, outdir, wdir, fids, length=None, lengthmax=20, maskpadtype='randshift', cropmode='begend')
    assert Xb.shape == (3, 20, 7) and Yb.shape == (3, 20, 5) and Wb.shape == (3, 20, 1) and Xb.dtype == np.float32
    assert abs(data.cost_0pred_rmse(Y) - np.sqrt(np.mean(np.concatenate([y.ravel() for y in Y]) ** 2))) < 1e-6

    class Stub:                                    # tests/test_smoke.py:123-127
        def predict(self, x):
            return np.zeros((1, x.shape[1], 5), dtype=np.float32)
    assert abs(data.cost_model_prediction_rmse(Stub(), [X], Y) - data.cost_0pred_rmse(Y)) < 1e-6
    assert data.prediction_rms(Stub(), [X]) == 0.0


def test_loader_windows_are_the_files_frames_and_rank_shards_agree(tmp_path):
    """load_inoutset (reference data.py:297-322 -> croplen :143-171, croplen_weight :173-231, batching :234-284): every returned
    window must be the file's frames -- after the common-length crop and the silence crop of the time weights -- at the drawn
    shift, for inputs, outputs and weights alike.  Then the data-parallel form: with the per-sample uniforms of the shifts
    drawn for the whole batch (`rand`), a rank that loads only ITS shard of the file list gets exactly the rows the whole-batch
    load gives (optimizertts.train_oneparamset shards the file ids before reading)."""
    rng = np.random.RandomState(3)
    fids = ['u{}'.format(i) for i in range(6)]
    lens = [61, 75, 58, 90, 66, 71]
    files = {}
    for sub, dim in (('lab', 7), ('cmp', 5), ('w', 1)):
        os.makedirs(str(tmp_path / sub))
        for k, (fid, n) in enumerate(zip(fids, lens)):
            arr = rng.rand(n + (2 if sub == 'lab' else 0), dim).astype(np.float32)      # labels two frames longer: croplen cuts them
            if sub == 'w':
                arr[:] = 1.0
                arr[:4 + k] = 0.0                       # leading silence of 4 + k frames
                arr[n - 3 - k:] = 0.0                   # trailing silence
                arr[20:23] = 0.2                        # a short pause inside: 'begend' keeps it
            files[(sub, fid)] = arr
            arr.tofile(str(tmp_path / sub / (fid + '.' + sub)))
    indir = str(tmp_path / 'lab' / '*.lab') + ':(-1,7)'
    outdir = str(tmp_path / 'cmp' / '*.cmp') + ':(-1,5)'
    wdir = str(tmp_path / 'w' / '*.w') + ':(-1,1)'
    L = 24

    def expected_rows(fid, k, shift):
        n = lens[k]
        w = files[('w', fid)][:n, 0]
        on = np.where(w > 0.5)[0]
        sel = slice(int(on.min()), int(on.max()))            # croplen_weight 'begend' (the reference's slice drops the last kept frame)
        return (files[('lab', fid)][:n][sel][shift:shift + L], files[('cmp', fid)][:n][sel][shift:shift + L],
                files[('w', fid)][:n][sel][shift:shift + L]), (sel.stop - sel.start)

    # ---- np.random.randint shifts (one process): replay the draws
    np.random.seed(11)
    Xb, Yb, Wb = data.load_inoutset(indir, outdir, wdir, fids, length=None, lengthmax=L, maskpadtype='randshift', cropmode='begend')
    assert Xb.shape == (6, L, 7) and Yb.shape == (6, L, 5) and Wb.shape == (6, L, 1)
    np.random.seed(11)
    for k, fid in enumerate(fids):
        _, kept = expected_rows(fid, k, 0)
        shift = np.random.randint(0, kept - L + 1)
        (ex, ey, ew), _ = expected_rows(fid, k, shift)
        assert np.array_equal(Xb[k], ex) and np.array_equal(Yb[k], ey) and np.array_equal(Wb[k], ew), (fid, shift)
        assert kept == lens[k] - (4 + k) - (3 + k) - 1       # silence cropped at both ends
    # ---- uniforms drawn for the whole batch: windows at floor(u * number of shifts); a rank's shard == the batch's rows
    u = np.random.RandomState(5).random_sample(6)
    Xa, Ya, Wa = data.load_inoutset(indir, outdir, wdir, fids, length=None, lengthmax=L, maskpadtype='randshift', cropmode='begend', rand=u)
    for k, fid in enumerate(fids):
        _, kept = expected_rows(fid, k, 0)
        (ex, ey, ew), _ = expected_rows(fid, k, int(u[k] * (kept - L + 1)))
        assert np.array_equal(Xa[k], ex) and np.array_equal(Ya[k], ey) and np.array_equal(Wa[k], ew)
    from percivaltts_amd import parallel
    for world in (2, 3):
        for rank in range(world):
            lo, hi = parallel.shard_batch(6, world, rank)
            Xr, Yr, Wr = data.load_inoutset(indir, outdir, wdir, fids[lo:hi], length=None, lengthmax=L, maskpadtype='randshift',
                                            cropmode='begend', rand=u[lo:hi])
            assert np.array_equal(Xr, Xa[lo:hi]) and np.array_equal(Yr, Ya[lo:hi]) and np.array_equal(Wr, Wa[lo:hi]), (world, rank)
    # ---- a sample SHORTER than lengthmax in another rank's shard (train_batch_length None, the default): the window length is the
    # global batch's, found before sharding (data.batch_window_length: file sizes + the one-column weight files only)
    Lbig = 60                                                # kept lengths are 53, 65, 46, 76, 50, 53: min 46 < 60
    kept_all = [expected_rows(fid, k, 0)[1] for k, fid in enumerate(fids)]
    for padtype, want_T in (('randshift', min(kept_all)), ('padright', Lbig)):
        Tg = data.batch_window_length(indir, outdir, wdir, fids, length=None, lengthmax=Lbig, maskpadtype=padtype, cropmode='begend')
        assert Tg == want_T
        Xg, Yg, Wg = data.load_inoutset(indir, outdir, wdir, fids, length=None, lengthmax=Lbig, maskpadtype=padtype, cropmode='begend', rand=u)
        assert Xg.shape[1] == Tg
        for world in (2, 3):
            for rank in range(world):
                lo, hi = parallel.shard_batch(6, world, rank)
                Xr, Yr, Wr = data.load_inoutset(indir, outdir, wdir, fids[lo:hi], length=Tg, lengthmax=Lbig, maskpadtype=padtype,
                                                cropmode='begend', rand=u[lo:hi])
                assert np.array_equal(Xr, Xg[lo:hi]) and np.array_equal(Yr, Yg[lo:hi]) and np.array_equal(Wr, Wg[lo:hi]), (padtype, world, rank)
        # (without the global length the shard of samples 3..5 would have windowed to its own minimum, 50, not 46)
    assert min(kept_all[3:]) != min(kept_all)
    assert data.batch_window_length(indir, outdir, wdir, fids, length=30, lengthmax=Lbig) == 30
    for cm in ('all', 'begendbigger'):
        Tc = data.batch_window_length(indir, outdir, wdir, fids, length=None, lengthmax=None, maskpadtype='randshift', cropmode=cm)
        assert Tc == data.load_inoutset(indir, outdir, wdir, fids, length=None, lengthmax=None, maskpadtype='randshift', cropmode=cm, rand=u)[0].shape[1]
    # ---- cropmode 'all' drops the pause inside as well
    Xc, Yc, Wc = data.load_inoutset(indir, outdir, wdir, fids[:1], length=None, lengthmax=L, maskpadtype='randshift', cropmode='all', rand=np.zeros(1))
    w0 = files[('w', 'u0')][:lens[0], 0]
    assert np.array_equal(Yc[0], files[('cmp', 'u0')][:lens[0]][np.where(w0 > 0.5)[0]][:L]) and float(Wc.min()) == 1.0


def test_batch_prefetcher_host_mode_order_and_errors():
    """data.BatchPrefetcher without a GPU: same iterator (order, contents, length), loader failures reach the consumer."""
    import numpy as np
    import pytest
    from percivaltts_amd import data

    def make(i):
        return np.full((2, 3), float(i), dtype=np.float32), np.arange(4, dtype=np.float64) + i

    got = list(data.BatchPrefetcher(make, 5, device=None, depth=2))
    assert len(got) == 5
    for i, (x, y) in enumerate(got):
        assert x.dtype.is_floating_point and tuple(x.shape) == (2, 3) and float(x[0, 0]) == float(i)
        assert np.allclose(y.numpy(), np.arange(4) + i)

    def bad(i):
        if i == 2:
            raise IOError('missing file')
        return (np.zeros(3, dtype=np.float32),)

    it = data.BatchPrefetcher(bad, 4, device=None)
    next(it); next(it)
    with pytest.raises(IOError):
        next(it)
    it.close()


def test_training_loop_names_the_next_batch():
    """optimizertts._with_next: the batch loop of train_oneparamset sees (index, batch, next batch or None) -- the very objects the
    prefetcher yields, so that OptimizerTTSWGAN.hint_next_batch can recognise the batch when it comes (device_step's look-ahead)."""
    from percivaltts_amd import data
    from percivaltts_amd.optimizertts import _with_next

    assert list(_with_next([])) == []
    assert list(_with_next(['a'])) == [(0, 'a', None)]
    assert list(_with_next('abc')) == [(0, 'a', 'b'), (1, 'b', 'c'), (2, 'c', None)]
    items = list(_with_next(data.BatchPrefetcher(lambda i: (np.full(2, float(i), dtype=np.float32),), 4, device=None, depth=2)))
    assert [i for i, _, _ in items] == [0, 1, 2, 3] and items[-1][2] is None
    for (i, cur, nx), (_, cur1, _) in zip(items[:-1], items[1:]):
        assert nx is cur1 and float(cur[0][0]) == float(i)          # identity, not equality: the optimiser compares with `is`
    # the base optimiser ignores the hint; the WGAN optimiser keeps it for the next train_on_batch only
    assert optimizertts.OptimizerTTS.hint_next_batch(None, 1, 2) is None
    class W(optimizertts_wgan.OptimizerTTSWGAN):
        def __init__(self): pass
    w = W()
    w.hint_next_batch('x', 'y'); assert w._next_batch == ('x', 'y')
    w.hint_next_batch(None, None); assert w._next_batch is None
