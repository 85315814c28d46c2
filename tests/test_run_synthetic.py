"""End-to-end counterpart of the reference's tests/test_run.py:10-42 on the MI355X: the run.py-shaped script with the
same shrunk configuration (batch 2, hidden 4, 1 conv layer, <=10 epochs) over a synthetic corpus of [T,425] -> [T,163]
files, through OptimizerTTSWGAN.train (epoch loop, validation costs, checkpoints), then --continue and generate."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_run_training_continue_generate(tmp_path, monkeypatch):
    monkeypatch.setenv('PERCIVAL_CORPUS', str(tmp_path / 'corpus'))
    monkeypatch.chdir(tmp_path)
    import importlib
    import percivaltts_amd.run as run
    run = importlib.reload(run)

    print('Overwrite the configuration to run a smoke test')   # tests/test_run.py:24-35
    run.cfg.id_valid_start = 8
    run.cfg.id_valid_nb = 1
    run.cfg.id_test_nb = 1
    run.cfg.train_min_nbepochs = 1
    run.cfg.train_max_nbepochs = 4
    run.cfg.train_cancel_nodecepochs = 3
    run.cfg.train_nbepochs_scalewdata = False
    run.cfg.train_batch_size = 2
    run.cfg.arch_hiddenwidth = 4
    run.cfg.arch_gen_nbcnnlayers = 1
    run.cfg.train_batch_lengthmax = 60
    run.synthesize_corpus(nfiles=10, minlen=90, maxlen=140)

    run.training(cont=False)
    stem = 'model'
    for f in ('model.h5.arch.json', 'model.h5.weights.npz', 'model.h5.cfgextras.pkl', stem + '-last.h5.weights.npz',
              stem + '-trainingstate-last.h5.generator.optimizer.npz', stem + '-trainingstate-last.h5.critic.optimizer.npz',
              stem + '-trainingstate-last.h5.critic.weights.npz', stem + '-trainingstate-last.h5.model.cfgextras.pkl'):
        assert os.path.exists(f), f

    run.cfg.train_max_nbepochs = 6
    run.training(cont=True)            # resumes at epoch 5 from the saved state
    run.generate()
    outs = glob.glob('model-gen/*.cmp')
    assert len(outs) == 1
    y = np.fromfile(outs[0], dtype=np.float32).reshape(-1, 163)
    assert y.shape[0] >= 90 and np.isfinite(y).all()
    assert (y[:, 130:] > 0).all() and (y[:, 130:] < 1).all()     # sigmoid noise-mask head


def test_resumed_run_reproduces_the_uninterrupted_one(tmp_path, monkeypatch):
    """--continue (reference optimizertts.py:197-211): 4 epochs straight against 2 epochs + a resumed run of 2 more, in
    deterministic mode (ops.deterministic: fixed-order reductions).  Generator and critic weights, both Adam states, the
    BatchNorm moving statistics, the per-epoch costs and the generator-update counter must agree -- the training state
    carries the critic's weights, the device RNG and `generator_updates` beyond what the reference saves."""
    import importlib, pickle
    from percivaltts_amd import ops
    monkeypatch.setenv('PERCIVAL_CORPUS', str(tmp_path / 'corpus'))
    import percivaltts_amd.run as run
    ops.deterministic(True)
    try:
        states = {}
        for name, plan in (('straight', [(4, False)]), ('resumed', [(2, False), (4, True)])):
            wd = tmp_path / name
            wd.mkdir()
            monkeypatch.chdir(wd)
            run = importlib.reload(run)
            run.cfg.id_valid_start = 8; run.cfg.id_valid_nb = 1; run.cfg.id_test_nb = 1
            run.cfg.train_min_nbepochs = 1; run.cfg.train_cancel_nodecepochs = 10
            run.cfg.train_nbepochs_scalewdata = False
            run.cfg.train_batch_size = 2; run.cfg.arch_hiddenwidth = 8; run.cfg.arch_gen_nbcnnlayers = 2
            run.cfg.train_batch_lengthmax = 60
            if not os.path.exists(run.cfg.fileids):
                run.synthesize_corpus(nfiles=10, minlen=90, maxlen=140)
            for (nep, cont) in plan:
                np.random.seed(123); __import__('torch').manual_seed(123)
                if cont:
                    np.random.seed(999); __import__('torch').manual_seed(999)      # a resumed run must not depend on the fresh seeds
                run.cfg.train_max_nbepochs = nep
                run.training(cont=cont)
            st = 'model-trainingstate-last.h5'
            with open(st + '.model.cfgextras.pkl', 'rb') as f:
                _, extras, _ = pickle.load(f)
            states[name] = {k: dict(np.load(st + k)) for k in ('.model.weights.npz', '.critic.weights.npz', '.generator.optimizer.npz', '.critic.optimizer.npz')}
            states[name]['extras'] = extras
    finally:
        ops.deterministic(False)
    a, b = states['straight'], states['resumed']
    assert a['extras']['epoch'] == b['extras']['epoch'] == 4
    assert a['extras']['generator_updates'] == b['extras']['generator_updates'] > 0
    for k in ('model_training', 'critic_training', 'critic_validation', 'model_validation', 'model_rmse_validation'):
        np.testing.assert_allclose(b['extras']['costs'][k], a['extras']['costs'][k], rtol=1e-6, atol=1e-7, err_msg=k)
    for part in ('.model.weights.npz', '.critic.weights.npz', '.generator.optimizer.npz', '.critic.optimizer.npz'):
        assert sorted(a[part]) == sorted(b[part])
        for k in a[part]:
            np.testing.assert_allclose(b[part][k], a[part][k], rtol=1e-6, atol=1e-8, err_msg=part + ':' + k)


def test_training_run_with_and_without_the_generator_look_ahead(tmp_path, monkeypatch):
    """The training driver names the next batch to the optimiser (optimizertts.train_oneparamset -> hint_next_batch), which launches a
    generator step's forward one batch ahead (cfg.train_wgan_generator_lookahead).  Eleven batches per epoch, so that batch 9 is followed
    by a batch that trains the generator (reference optimizertts_wgan.py:225-228: every 10th batch at the start): two epochs with and
    without the look-ahead, in deterministic mode, must leave the same weights, optimiser states, moving statistics and costs."""
    import importlib, pickle
    from percivaltts_amd import ops, optimizertts_wgan
    monkeypatch.setenv('PERCIVAL_CORPUS', str(tmp_path / 'corpus'))
    import percivaltts_amd.run as run
    ahead_calls = []
    orig = optimizertts_wgan.OptimizerTTSWGAN._batch_steps

    launched = []

    def spy(self, X, Y, alpha, gen_too, graph_c, graph_g, nxt=None, nxt_critic=None):
        ahead_calls.append(nxt is not None)
        r = orig(self, X, Y, alpha, gen_too, graph_c, graph_g, nxt, nxt_critic)
        launched.append(getattr(self, '_ahead', None) is not None)       # a generator forward was launched for the next batch
        return r
    monkeypatch.setattr(optimizertts_wgan.OptimizerTTSWGAN, '_batch_steps', spy)
    ops.deterministic(True)
    try:
        states, counts = {}, {}
        for name, look in (('ahead', True), ('plain', False)):
            wd = tmp_path / name
            wd.mkdir()
            monkeypatch.chdir(wd)
            run = importlib.reload(run)
            run.cfg.id_valid_start = 22; run.cfg.id_valid_nb = 1; run.cfg.id_test_nb = 1
            run.cfg.train_min_nbepochs = 1; run.cfg.train_cancel_nodecepochs = 10
            run.cfg.train_nbepochs_scalewdata = False
            run.cfg.train_batch_size = 2; run.cfg.arch_hiddenwidth = 8; run.cfg.arch_gen_nbcnnlayers = 2
            run.cfg.train_batch_lengthmax = 60
            run.cfg.train_max_nbepochs = 2
            run.cfg.train_wgan_generator_lookahead = look
            run.cfg.train_wgan_hipgraph = False             # (batches this small are replayed as hipGraphs otherwise: no look-ahead then)
            if not os.path.exists(run.cfg.fileids):
                run.synthesize_corpus(nfiles=24, minlen=90, maxlen=140)
            np.random.seed(123); __import__('torch').manual_seed(123)
            del ahead_calls[:]; del launched[:]
            run.training(cont=False)
            counts[name] = (len(ahead_calls), sum(ahead_calls), sum(launched))
            st = 'model-trainingstate-last.h5'
            with open(st + '.model.cfgextras.pkl', 'rb') as f:
                _, extras, _ = pickle.load(f)
            states[name] = {k: dict(np.load(st + k)) for k in ('.model.weights.npz', '.critic.weights.npz', '.generator.optimizer.npz', '.critic.optimizer.npz')}
            states[name]['extras'] = extras
    finally:
        ops.deterministic(False)
    assert counts['ahead'][0] == counts['plain'][0] >= 22        # 11 batches an epoch
    assert counts['ahead'][1] >= 2 and counts['plain'][1] >= 2   # the driver names the batch behind batch 9 in both runs ...
    assert counts['ahead'][2] >= 2 and counts['plain'][2] == 0, counts
    a, b = states['plain'], states['ahead']                      # ... and only the optimiser's switch decides what is done with it
    assert a['extras']['generator_updates'] == b['extras']['generator_updates'] >= 4
    for k in ('model_training', 'critic_training', 'critic_validation', 'model_validation', 'model_rmse_validation'):
        np.testing.assert_allclose(b['extras']['costs'][k], a['extras']['costs'][k], rtol=1e-6, atol=1e-7, err_msg=k)
    for part in ('.model.weights.npz', '.critic.weights.npz', '.generator.optimizer.npz', '.critic.optimizer.npz'):
        assert sorted(a[part]) == sorted(b[part])
        for k in a[part]:
            np.testing.assert_allclose(b[part][k], a[part][k], rtol=1e-6, atol=1e-8, err_msg=part + ':' + k)


@pytest.mark.gpu
def test_batch_prefetcher_device_mode():
    """data.BatchPrefetcher on the GPU: pinned slots are reused only after their copy completed (more batches than
    slots, each with its own contents), tensors arrive on the device in order, shapes may change between batches."""
    import numpy as np
    import torch
    from percivaltts_amd import data

    rng = np.random.RandomState(0)
    host = [(rng.randn(3 + (i % 2), 50, 7).astype(np.float32), rng.randn(3 + (i % 2), 50, 2).astype(np.float32)) for i in range(9)]
    for stage in (False, True):
        pf = data.BatchPrefetcher(lambda i: host[i], len(host), device='cuda', depth=2, stage_pinned=stage)
        n = 0
        for i, (x, y) in enumerate(pf):
            assert x.is_cuda and y.is_cuda and x.dtype == torch.float32
            # consume on the current stream (the iterator made it wait for the copy)
            assert torch.equal(x.cpu(), torch.from_numpy(host[i][0])) and torch.equal(y.cpu(), torch.from_numpy(host[i][1]))
            n += 1
        assert n == len(host)
        pf.close()
