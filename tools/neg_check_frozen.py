"""Negative control of tests/test_fullsize_gpu.py::test_replayed_critic_step_follows_the_generators_updates: with the refresh of the
frozen generator's kernel planes switched OFF the replayed critic step must DISAGREE with the eager one once the generator's weights
have moved (the test must be able to fail).    python tools/neg_check_frozen.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest


class Plugin(object):
    @pytest.fixture(autouse=True)
    def no_refresh(self, monkeypatch):
        from percivaltts_amd import ops
        monkeypatch.setattr(ops._C1FFT, 'refresh_planes', classmethod(lambda cls, items: 0))
        yield


if __name__ == '__main__':
    rc = pytest.main(['-x', '-q', '-m', 'gpu', os.path.join(os.path.dirname(__file__), '..', 'tests', 'test_fullsize_gpu.py'), '-k', 'follows_the_generators',
                      '-p', 'no:cacheprovider'], plugins=[Plugin()])
    print('pytest exit code', int(rc), '(1 = the test failed, as it must without the refresh)')
    sys.exit(0 if int(rc) == 1 else 1)
