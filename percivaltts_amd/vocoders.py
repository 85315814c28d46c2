"""Vocoder descriptors: only what the acoustic-model hot path reads.

The reference's vocoders.py:33-352 also does signal analysis/synthesis through the `pulsemodel`
submodule (absent from the reference checkout) and pyworld; that DSP is out of scope here (SURVEY.md
section 2, row 9).  The networks, the critic and the WGAN losses only need the feature-size accessors
(vocoders.py:95-109,130-131,176-179,228-232), `fs`, `shift`, `mlpg_wins`, and the class identity that
network_final switches on (networktts.py:195,212).
"""


class Vocoder(object):
    def __init__(self, name, fs, shift, mlpg_wins=None):
        self._name, self.fs, self.shift, self.mlpg_wins = name, fs, shift, mlpg_wins

    def __str__(self):
        return '{} (fs={}, shift={})'.format(self.name(), self.fs, self.shift)

    def name(self):
        return self._name

    def featuressizeraw(self):
        raise ValueError('This member function has to be re-implemented in the sub-classes')

    def featuressize(self):
        n = self.featuressizeraw()
        return n * (len(self.mlpg_wins) + 1) if self.mlpg_wins is not None else n

    def f0size(self): return -1
    def specsize(self): return -1
    def noisesize(self): return -1
    def vuvsize(self): return -1

    def _out_of_scope(self, *a, **k):
        raise NotImplementedError('waveform analysis/synthesis is outside the MI355X hot-path build '
                                  '(needs the pulsemodel/pyworld DSP of the reference)')
    analysisf = analysisfid = synthesis = _out_of_scope


class VocoderF0Spec(Vocoder):
    def __init__(self, name, fs, shift, spec_size, spec_type='fwbnd', dftlen=4096, mlpg_wins=None):
        Vocoder.__init__(self, name, fs, shift, mlpg_wins=mlpg_wins)
        self.spec_size, self.spec_type, self.dftlen = spec_size, spec_type, dftlen

    def f0size(self): return 1
    def specsize(self): return self.spec_size


class VocoderPML(VocoderF0Spec):
    """f0 | spec | noise mask"""
    def __init__(self, fs, shift, spec_size, nm_size, dftlen=4096, mlpg_wins=None):
        VocoderF0Spec.__init__(self, 'PML', fs, shift, spec_size, 'fwbnd', dftlen, mlpg_wins=mlpg_wins)
        self.nm_size = nm_size

    def featuressizeraw(self): return 1 + self.spec_size + self.nm_size
    def noisesize(self): return self.nm_size
    def vuvsize(self): return 0


class VocoderWORLD(VocoderF0Spec):
    """f0 | spec | aperiodicity | vuv"""
    def __init__(self, fs, shift, spec_size, aper_size, dftlen=4096, mlpg_wins=None):
        VocoderF0Spec.__init__(self, 'WORLD', fs, shift, spec_size, 'fwbnd', dftlen, mlpg_wins=mlpg_wins)
        self.aper_size = aper_size

    def featuressizeraw(self): return 1 + self.spec_size + self.aper_size + 1
    def noisesize(self): return self.aper_size
    def vuvsize(self): return 1
