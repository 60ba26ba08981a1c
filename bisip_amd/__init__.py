"""bisip_amd -- MI355X-native implementation of BISIP's ensemble-MCMC likelihood hot path.

Drop-in for the model-class surface of ``bisip`` (reference: src/bisip/__init__.py):

    from bisip_amd import PolynomialDecomposition, DataFiles
    model = PolynomialDecomposition(DataFiles()['SIP-K389175'], nwalkers=32, nsteps=1000)
    model.fit()
    chain = model.get_chain(discard=500, thin=2, flat=True)

The arithmetic (prior + forward model + Gaussian log-likelihood for every walker)
runs in hand-written gfx950 kernels behind the C ABI of ``include/bisip_hip.h``.
"""

from .batch import SpectraBatch
from .data import DataFiles
from .models import (ColeCole, Dias2000, Inversion, PeltonColeCole, PolynomialDecomposition,
                     Shin2015)
from .sampler import DeviceEnsembleSampler, EnsembleSampler
from .utils import load_data, load_data_batch

__all__ = ('Inversion', 'PolynomialDecomposition', 'PeltonColeCole', 'ColeCole', 'Dias2000',
           'Shin2015', 'DataFiles', 'SpectraBatch', 'EnsembleSampler', 'DeviceEnsembleSampler',
           'load_data', 'load_data_batch')
