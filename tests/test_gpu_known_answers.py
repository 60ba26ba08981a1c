"""End-to-end known answers: the posterior summaries the reference's tutorials record
(docs/tutorials/decomposition.ipynb:519-524, pelton.ipynb, dias.ipynb -- SURVEY.md §4).
They are stochastic (emcee version unpinned), so the comparison is statistical: our
posterior mean must sit within a fraction of a posterior standard deviation of the
recorded mean."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# PolynomialDecomposition(poly_deg=4, c_exp=1), 32 walkers x 1000 steps, discard 500
DEBYE_TUTORIAL = {
    'SIP-K389170': [0.989771, 0.014033, -0.002350, -0.004634, -0.000338, 0.000219],
    'SIP-K389172': [1.018388, 0.017680, -0.011578, -0.003509, 0.002061, 0.000526],
    'SIP-K389173': [1.011923, 0.003390, -0.000995, -0.000499, 0.000396, 0.000172],
    'SIP-K389174': [1.004851, 0.007547, -0.003206, -0.001871, 0.000548, 0.000242],
    'SIP-K389175': [0.997613, 0.006870, -0.003937, -0.001338, 0.000741, 0.000219],
    'SIP-K389176': [1.006941, 0.002426, -0.001084, -0.000048, 0.000564, 0.000159],
}


@pytest.mark.parametrize('name', sorted(DEBYE_TUTORIAL))
def test_debye_decomposition_tutorial_means(name):
    import bisip_amd
    np.random.seed(42)
    model = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()[name], nwalkers=32, poly_deg=4,
                                              c_exp=1, nsteps=3000)
    model.fit()                       # prior-uniform start, as in the tutorial
    chain = model.get_chain(discard=1500, flat=True)
    mean, std = model.get_param_mean(chain), model.get_param_std(chain)
    want = np.array(DEBYE_TUTORIAL[name])
    z = np.abs(mean - want) / std
    assert np.all(z < 0.35), (name, z)
    # the stored log-prob is the log-prob of the stored position
    lp = model.sampler.get_log_prob()[-1]
    assert np.allclose(lp, model.log_prob(model.get_chain()[-1]), rtol=1e-12, atol=1e-9)


def test_pelton_two_mode_tutorial():
    """docs/tutorials/pelton.ipynb:346-418 -- 2 modes on K389174, bounds log_tau1 in [-5,5],
    log_tau2 in [-15,-10], 64 walkers x 1000, discard 500."""
    import bisip_amd
    np.random.seed(42)
    m = bisip_amd.PeltonColeCole(bisip_amd.DataFiles()['SIP-K389174'], nwalkers=64, n_modes=2,
                                 nsteps=4000)
    m.params.update(log_tau1=[-5, 5], log_tau2=[-15, -10])
    m.fit()
    chain = m.get_chain(discard=2500, thin=5, flat=True)
    mean, std = m.get_param_mean(chain), m.get_param_std(chain)
    want = np.array([1.010, 0.140, 0.935, -1.574, -12.838, 0.451, 0.608])
    rec_std = np.array([0.002, 0.004, 0.055, 0.089, 0.148, 0.015, 0.015])
    assert np.all(np.abs(mean - want) < 1.0 * np.maximum(std, rec_std)), (mean, std)


def test_dias_tutorial():
    """docs/tutorials/dias.ipynb:362-410 -- K389172, bounds eta in [0,25], log_tau in [-15,-5]."""
    import bisip_amd
    np.random.seed(42)
    m = bisip_amd.Dias2000(bisip_amd.DataFiles()['SIP-K389172'], nwalkers=32, nsteps=6000)
    m.params.update(eta=[0, 25], log_tau=[-15, -5])
    m.fit()
    chain = m.get_chain(discard=3000, thin=5, flat=True)
    mean, std = m.get_param_mean(chain), m.get_param_std(chain)
    want = np.array([1.023, 0.688, -10.315, 7.767, 0.707])
    rec_std = np.array([0.003, 0.058, 0.162, 0.838, 0.068])
    assert np.all(np.abs(mean - want) < 1.5 * np.maximum(std, rec_std)), (mean, std)
