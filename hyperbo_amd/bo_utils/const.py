"""Registries -- hyperbo/bo_utils/const.py:22-81: the closed set the native dispatcher covers."""
from hyperbo_amd.bo_utils import acfun
from hyperbo_amd.bo_utils import data
from hyperbo_amd.gp_utils import kernel
from hyperbo_amd.gp_utils import mean

MEAN = {'constant': mean.constant, 'linear': mean.linear, 'linear_mlp': mean.linear_mlp, 'zero': mean.zero}
KERNEL = {
    'squared_exponential': kernel.squared_exponential,
    'matern32': kernel.matern32,
    'matern52': kernel.matern52,
    'dot_product': kernel.dot_product,
    'dot_product_mlp': kernel.dot_product_mlp,
    'squared_exponential_mlp': kernel.squared_exponential_mlp,
    'matern32_mlp': kernel.matern32_mlp,
    'matern52_mlp': kernel.matern52_mlp,
}
ACFUN = {
    'expected_improvement': acfun.expected_improvement,
    'probability_of_improvement': acfun.probability_of_improvement,
    'ucb3': acfun.ucb3,
    'random_search': acfun.random_search,
    'ucb2': acfun.ucb2,
    'ucb': acfun.ucb,
}
ACFUN_SUB = {
    'expected_improvement': acfun.expected_improvement_sub,
    'probability_of_improvement': acfun.probability_of_improvement_sub,
    'ucb': acfun.ucb_sub,
}
EPS = 1e-6

HYPERBO_DATASETS = {'pd1': data.pd1, 'random': data.random}   # const.py:54-59
INPUT_SAMPLERS = {}                                             # const.py:61 (empty in the reference too)

# Method names of the reference's offline experiment manager (const.py:63-81).  The experiment manager itself is out of scope
# (SURVEY section 8); the names stay importable for user code written against hyperbo.bo_utils.const.  This package only reads
# USE_HGP (which method draws hyper-parameter samples: an HGP instead of a GP, bayesopt.py).
RAND, STBO, MTBO, STBOV = 'rand', 'stbo', 'mtbo', 'gp'
HBO, HBO_SS, HBO_NLL, HBO_NLLKL, HBO_NLLEUC = 'hyperbo', 'hyperbo_ss', 'hyperbo_nll', 'hyperbo_nllkl', 'hyperbo_nlleuc'
CONTEXTUAL_METHODS = ['rfgp', 'mimo', STBOV]
HBO_METHODS = [HBO_SS, HBO_NLL, HBO_NLLKL, HBO_NLLEUC]
OFFLINE_METHODS = [RAND, STBO, MTBO, HBO, HBO_SS] + CONTEXTUAL_METHODS
ONLINE_METHODS = [STBO, MTBO] + HBO_METHODS
USE_HGP = [HBO_SS]
ST_METHODS = [STBO, STBOV]
