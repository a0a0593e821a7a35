"""Dataset plumbing -- hyperbo/bo_utils/data.py:103-443,720-775: PD1 jsonl -> Dict[key, SubDataset].

Host-only (pandas).  The PD1 files are not part of either repository (reference README.md:26-31); `pd1`
takes the same `data_files` mapping {(phase, 'matched'|'unmatched'): path}.  Randomness: NumPy Generators
(or integer seeds) replace the JAX PRNG keys, so draws differ from the reference's for equal seeds.
"""
import itertools
import logging

import numpy as np

from hyperbo_amd.basics import definitions as defs

SubDataset = defs.SubDataset

PD1 = {
    ('phase0', 'matched'): '../pd1/pd1_matched_phase0_results.jsonl',
    ('phase1', 'matched'): '../pd1/pd1_matched_phase1_results.jsonl',
    ('phase0', 'unmatched'): '../pd1/pd1_unmatched_phase0_results.jsonl',
    ('phase1', 'unmatched'): '../pd1/pd1_unmatched_phase1_results.jsonl',
}
PD1_HPARAMS = ['hps.lr_hparams.decay_steps_factor', 'hps.lr_hparams.initial_value', 'hps.lr_hparams.power',
               'hps.opt_hparams.momentum']


def _rng(key):
  return key if isinstance(key, np.random.Generator) else np.random.default_rng(0 if key is None else key)


def _seed(rng):
  return int(rng.integers(0, 2**31 - 1))


def sample_dataframe(key, df, p_remove=0.):
  """data.py:103-112: keep ceil((1 - p_remove) * len) random rows."""
  if p_remove < 0 or p_remove >= 1:
    raise ValueError(f'p_remove={p_remove} but p_remove must be <1 and >= 0.')
  if p_remove > 0:
    n_remain = int(np.ceil((1 - p_remove) * len(df)))
    df = df.sample(n=n_remain, replace=False, random_state=_seed(_rng(key)))
  return df


def get_aligned_dataset(trials, study_identifier, labels, key=None, p_remove=0., verbose=True):
  """data.py:115-173.  Per phase (`aligned_suffix`) pivot the matched trials to (hparams) x (study group).
  Besides the fully observed rows, rows that miss up to two of the groups with gaps are emitted as separate
  aligned sub-datasets keyed '<missing groups>;<suffix>', y columns = the remaining groups."""
  rng = _rng(key)
  out = {}
  trials = trials[trials['aligned']]
  for suffix in trials['aligned_suffix'].unique():
    phase = trials[trials['aligned_suffix'] == suffix]
    groups = list(phase[study_identifier].unique())
    table = phase.pivot(index=labels[:-1], columns=study_identifier, values=labels[-1])
    gappy = [g for g in table.columns if table[g].isna().values.any()]
    max_missing = min(3, len(gappy) + 1, len(groups) - 1)
    for r in range(max_missing):
      for missing in itertools.combinations(gappy, r):
        keep = [g for g in groups if g not in missing]
        if missing:
          rows = np.all([table[g].isnull() for g in missing], axis=0)
          sub = table.loc[rows, keep].dropna().reset_index()
        else:
          sub = table.dropna().reset_index()
        if sub.shape[0] == 0:
          continue
        sub = sample_dataframe(rng, sub, p_remove=p_remove)
        out[';'.join(list(missing) + [suffix])] = SubDataset(
            x=np.asarray(sub[labels[:-1]], dtype=np.float64), y=np.asarray(sub[keep], dtype=np.float64),
            aligned=';'.join(keep + [suffix]))
  msg = f'aligned dataset: { {k: (v.x.shape, v.y.shape) for k, v in out.items()} }'
  logging.info(msg)
  if verbose:
    print(msg)
  return out


def get_dataset(trials, study_identifier, labels, verbose=True):
  """data.py:176-199: one (n,d)/(n,1) SubDataset per study group."""
  dataset = {}
  for sg in trials[study_identifier].unique():
    rows = trials.loc[trials[study_identifier] == sg, labels]
    dataset[sg] = SubDataset(x=np.asarray(rows[labels[:-1]], dtype=np.float64),
                             y=np.asarray(rows[labels[-1:]], dtype=np.float64))
  msg = f'dataset before align: { {k: (v.x.shape, v.y.shape) for k, v in dataset.items()} }'
  logging.info(msg)
  if verbose:
    print(msg)
  return dataset


def sample_sub_dataset(key, trials, study_identifier, labels, p_observed=0., verbose=True, sub_dataset_key=None):
  """data.py:202-250: split off the test study; a fraction 1 - p_observed of it becomes the query pool."""
  rng = _rng(key)
  study_groups = trials[study_identifier].unique()
  if sub_dataset_key is None:
    sub_dataset_key = study_groups[int(rng.integers(len(study_groups)))]
  elif sub_dataset_key not in study_groups:
    raise ValueError(f'{sub_dataset_key} must be in dataframe.')
  queried = trials[trials[study_identifier] == sub_dataset_key].sample(
      frac=1. - p_observed, replace=False, random_state=_seed(rng))
  trials = trials.drop(queried.index)
  queried_sub_dataset = SubDataset(x=np.asarray(queried[labels[:-1]], dtype=np.float64),
                                   y=np.asarray(queried[labels[-1:]], dtype=np.float64))
  msg = (f'removed study={sub_dataset_key}  removed study shape: x-{queried_sub_dataset.x.shape}, '
         f'y-{queried_sub_dataset.y.shape}')
  logging.info(msg)
  if verbose:
    print(msg)
  return trials, sub_dataset_key, queried_sub_dataset


def process_dataframe(key, trials, study_identifier, labels, p_observed=0., maximize_metric=True, warp_func=None,
                      verbose=True, sub_dataset_key=None, num_remove=0, p_remove=0.):
  """data.py:253-353 -> (dataset, sub_dataset_key, queried_sub_dataset)."""
  rng = _rng(key)
  trials = trials[[study_identifier] + labels + ['aligned', 'aligned_suffix']].copy(deep=True).dropna()
  warp_func = dict(warp_func or {})
  if labels[-1] not in warp_func and not maximize_metric:
    warp_func[labels[-1]] = lambda v: -v
  for label, fn in warp_func.items():
    if label in labels:
      trials[label] = fn(trials[label])
  assert len(trials) == len(trials.dropna()), f'nan appeared after applying warp_func={warp_func}'
  trials, sub_dataset_key, queried_sub_dataset = sample_sub_dataset(
      rng, trials, study_identifier, labels, p_observed=p_observed, verbose=verbose, sub_dataset_key=sub_dataset_key)
  for _ in range(num_remove):
    # hold out further training tasks: prefer the study that shares the test task's dataset name
    removed_key = None
    parts = str(sub_dataset_key).split(',')
    if len(parts) > 1:
      for sg in trials[study_identifier].unique():
        if parts[1] in sg:
          removed_key = sg
    trials, _, _ = sample_sub_dataset(rng, trials, study_identifier, labels, p_observed=p_observed, verbose=verbose,
                                      sub_dataset_key=removed_key)
    if trials.empty:
      raise ValueError(f'All datapoints are removed. Is num_remove={num_remove} too large?')
  aligned = get_aligned_dataset(trials, study_identifier, labels, key=rng, p_remove=p_remove, verbose=verbose)
  trials = sample_dataframe(rng, trials, p_remove=p_remove)
  dataset = get_dataset(trials, study_identifier, labels, verbose=verbose)
  dataset.update(aligned)
  return dataset, sub_dataset_key, queried_sub_dataset


def pd1(key, p_observed, verbose=True, sub_dataset_key=None, input_warp=True, output_log_warp=True, num_remove=0,
        metric_name='best_valid/error_rate', p_remove=0., data_files=None):
  """data.py:356-443: PD1 (Nesterov) jsonl / pickle files -> training dataset + held-out test study."""
  import pandas as pd
  import pickle
  data_files = dict(PD1 if data_files is None else data_files)
  frames = []
  for (phase, kind), path in data_files.items():
    if 'pkl' in path:
      with open(path, 'rb') as f:
        frame = pickle.load(f)
    else:
      frame = pd.read_json(path, orient='records', lines=True, precise_float=True)
    frame = frame.copy()
    frame['aligned'] = (kind == 'matched')
    frame['aligned_suffix'] = phase
    frames.append(frame)
  trials = pd.concat(frames).reset_index(drop=True)
  labels = PD1_HPARAMS + [metric_name]
  warp_func = {}
  if input_warp:
    warp_func = {'hps.opt_hparams.momentum': lambda v: np.log(1 - v), 'hps.lr_hparams.initial_value': np.log}
  if output_log_warp:
    warp_func['best_valid/error_rate'] = lambda v: -np.log(v + 1e-10)
  return process_dataframe(key=key, trials=trials, study_identifier='study_group', labels=labels,
                           p_observed=p_observed, maximize_metric=False, warp_func=warp_func if input_warp else None,
                           verbose=verbose, sub_dataset_key=sub_dataset_key, num_remove=num_remove, p_remove=p_remove)


def _deduplicate(x, y, dataset_name, verbose=True):
  """data.py:446-457: one row per distinct x, the one with the highest y; rows come out in np.unique order."""
  x = np.asarray(x); y = np.asarray(y)
  order = np.argsort(-y.reshape(y.shape[0], -1)[:, 0], kind='stable')   # best reward first
  x, y = x[order], y[order]
  _, first = np.unique(x, axis=0, return_index=True)
  if verbose:
    print(f'Removed {x.shape[0] - len(first)} duplicated points from {dataset_name}')
  return x[first, :], y[first, :]


def _normalize_maf_dataset(maf_dataset, num_hparams, neg_error_to_accuracy):
  """data.py:460-486: affine map of every hyper-parameter to [0, 1] over ALL sub-datasets; optionally y -> 1 + y
  (negative error rate -> accuracy).  Modifies and returns `maf_dataset` ({name: {'X': ..., 'Y': ...}})."""
  lo = np.full(num_hparams, np.inf); hi = np.full(num_hparams, -np.inf)
  for sub in maf_dataset.values():
    lo = np.minimum(lo, np.min(sub['X'], axis=0))
    hi = np.maximum(hi, np.max(sub['X'], axis=0))
  for sub in maf_dataset.values():
    sub['X'] = (sub['X'] - lo) / (hi - lo)
    if neg_error_to_accuracy:
      sub['Y'] = 1 + sub['Y']
  return maf_dataset


def random(key, mean_func, cov_func, params, dim, n_observed, n_queries, n_func_historical=0,
           m_points_historical=0, warp_func=None):
  """data.py:720-775: historical functions + one queried function, all drawn from the GP prior
  (gp.sample_from_gp: Gram and Cholesky on the device)."""
  from hyperbo_amd.gp_utils import gp
  rng = _rng(key)
  dataset = {}
  for i in range(n_func_historical):
    vx = rng.uniform(size=(m_points_historical, dim))
    dataset[i] = SubDataset(x=vx, y=gp.sample_from_gp(rng, mean_func, cov_func, params, vx, warp_func=warp_func))
  vx = rng.uniform(size=(n_observed + n_queries, dim))
  vy = gp.sample_from_gp(rng, mean_func, cov_func, params, vx, warp_func=warp_func)
  dataset[n_func_historical] = SubDataset(x=vx[n_queries:], y=vy[n_queries:])
  return dataset, n_func_historical, SubDataset(x=vx[:n_queries], y=vy[:n_queries])
