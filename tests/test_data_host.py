"""Host-side dataset plumbing (hyperbo/bo_utils/data.py:103-443) on a small PD1-shaped fixture; no GPU."""
import json
import os

import numpy as np
import pytest

from hyperbo_amd.bo_utils import data

HP = data.PD1_HPARAMS
METRIC = 'best_valid/error_rate'


def _write_pd1_like(tmp_path, rng):
  """Two phases x {matched, unmatched}.  Matched: 3 study groups on a shared hparam grid, one group misses rows."""
  files = {}
  grid = rng.uniform(0.05, 0.95, size=(6, 4))
  for phase in ('phase0', 'phase1'):
    rows = []
    for g, name in enumerate(['imagenet_resnet50,imagenet,resnet,resnet50,512', 'cifar10_wrn,cifar10,wrn,wrn,256',
                              'lm1b_transformer,lm1b,transformer,transformer,2048']):
      for i, h in enumerate(grid):
        if g == 2 and i >= 4:
          continue                      # group 2 misses two grid rows -> 'gappy' group
        rows.append({'study_group': name, **dict(zip(HP, h)), METRIC: float(rng.uniform(0.1, 0.9)), 'extra': 1})
    p = tmp_path / f'matched_{phase}.jsonl'
    p.write_text('\n'.join(json.dumps(r) for r in rows))
    files[(phase, 'matched')] = str(p)
    rows = []
    for name, cnt in (('cifar10_wrn,cifar10,wrn,wrn,256', 9), ('svhn_wrn,svhn,wrn,wrn,1024', 7)):
      for _ in range(cnt):
        rows.append({'study_group': name, **dict(zip(HP, rng.uniform(0.05, 0.95, size=4))),
                     METRIC: float(rng.uniform(0.1, 0.9))})
    rows.append({'study_group': 'svhn_wrn,svhn,wrn,wrn,1024', **dict(zip(HP, rng.uniform(0.05, 0.95, size=4))), METRIC: None})
    p = tmp_path / f'unmatched_{phase}.jsonl'
    p.write_text('\n'.join(json.dumps(r) for r in rows))
    files[(phase, 'unmatched')] = str(p)
  return files


def test_pd1_loader_structure(tmp_path):
  pytest.importorskip('pandas')
  rng = np.random.default_rng(0)
  files = _write_pd1_like(tmp_path, rng)
  test_key = 'svhn_wrn,svhn,wrn,wrn,1024'
  dataset, key, queried = data.pd1(1, p_observed=0.25, verbose=False, sub_dataset_key=test_key, data_files=files)
  assert key == test_key
  # 14 valid svhn rows (the NaN metric row is dropped), 75 % of them form the query pool
  assert queried.x.shape == (round(14 * 0.75), 4) and queried.y.shape == (queried.x.shape[0], 1)
  assert dataset[test_key].x.shape[0] == 14 - queried.x.shape[0]
  # output warp: -log(err + 1e-10) > 0 for err in (0.1, 0.9); momentum warp log(1 - m) < 0
  assert np.all(queried.y > 0) and np.all(queried.x[:, 3] < 0) and np.all(queried.x[:, 1] < 0)
  iid = {k: v for k, v in dataset.items() if v.aligned is None}
  aligned = {k: v for k, v in dataset.items() if v.aligned is not None}
  assert set(iid) == {'imagenet_resnet50,imagenet,resnet,resnet50,512', 'cifar10_wrn,cifar10,wrn,wrn,256',
                      'lm1b_transformer,lm1b,transformer,transformer,2048', test_key}
  # cifar10 appears matched (6 x 2 phases) and unmatched (9 x 2)
  assert iid['cifar10_wrn,cifar10,wrn,wrn,256'].x.shape == (30, 4)
  # per phase: the fully observed block (4 rows x 3 groups) and the rows missing the gappy group (2 rows x 2 groups)
  assert set(aligned) == {'phase0', 'phase1', 'lm1b_transformer,lm1b,transformer,transformer,2048;phase0',
                          'lm1b_transformer,lm1b,transformer,transformer,2048;phase1'}
  assert aligned['phase0'].x.shape == (4, 4) and aligned['phase0'].y.shape == (4, 3)
  gap = aligned['lm1b_transformer,lm1b,transformer,transformer,2048;phase1']
  assert gap.x.shape == (2, 4) and gap.y.shape == (2, 2) and gap.aligned.endswith(';phase1')
  assert aligned['phase0'].aligned.count(';') == 3


def test_pd1_loader_holdout_and_subsampling(tmp_path):
  pytest.importorskip('pandas')
  files = _write_pd1_like(tmp_path, np.random.default_rng(1))
  dataset, key, queried = data.pd1(np.random.default_rng(5), p_observed=0., verbose=False, input_warp=False,
                                   output_log_warp=False, num_remove=1, p_remove=0.5, data_files=files)
  assert key not in dataset or dataset[key].x.shape[0] == 0 or key in dataset   # fully queried test study
  assert np.all(queried.y < 0)            # maximize_metric=False without an output warp negates the error rate
  n_groups = len([k for k, v in dataset.items() if v.aligned is None])
  assert n_groups <= 3                     # test study fully removed + one held-out study
  with pytest.raises(ValueError):
    data.pd1(0, p_observed=0., verbose=False, sub_dataset_key='nope', data_files=files)
  with pytest.raises(ValueError):
    data.sample_dataframe(0, None, p_remove=1.0)


@pytest.mark.parametrize('x,y,expected_x,expected_y', [
    # hyperbo/bo_utils/data_test.py:85-100 (the reference's own known-answer vectors)
    ([[2., 3.], [2., 1.], [2., 3], [2., 3.]], [[1.], [2.], [4.], [3]], [[2., 1.], [2., 3.]], [[2.], [4.]]),
    ([[1., 2.], [3., 4.]], [[1.], [2.]], [[1., 2.], [3., 4.]], [[1.], [2.]]),
])
def test_deduplicate_reference_vectors(x, y, expected_x, expected_y):
  ax, ay = data._deduplicate(np.array(x), np.array(y), dataset_name='', verbose=False)
  np.testing.assert_array_equal(ax, expected_x)
  np.testing.assert_array_equal(ay, expected_y)


@pytest.mark.parametrize('neg_error_to_accuracy', [True, False])
def test_normalize_maf_dataset_reference_vectors(neg_error_to_accuracy):
  # hyperbo/bo_utils/data_test.py:111-157
  maf = {'workload_a': {'X': np.array([[-1, 2, 1], [2, 2, 2]]), 'Y': np.array([[-.1], [0.]])},
         'workload_b': {'X': np.array([[1, 1, 2], [3, 2, 2]]), 'Y': np.array([[-.9], [-.2]])}}
  upd = (lambda v: v + 1) if neg_error_to_accuracy else (lambda v: v)
  expected = {'workload_a': {'X': np.array([[0, 1, 0], [.75, 1, 1]]), 'Y': upd(np.array([[-.1], [0.]]))},
              'workload_b': {'X': np.array([[.5, 0, 1], [1, 1, 1]]), 'Y': upd(np.array([[-.9], [-.2]]))}}
  out = data._normalize_maf_dataset(maf, num_hparams=3, neg_error_to_accuracy=neg_error_to_accuracy)
  for wl in expected:
    np.testing.assert_array_equal(expected[wl]['X'], out[wl]['X'])
    np.testing.assert_array_equal(expected[wl]['Y'], out[wl]['Y'])
