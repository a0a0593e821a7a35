"""Per-task sub-sampling iterator -- hyperbo/basics/data_utils.py:72-100 with a NumPy Generator
in place of the JAX PRNG key (the reference's threefry streams cannot be reproduced without jax)."""
import numpy as np

from hyperbo_amd.basics import definitions as defs

SubDataset = defs.SubDataset


def draw_batch_indices(rng, sizes, batch_size):
  """Row indices of one batch: for every sub-dataset with at least batch_size rows a uniformly random subset of batch_size rows
  (what `permutation(n)[:batch_size]` gives, data_utils.py:86-90), None for the smaller ones (kept whole, no draw).
  ONE vectorised draw for the whole batch: a matrix of uniform keys, the batch_size smallest per row (rows padded beyond a
  task's size with keys that never win).  24 tasks of 400 rows: 60 us against 24 x rng.choice = 300 us -- the draw ran on a
  helper thread, but it holds the interpreter lock and was the largest part of a small Adam step.  Both iterators below use it,
  so they consume the generator identically."""
  sizes = [int(n) for n in sizes]
  out = [None] * len(sizes)
  need = [i for i, n in enumerate(sizes) if n >= batch_size]
  if not need:
    return out
  nmax = max(sizes[i] for i in need)
  if len(need) * nmax > (1 << 22):      # very large sub-datasets: task by task, without the key matrix
    for i in need:
      out[i] = rng.choice(sizes[i], batch_size, replace=False).astype(np.int32)
    return out
  keys = rng.random((len(need), nmax))
  for row, i in enumerate(need):
    if sizes[i] < nmax:
      keys[row, sizes[i]:] = 2.0
  idx = np.argpartition(keys, batch_size - 1, axis=1)[:, :batch_size] if batch_size < nmax else np.argsort(keys, axis=1)[:, :batch_size]
  idx = idx.astype(np.int32)
  for row, i in enumerate(need):
    out[i] = idx[row]
  return out


def sub_sample_dataset_iterator(key, dataset, batch_size):
  """Yields batches in which every sub-dataset has at most batch_size rows (random subset)."""
  rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
  while True:
    sub_sampled_dataset = {}
    drawn = draw_batch_indices(rng, [s.x.shape[0] for s in dataset.values()], batch_size)
    for i, (sub_dataset_key, sub_dataset) in enumerate(dataset.items()):
      if drawn[i] is not None:
        indices = drawn[i]   # (take() gathers rows three times faster than fancy indexing)
        new_sub_dataset = SubDataset(x=np.take(sub_dataset.x, indices, axis=0), y=np.take(sub_dataset.y, indices, axis=0),
                                     aligned=sub_dataset.aligned)
      else:
        new_sub_dataset = sub_dataset
      if isinstance(new_sub_dataset.aligned, str):   # data_utils.py:95-98
        new_sub_dataset = SubDataset(x=new_sub_dataset.x, y=new_sub_dataset.y, aligned=i)
      sub_sampled_dataset[sub_dataset_key] = new_sub_dataset
    yield sub_sampled_dataset


def sub_sample_index_iterator(key, dataset, batch_size):
  """The same draws as sub_sample_dataset_iterator, as row indices: yields {sub_dataset_key: int32 indices or None (kept whole)}.
  For batches gathered on the device from a resident copy of the dataset (objectives.DeviceBatch.subsample)."""
  rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
  keys = list(dataset)
  sizes = [dataset[k].x.shape[0] for k in keys]
  while True:
    yield dict(zip(keys, draw_batch_indices(rng, sizes, batch_size)))
