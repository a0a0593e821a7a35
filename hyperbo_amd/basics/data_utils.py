"""Per-task sub-sampling iterator -- hyperbo/basics/data_utils.py:72-100 with a NumPy Generator
in place of the JAX PRNG key (the reference's threefry streams cannot be reproduced without jax)."""
import numpy as np

from hyperbo_amd.basics import definitions as defs

SubDataset = defs.SubDataset


def sub_sample_dataset_iterator(key, dataset, batch_size):
  """Yields batches in which every sub-dataset has at most batch_size rows (random subset)."""
  rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
  while True:
    sub_sampled_dataset = {}
    for i, (sub_dataset_key, sub_dataset) in enumerate(dataset.items()):
      if sub_dataset.x.shape[0] >= batch_size:
        # a uniformly random ordered subset, as permutation(n)[:batch_size] (data_utils.py:86-90 uses jax.random.permutation),
        # drawn without shuffling all n indices; take() gathers rows three times faster than fancy indexing
        indices = rng.choice(sub_dataset.x.shape[0], batch_size, replace=False)
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
  while True:
    out = {}
    for sub_dataset_key, sub_dataset in dataset.items():
      n = sub_dataset.x.shape[0]
      out[sub_dataset_key] = rng.choice(n, batch_size, replace=False).astype(np.int32) if n >= batch_size else None
    yield out
