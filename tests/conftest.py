import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu_ctx():
  from hyperbo_amd import _native as nat
  if nat.lib().hbo_device_count() <= 0:
    pytest.fail('GPU test selected but no HIP device is visible (libhbo has no CPU fallback)')
  return nat.default_context()
