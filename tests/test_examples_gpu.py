"""examples/ stay runnable: the Wide&Deep Criteo flow learns (AUC above chance on the held-out split) and the SASRec flow
retrieves the true next item."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wide_deep_criteo_example(dev):
    assert load("train_wide_deep_criteo").main() > 0.55


def test_sasrec_example(dev):
    assert load("train_sasrec_synthetic").main() > 0.5
