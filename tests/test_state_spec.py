"""CPU: parameter layout == the reference's state_dict layout (key order and
shapes), for the product tables, the oracle tables and the golden dump."""
import json
import os

import pytest

from jarvis_hybridnet_amd import arch
from oracle import hybridnet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "state_spec.json")) as f:
    SPEC = json.load(f)


@pytest.mark.parametrize("size", ["small", "medium", "large"])
@pytest.mark.parametrize("J", [1, 23])
def test_efficienttrack_layout(size, J):
    ref = [(k, tuple(s)) for k, s in SPEC["efficienttrack.%s.%d" % (size, J)]]
    assert arch.efficienttrack_params(size, J) == ref
    assert O.efficienttrack_state_spec(size, J) == ref


def test_hybridnet_layout():
    ref = [(k, tuple(s)) for k, s in SPEC["hybridnet.small.23"]]
    assert len(ref) == 188
    assert arch.hybridnet_params("small", 23) == ref
    assert O.hybridnet_state_spec("small", 23) == ref
