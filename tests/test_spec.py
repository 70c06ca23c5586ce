"""Drop-in boundary: parameter names/shapes identical to the reference state_dict."""
import json
import os

import pytest

from s2m2_amd.spec import MODEL_CONFIGS, num_parameters, param_table

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", ["S", "M", "L", "XL"])
def test_param_table_matches_reference_dump(name):
    ref = json.load(open(os.path.join(HERE, "golden", f"state_dict_spec_{name}.json")))
    c, ntr = MODEL_CONFIGS[name]
    assert (c, ntr) == (ref["feature_channels"], ref["num_transformer"])
    mine = [[k, list(v)] for k, v in param_table(c, 1, ntr).items()]
    assert mine == ref["entries"]                     # same keys, same order, same shapes
    assert num_parameters(c, 1, ntr) == ref["num_parameters"]


def test_published_parameter_counts():
    # README.md:164-169 of the reference (verified by instantiation)
    assert num_parameters(128, 1, 1) == 26_501_043
    assert num_parameters(384, 1, 3) == 405_710_003
