"""Drop-in check of the checkpoint format: every host-side mirror built from the config the REFERENCE class was built from
(oracle/make_golden_state_dicts.py -> tests/golden/ref_state_dicts.json) has the reference's parameter names and shapes --
no missing key, no extra key -- so `load_pretrained_model` of a published model.pt works unchanged."""
import json
import os

import pytest

import funasr_amd.auto_model  # noqa: F401  (registers every class)
from funasr_amd.register import tables

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_state_dicts.json")
with open(GOLD, encoding="utf-8") as _f:
    REF = json.load(_f)


@pytest.mark.parametrize("name", sorted(REF))
def test_state_dict_names_and_shapes_equal_the_reference(name):
    conf = REF[name]["config"]
    model = tables.model_classes.get(name.split("+")[0])(**conf)
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    ref = REF[name]["state_dict"]
    only_ref = sorted(set(ref) - set(mine))
    only_mine = sorted(set(mine) - set(ref))
    assert not only_ref and not only_mine, (name, only_ref[:8], only_mine[:8])
    wrong = [(k, mine[k], ref[k]) for k in ref if mine[k] != ref[k]]
    assert not wrong, (name, wrong[:8])
