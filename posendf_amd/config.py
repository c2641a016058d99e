"""Config surface: the reference's YAML -> nested dict loader (reference configs/config.py:2-6) plus the
architecture block of configs/amass.yaml:22-43 as a ready-made dict."""
from __future__ import annotations

import copy
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
AMASS_YAML = os.path.join(_HERE, "configs", "amass.yaml")


def load_config(path):
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def amass_config(act: str = "lrelu", device: str = "cuda"):
    cfg = copy.deepcopy(load_config(AMASS_YAML))
    cfg["model"]["DFNet"]["act"] = act
    cfg["model"]["StrEnc"]["act"] = act
    cfg["train"]["device"] = device
    return cfg
