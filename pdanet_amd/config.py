"""yaml -> attribute dict with the reference's `_BASE_CONFIG_` include + recursive merge rule
(/root/reference/pcdet/config.py:51-68), without the easydict dependency."""
import copy
import os

import yaml

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs")


class AttrDict(dict):
    """dict with attribute access (what the model code needs from EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_attr(v) for v in obj]
    return obj


def _merge(config, new_config, base_dir):
    # config.py:51-68: a `_BASE_CONFIG_` key pulls another yaml in first, then keys are merged
    # recursively (dicts merge, everything else is overwritten)
    if "_BASE_CONFIG_" in new_config:
        base_path = new_config["_BASE_CONFIG_"]
        for cand in (base_path, os.path.join(base_dir, base_path),
                     os.path.join(os.path.dirname(base_dir), base_path),
                     os.path.join(os.path.dirname(os.path.dirname(base_dir)), base_path)):
            if os.path.exists(cand):
                with open(cand) as f:
                    config.update(yaml.safe_load(f))
                break
        else:
            raise FileNotFoundError("_BASE_CONFIG_ %s" % base_path)
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config or not isinstance(config[key], dict):
            config[key] = {}
        _merge(config[key], val, base_dir)
    return config


def load_yaml(path):
    """Loads one of this repo's cfgs/*.yaml or a reference tools/cfgs/**/*.yaml unchanged."""
    if not os.path.exists(path) and os.path.exists(os.path.join(CFG_DIR, path)):
        path = os.path.join(CFG_DIR, path)
    with open(path) as f:
        raw = yaml.safe_load(f)
    return to_attr(_merge({}, raw, os.path.dirname(os.path.abspath(path))))
