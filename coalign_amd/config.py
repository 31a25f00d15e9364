"""Config loading for the CoAlign hot path.

Host-side mirror of the reference's yaml loader so that *unchanged* opencood
``hypes_yaml`` files drive this package:

* ``load_yaml``                 <- opencood/hypes_yaml/yaml_utils.py:14-49
* ``load_point_pillar_params``  <- opencood/hypes_yaml/yaml_utils.py:97-137

Differences on purpose: the per-family post-parser named by the yaml's
``yaml_parser`` key is looked up in a table instead of being ``eval``-ed, and
unknown parser names raise instead of executing arbitrary text.
"""
from __future__ import annotations

import math
import os
import re
from typing import Callable, Dict

import numpy as np
import yaml

# PyYAML's default float resolver rejects "1e-10" (no dot); the reference
# patches the resolver the same way (yaml_utils.py:35-44) so that optimizer
# eps etc. parse as floats.  Semantics kept, expression written afresh.
_FLOAT_RE = re.compile(
    r"""^(?:[-+]?[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?
        |[-+]?[0-9][0-9_]*[eE][-+]?[0-9]+
        |\.[0-9_]+(?:[eE][-+][0-9]+)?
        |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
        |[-+]?\.(?:inf|Inf|INF)
        |\.(?:nan|NaN|NAN))$""",
    re.X,
)


class _Loader(yaml.Loader):
    pass


_Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT_RE, list("-+0123456789."))


def load_point_pillar_params(param: dict) -> dict:
    """Derive ``grid_size`` and anchor ``W/H/D`` from range and voxel size.

    Mirrors yaml_utils.py:97-137: ``grid_size = round((max-min)/voxel)`` as
    int64 ``[nx, ny, nz]`` injected into ``model.args.point_pillar_scatter``;
    ``anchor_args`` gains ``vw, vh, vd`` and ``W, H, D = ceil(extent/voxel)``.
    """
    rng = param["preprocess"]["cav_lidar_range"]
    vox = param["preprocess"]["args"]["voxel_size"]
    extent = np.asarray(rng[3:6], dtype=np.float64) - np.asarray(rng[0:3], dtype=np.float64)
    grid = np.round(extent / np.asarray(vox, dtype=np.float64)).astype(np.int64)
    if "model" in param:
        param["model"]["args"]["point_pillar_scatter"]["grid_size"] = grid

    anchor = param["postprocess"]["anchor_args"]
    anchor["vw"], anchor["vh"], anchor["vd"] = vox[0], vox[1], vox[2]
    anchor["W"] = math.ceil((rng[3] - rng[0]) / vox[0])
    anchor["H"] = math.ceil((rng[4] - rng[1]) / vox[1])
    anchor["D"] = math.ceil((rng[5] - rng[2]) / vox[2])
    param["postprocess"]["anchor_args"] = anchor
    return param


YAML_PARSERS: Dict[str, Callable[[dict], dict]] = {
    "load_point_pillar_params": load_point_pillar_params,
}


def load_yaml(file: str, opt=None) -> dict:
    """Load a hypes yaml; if ``opt.model_dir`` is set read ``config.yaml`` there
    (yaml_utils.py:30-31)."""
    if opt is not None and getattr(opt, "model_dir", None):
        file = os.path.join(opt.model_dir, "config.yaml")
    with open(file, "r") as fh:
        param = yaml.load(fh, Loader=_Loader)
    parser = param.get("yaml_parser")
    if parser:
        if parser not in YAML_PARSERS:
            raise KeyError(
                f"yaml_parser '{parser}' is outside the CoAlign hot path "
                f"(supported: {sorted(YAML_PARSERS)})"
            )
        param = YAML_PARSERS[parser](param)
    return param


CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def builtin_config(name: str) -> dict:
    """Load one of the hot-path configs shipped with this package
    (``coalign_amd/configs/<name>.yaml``)."""
    path = os.path.join(CONFIG_DIR, name if name.endswith(".yaml") else name + ".yaml")
    return load_yaml(path)
