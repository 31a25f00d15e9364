"""Regenerate tests/golden/yaml_routes.json: coalign_amd.routes.summary(plan(.)) of every hypes_yaml/**/pointpillar*.yaml of the reference checkout
(/root/reference, build container only).  The table records which kernel serves which layer of each of the reference's configs, decided without a GPU;
tests/test_host_cpu.py::test_kernel_routes_of_every_config compares against it.

    python tests/golden/make_routes.py
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coalign_amd.config import load_yaml  # noqa: E402
from coalign_amd.routes import plan, summary  # noqa: E402

BASE = "/root/reference/opencood/hypes_yaml/"


def main():
    table = {}
    for path in sorted(glob.glob(BASE + "**/pointpillar*.yaml", recursive=True)):
        try:
            table[path[len(BASE):]] = summary(plan(load_yaml(path)))
        except Exception as e:      # noqa: BLE001  (other model families' yamls need parsers / keys outside the hot path)
            table[path[len(BASE):]] = {"outside_hot_path": f"{type(e).__name__}: {str(e)[:120]}"}
    json.dump(table, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "yaml_routes.json"), "w"), indent=1, sort_keys=True)
    hot = [k for k, v in table.items() if not v.get("outside_hot_path")]
    print(f"{len(table)} yamls, {len(hot)} in the hot path")


if __name__ == "__main__":
    main()
