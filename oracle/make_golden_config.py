"""TEST INFRASTRUCTURE (development container only): records the SCHEMA of the reference's released experiment configs
into tests/golden/ref_config_keys.json, so that the config ingestion of SURVEY.md 8(f1) (holo_diffusion_amd/checkpoint.py)
is pinned to the reference's real YAMLs (configs/apple.yaml:68-253 and its four siblings) instead of a look-alike.

What is recorded per YAML (configs/{apple,hydrant,teddybear,donut,unet_with_no_diffusion}.yaml):
  * `top_level`: the top-level keys and, one level down, the keys of every `*_args` mapping outside the model (names only);
  * `model_factory`: the scalar fields of `model_factory_ImplicitronModelFactory_args` (resume, model_class_type, ...);
  * `model_args`: every leaf under `model_factory_ImplicitronModelFactory_args.model_HoloDiffusionModel_args` as
    [dotted key path, value] (scalars and lists of scalars: the inputs of the loader; `log_vars` - a list of 26 logging
    names - is recorded by length only), empty mappings as [path, {}];
  * `class_types`: every `*_class_type` value found anywhere in the file.
The YAML text itself is not kept.  Usage (here only; /root/reference does not exist on the GPU box):
    python oracle/make_golden_config.py
"""
import json
import os
import sys

import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HOLO_REFERENCE_ROOT", "/root/reference")
NAMES = ("apple", "hydrant", "teddybear", "donut", "unet_with_no_diffusion")
FACTORY = "model_factory_ImplicitronModelFactory_args"
MARGS = "model_HoloDiffusionModel_args"


def leaves(node, prefix=""):
    out = []
    for k, v in node.items():
        path = f"{prefix}.{k}" if prefix else k
        if isinstance(v, dict) and v:
            out.extend(leaves(v, path))
        elif path.endswith("log_vars"):
            out.append([path, {"__list_of_names__": len(v)}])
        else:
            out.append([path, v])
    return out


def class_types(node, prefix=""):
    out = {}
    for k, v in node.items():
        path = f"{prefix}.{k}" if prefix else k
        if isinstance(v, dict):
            out.update(class_types(v, path))
        elif k.endswith("_class_type"):
            out[path] = v
    return out


def main():
    rec = {}
    for name in NAMES:
        with open(os.path.join(REF, "configs", name + ".yaml")) as f:
            cfg = yaml.safe_load(f)
        fac = cfg[FACTORY]
        rec[name] = {
            "top_level": {k: (sorted(v) if isinstance(v, dict) and k != FACTORY else None) for k, v in cfg.items()},
            "model_factory": {k: v for k, v in fac.items() if k != MARGS},
            "model_args": leaves(fac[MARGS]),
            "class_types": class_types(cfg),
        }
    out = os.path.join(REPO, "tests", "golden", "ref_config_keys.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(out, {k: len(v["model_args"]) for k, v in rec.items()})


if __name__ == "__main__":
    sys.exit(main())
