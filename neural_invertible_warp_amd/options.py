"""Command-line / yaml option handling with the reference's syntax (options.py:13-92):

    --key1.key2=value   (value parsed as yaml)      --key=   -> None      --key -> True      --key! -> False

`set()` builds the option tree from `--yaml=<name>`: from `<options_dir>/<name>.yaml` with the reference's
`_parent_` inheritance when a directory of yaml files is given (`--options_dir=...` or $NIW_OPTIONS_DIR, e.g. the
`options/` folder of a reference checkout), otherwise from the built-in trees of configs.py that restate the same
files' keys for the render path.  Unknown command-line keys are added (the reference asks interactively).
"""
import os
import random

import numpy as np
import torch
import yaml

from . import configs
from .util import edict


def _split_flag(token):
    """One command-line token -> (dotted key path as a list, yaml text of the value).  The three value-less spellings of the
    reference's syntax (options.py:16-22) map onto yaml literals: `--key` true, `--key!` false, `--key=` null (empty text)."""
    if not token.startswith("--") or len(token) < 3:
        raise ValueError(f"option {token!r}: expected --key[.subkey...][=value]")
    body = token[2:]
    name, eq, text = body.partition("=")
    if not eq:
        name, text = (body[:-1], "false") if body.endswith("!") else (body, "true")
    path = name.split(".")
    if not all(path):
        raise ValueError(f"option {token!r}: empty key component")
    return path, text


def parse_arguments(args):
    """`--a.b.c=value` tokens -> nested option tree (values parsed as yaml); a key given twice is an error, as in the reference."""
    tree = {}
    for token in args:
        path, text = _split_flag(token)
        node = tree
        for part in path[:-1]:
            node = node.setdefault(part, {})
            if not isinstance(node, dict):
                raise ValueError(f"option {token!r}: {part!r} already holds a value")
        leaf = path[-1]
        if leaf in node:
            raise ValueError(f"option {token!r}: {'.'.join(path)} given twice")
        node[leaf] = yaml.safe_load(text)
    return edict(tree)


def override_options(opt, opt_over):
    for key, value in opt_over.items():
        if isinstance(value, dict):
            opt[key] = override_options(opt.get(key, edict()), value)
        else:
            opt[key] = value
    return opt


def load_options(fname):
    """yaml file with `_parent_` inheritance; parent paths are relative to the directory above the file's folder
    (the reference writes them as "options/base.yaml") or to the file's folder"""
    with open(fname) as f:
        opt = edict(yaml.safe_load(f))
    if "_parent_" in opt:
        parents = opt.pop("_parent_")
        for parent in [parents] if isinstance(parents, str) else parents:
            here = os.path.dirname(os.path.abspath(fname))
            cands = [os.path.join(os.path.dirname(here), parent), os.path.join(here, os.path.basename(parent)), parent]
            opt = override_options(load_options(next(c for c in cands if os.path.isfile(c))), opt)
    return opt


def process_options(opt):
    if opt.seed is not None:
        random.seed(opt.seed)
        np.random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        if opt.seed != 0:
            opt.name = "{}_seed{}".format(opt.name, opt.seed)
    opt.output_path = "{0}/{1}/{2}".format(opt.output_root, opt.group, opt.name)
    os.makedirs(opt.output_path, exist_ok=True)
    local = int(os.environ.get("LOCAL_RANK", opt.gpu or 0))
    opt.device = "cpu" if opt.cpu or not torch.cuda.is_available() else "cuda:{}".format(local)
    opt.H, opt.W = opt.data.image_size
    return opt


def set(opt_cmd):
    assert "model" in opt_cmd and "yaml" in opt_cmd
    opt_cmd = edict(opt_cmd)
    options_dir = opt_cmd.pop("options_dir", None) or os.environ.get("NIW_OPTIONS_DIR")
    if options_dir:
        opt = load_options(os.path.join(options_dir, "{}.yaml".format(opt_cmd.yaml)))
    else:
        if opt_cmd.yaml not in configs.BY_YAML:
            raise KeyError("no built-in option tree for --yaml={} (have {}); pass --options_dir".format(opt_cmd.yaml, sorted(configs.BY_YAML)))
        opt = configs.BY_YAML[opt_cmd.yaml]()
    opt = override_options(opt, opt_cmd)
    return process_options(opt)


def save_options_file(opt):
    with open("{}/options.yaml".format(opt.output_path), "w") as f:
        yaml.safe_dump(_to_dict(opt), f, default_flow_style=False, indent=4)


def _to_dict(d):
    return {k: _to_dict(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}
