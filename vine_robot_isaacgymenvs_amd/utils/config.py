"""Config composition for ``train.py``: the subset of Hydra 1.1 + OmegaConf 2.1 grammar that the reference's
``cfg/config.yaml``, ``cfg/task/Vine5LinkMovingBase.yaml`` and ``cfg/train/Vine5LinkMovingBasePPO.yaml`` use
(neither hydra nor omegaconf exists on the target image).

Supported, with the reference line that needs it:
  * defaults list with ``_self_``, config groups and ``${task}PPO``        (cfg/config.yaml:2-5)
  * CLI overrides ``task=Name key=val a.b.c=val +new=val``                  (README.md:63,71; train.py:178)
  * absolute ``${task.name}``, relative ``${.k}`` ``${..k}`` ``${...k}``    (config.yaml:9; task yaml:4,8,65,111)
  * resolvers eq / contains / if / resolve_default / eval, nested args,
    quoted args, interpolation embedded in a quoted string                 (isaacgymenvs/__init__.py:8-12; task yaml:64)
  * OmegaConf scalar typing: ``2e-2`` and ``3e-4`` are floats (PyYAML alone reads them as strings)
Everything is resolved eagerly into plain dicts (what ``omegaconf_to_dict`` returns, utils/reformat.py:32).
"""
import copy
import os
import re

import yaml

# config_dir=None -> the packaged defaults (cfg/defaults.py, Python literals); a path -> a Hydra-style YAML directory
# (config.yaml + task/<name>.yaml + train/<name>.yaml), e.g. the reference's own isaacgymenvs/cfg.


class ConfigError(ValueError):
    pass


# --------------------------------------------------------------------------- YAML with OmegaConf float typing
class _Loader(yaml.SafeLoader):
    pass


_FLOAT_RE = re.compile(r"""^[-+]?(?:
      (?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
    | \.[0-9][0-9_]*(?:[eE][-+]?[0-9]+)?
    | [0-9][0-9_]*[eE][-+]?[0-9]+
    | \.(?:inf|Inf|INF)
    | \.(?:nan|NaN|NAN))$""", re.X)
_Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT_RE, list("-+0123456789."))
# OmegaConf (YAML 1.2 core schema) types only true/false as booleans; PyYAML's YAML 1.1 yes/no/on/off stay strings.
_Loader.yaml_implicit_resolvers = {
    ch: [(tag, rx) for tag, rx in lst if tag != "tag:yaml.org,2002:bool"]
    for ch, lst in _Loader.yaml_implicit_resolvers.items()}
_Loader.add_implicit_resolver("tag:yaml.org,2002:bool", re.compile(r"^(?:true|True|TRUE|false|False|FALSE)$"),
                              list("tTfF"))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


def parse_scalar(text):
    """Type a CLI override value / resolver argument the way OmegaConf does."""
    if not isinstance(text, str):
        return text
    s = text.strip()
    if s == "":
        return ""
    if (s[0] == s[-1]) and s[0] in "\"'" and len(s) >= 2:
        return s[1:-1]
    try:
        v = _yaml_load(s)
    except yaml.YAMLError:
        return s
    return v


# --------------------------------------------------------------------------- resolvers (isaacgymenvs/__init__.py:8-12)
def _eval(expr):
    return eval(str(expr), {"__builtins__": {}}, {})  # same contract as the reference's bare eval, no builtins


RESOLVERS = {
    "eq": lambda x, y: str(x).lower() == str(y).lower(),
    "contains": lambda x, y: str(x).lower() in str(y).lower(),
    "if": lambda pred, a, b: a if pred else b,
    "resolve_default": lambda default, arg: default if arg == "" else arg,
    "eval": _eval,
}


# --------------------------------------------------------------------------- interpolation
def _find_close(s, start):
    """Index of the '}' matching the '${' whose body starts at `start`; quotes protect braces."""
    depth, i, quote = 1, start, None
    while i < len(s):
        ch = s[i]
        if quote:
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
        elif s.startswith("${", i):
            depth += 1
            i += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ConfigError("unbalanced interpolation in %r" % s)


def _split_args(body):
    args, depth, quote, cur = [], 0, None, ""
    i = 0
    while i < len(body):
        ch = body[i]
        if quote:
            cur += ch
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
            cur += ch
        elif body.startswith("${", i):
            depth += 1
            cur += "${"
            i += 1
        elif ch == "}":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    args.append(cur)
    return args


class _Resolver:
    def __init__(self, root):
        self.root = root
        self._active = set()

    def lookup(self, path_keys):
        node = self.root
        for k in path_keys:
            if isinstance(node, list):
                node = node[int(k)]
            elif isinstance(node, dict) and k in node:
                node = node[k]
            else:
                raise ConfigError("interpolation key %r not found" % ".".join(path_keys))
        return node

    def resolve_path(self, ref, at):
        """`ref` like 'task.name', '.x', '...num_envs'; `at` = key path of the node holding the interpolation."""
        if ref.startswith("."):
            dots = len(ref) - len(ref.lstrip("."))
            base = list(at[:-1])            # '.' = the container of the current node
            up = dots - 1
            if up > len(base):
                raise ConfigError("relative interpolation %r climbs above the root" % ref)
            base = base[:len(base) - up] if up else base
            keys = base + [k for k in ref.lstrip(".").split(".") if k]
        else:
            keys = ref.split(".")
        key_id = tuple(keys)
        if key_id in self._active:
            raise ConfigError("interpolation cycle at %r" % ".".join(keys))
        self._active.add(key_id)
        try:
            return self.value(self.lookup(keys), keys)
        finally:
            self._active.discard(key_id)

    def expr(self, body, at):
        body = body.strip()
        m = re.match(r"^([A-Za-z_][A-Za-z0-9_]*)\s*:(.*)$", body, re.S)
        if m and m.group(1) in RESOLVERS:
            args = [self.arg(a, at) for a in _split_args(m.group(2))]
            return RESOLVERS[m.group(1)](*args)
        if m:
            raise ConfigError("unknown resolver %r" % m.group(1))
        return self.resolve_path(body, at)

    def arg(self, text, at):
        s = text.strip()
        if len(s) >= 2 and s[0] == s[-1] and s[0] in "\"'":
            return self.string(s[1:-1], at, typed=False)
        if "${" in s:
            return self.string(s, at, typed=True)
        return parse_scalar(s)

    def string(self, s, at, typed=True):
        """Resolve every ${...} in s.  A string that is exactly one interpolation keeps the value's type."""
        if "${" not in s:
            return s
        out, i, parts = "", 0, []
        while i < len(s):
            j = s.find("${", i)
            if j < 0:
                parts.append(("lit", s[i:]))
                break
            if j > i:
                parts.append(("lit", s[i:j]))
            k = _find_close(s, j + 2)
            parts.append(("val", self.expr(s[j + 2:k], at)))
            i = k + 1
        if typed and len(parts) == 1 and parts[0][0] == "val":
            return parts[0][1]
        for kind, v in parts:
            out += v if kind == "lit" else ("" if v is None else str(v))
        return out

    def value(self, node, at):
        if isinstance(node, str):
            return self.string(node, at)
        if isinstance(node, dict):
            return {k: self.value(v, list(at) + [k]) for k, v in node.items()}
        if isinstance(node, list):
            return [self.value(v, list(at) + [str(i)]) for i, v in enumerate(node)]
        return node


def resolve(cfg):
    """Eagerly resolve all interpolations; returns a new plain dict."""
    return _Resolver(cfg).value(cfg, [])


# --------------------------------------------------------------------------- composition
def _read(path):
    if not os.path.exists(path):
        raise ConfigError("config file not found: %s" % path)
    with open(path) as f:
        return _yaml_load(f.read()) or {}


def _load_primary(config_dir, config_name):
    if config_dir is None:
        from ..cfg import defaults
        if config_name != "config":
            raise ConfigError("packaged defaults only provide the root config 'config'")
        return copy.deepcopy(defaults.ROOT)
    return _read(os.path.join(config_dir, config_name + ".yaml"))


def _load_group(config_dir, group, name):
    if config_dir is None:
        from ..cfg import defaults
        table = {"task": defaults.TASK, "train": defaults.TRAIN}.get(group, {})
        if name not in table:
            raise ConfigError("no packaged %s config named %r (available: %s)" % (group, name, sorted(table)))
        return copy.deepcopy(table[name])
    return _read(os.path.join(config_dir, group, str(name) + ".yaml"))


def _deep_merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg, dotted, value, add):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not add:
                raise ConfigError("Could not override '%s': key '%s' is not in the config (use +%s=... to add it)"
                                  % (dotted, k, dotted))
            node[k] = {}
        node = node[k]
    if keys[-1] not in node and not add:
        raise ConfigError("Could not override '%s': no such key in the config (use +%s=... to add it)"
                          % (dotted, dotted))
    node[keys[-1]] = value


def compose(config_name="config", overrides=(), config_dir=None):
    """Hydra-style composition; returns the UNRESOLVED merged dict (interpolations still as strings)."""
    primary = _load_primary(config_dir, config_name)
    defaults = primary.pop("defaults", ["_self_"])
    primary.pop("hydra", None)
    group_choice, plain = {}, []
    groups = [list(d.keys())[0] for d in defaults if isinstance(d, dict)]
    for ov in overrides:
        if "=" not in ov:
            raise ConfigError("override %r is not of the form key=value" % ov)
        key, val = ov.split("=", 1)
        if key.lstrip("+") in groups and "." not in key:
            group_choice[key.lstrip("+")] = val
        else:
            plain.append((key, val))
    cfg = {}
    chosen = {}
    for d in defaults:
        if d == "_self_":
            _deep_merge(cfg, primary)
        elif isinstance(d, dict):
            (group, default_choice), = d.items()
            choice = group_choice.get(group, default_choice)
            if isinstance(choice, str) and "${" in choice:      # e.g. train: ${task}PPO
                choice = re.sub(r"\$\{([A-Za-z_]+)\}", lambda m: str(chosen[m.group(1)]), choice)
            chosen[group] = choice
            _deep_merge(cfg, {group: _load_group(config_dir, group, choice)})
        else:
            raise ConfigError("unsupported defaults entry %r" % (d,))
    if "_self_" not in defaults:
        _deep_merge(cfg, primary)
    for key, val in plain:
        add = key.startswith("+")
        _set_path(cfg, key.lstrip("+"), parse_scalar(val), add)
    return cfg


def load_config(config_name="config", overrides=(), config_dir=None):
    """Compose + resolve: the dict ``omegaconf_to_dict(cfg)`` yields in the reference (train.py:64)."""
    return resolve(compose(config_name, overrides, config_dir))


def load_task_config(task="Vine5LinkMovingBase", overrides=(), config_dir=None):
    """What ``isaacgymenvs.make`` builds when no cfg is passed (isaacgymenvs/__init__.py:37-40)."""
    ov = ["task=" + task] + [o for o in overrides if not o.startswith("task=")]
    return load_config("config", ov, config_dir)["task"]


def to_yaml(cfg):
    return yaml.safe_dump(cfg, sort_keys=False, default_flow_style=False)


def dump_default_yaml(directory):
    """Write the packaged defaults as an editable Hydra-style directory (config.yaml, task/, train/)."""
    from ..cfg import defaults
    os.makedirs(os.path.join(directory, "task"), exist_ok=True)
    os.makedirs(os.path.join(directory, "train"), exist_ok=True)
    with open(os.path.join(directory, "config.yaml"), "w") as f:
        f.write(to_yaml(defaults.ROOT))
    for name, tree in defaults.TASK.items():
        with open(os.path.join(directory, "task", name + ".yaml"), "w") as f:
            f.write(to_yaml(tree))
    for name, tree in defaults.TRAIN.items():
        with open(os.path.join(directory, "train", name + ".yaml"), "w") as f:
            f.write(to_yaml(tree))
    return directory


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "--dump-yaml":
        print("wrote", dump_default_yaml(sys.argv[2]))
    else:
        print(to_yaml(load_config(overrides=sys.argv[1:])))
