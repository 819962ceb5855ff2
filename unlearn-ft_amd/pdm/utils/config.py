"""Minimal OmegaConf-like attribute tree (omegaconf is not a dependency of the MI355X build): YAML -> nested `Cfg`
objects with attribute access, `.get`, `update(dict)` overlay of flat CLI args at the root (scripts/aptp/*.py:23-25)."""
import yaml


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [Cfg.wrap(v) for v in o]
        return o

    def get_path(self, path, default=None):
        cur = self
        for part in path.split("."):
            if not isinstance(cur, dict) or part not in cur or cur[part] is None:
                return default
            cur = cur[part]
        return cur


def load_config(path):
    with open(path) as f:
        return Cfg.wrap(yaml.safe_load(f) or {})
