"""Minimal stand-in for `yacs.config.CfgNode` (third-party, absent from this image).

Implements only the surface the reference's config module touches: nested
dict-with-attributes, clone/freeze/defrost, new_allowed, merge_from_*.
Our own code; used only by the golden-vector generator in the build container.
"""
import copy

import yaml

_IMMUTABLE = "__immutable__"
_NEW_ALLOWED = "__new_allowed__"


class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        key_list = [] if key_list is None else key_list
        init_dict = {
            k: (CfgNode(v, key_list + [k]) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)
            for k, v in init_dict.items()
        }
        super().__init__(init_dict)
        self.__dict__[_IMMUTABLE] = False
        self.__dict__[_NEW_ALLOWED] = new_allowed

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get(_IMMUTABLE, False):
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        if isinstance(value, dict) and not isinstance(value, CfgNode):
            value = CfgNode(value)
        self[name] = value

    def is_frozen(self):
        return self.__dict__[_IMMUTABLE]

    def is_new_allowed(self):
        return self.__dict__[_NEW_ALLOWED]

    def set_new_allowed(self, flag):
        self.__dict__[_NEW_ALLOWED] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v.set_new_allowed(flag)

    def _set_immutable(self, flag):
        self.__dict__[_IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def dump(self, **kwargs):
        def to_dict(n):
            return {k: (to_dict(v) if isinstance(v, CfgNode) else v) for k, v in n.items()}

        return yaml.safe_dump(to_dict(self), **kwargs)

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                if self.is_new_allowed():
                    self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else copy.deepcopy(v)
                    continue
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v, path + [k])
            else:
                self[k] = copy.deepcopy(v)

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(CfgNode(yaml.safe_load(f)), [])

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for full, v in zip(lst[0::2], lst[1::2]):
            keys = full.split(".")
            d = self
            for k in keys[:-1]:
                d = d[k]
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except Exception:
                    pass
            d[keys[-1]] = v

    def __deepcopy__(self, memo):
        new = CfgNode(new_allowed=self.__dict__[_NEW_ALLOWED])
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        new.__dict__[_IMMUTABLE] = self.__dict__[_IMMUTABLE]
        return new
