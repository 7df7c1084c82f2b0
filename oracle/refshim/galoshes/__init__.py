"""Stand-in for `galoshes` (AttributeMapper / BaseSCCache / SCFilter).

Semantics inferred from every call site in the reference (zephyr/backend/base.py:17-29,
zephyr/backend/discretization.py:116-153, zephyr/backend/distributors.py:36-67,283,452,540-563,
zephyr/middleware/problem.py:27-32, zephyr/middleware/survey.py:38-41).
Test infrastructure only (see README.md).
"""
import numpy as np


def _merged(cls, name, empty, merge):
    out = empty()
    for base in reversed(cls.__mro__):
        val = base.__dict__.get(name, None)
        if val:
            merge(out, val)
    return out


class AMMetaClass(type):
    def __new__(mcs, name, bases, attrs):
        cls = type.__new__(mcs, name, bases, attrs)
        full = {}
        for base in reversed(cls.__mro__):
            im = base.__dict__.get('_ownInitMap', None)
            if im:
                full.update(im)
        own = attrs.get('initMap', None) or {}
        cls._ownInitMap = dict(own)
        full.update(own)
        cls.initMap = full
        mk = set()
        for base in reversed(cls.__mro__):
            v = base.__dict__.get('_ownMaskKeys', None)
            if v:
                mk |= set(v)
        ownmk = set(attrs.get('maskKeys', ()) or ())
        cls._ownMaskKeys = ownmk
        cls.maskKeys = mk | ownmk
        ci = []
        for base in reversed(cls.__mro__):
            v = base.__dict__.get('_ownCacheItems', None)
            if v:
                ci.extend(v)
        ownci = list(attrs.get('cacheItems', ()) or ())
        cls._ownCacheItems = ownci
        cls.cacheItems = ci + ownci
        return cls


def _cast(typ, value):
    if typ is None:
        return value
    try:
        return typ(value)
    except TypeError:
        # complex scalar into a float field
        return typ(np.real(value))


class AttributeMapper(object, metaclass=AMMetaClass):
    initMap = {}
    maskKeys = set()
    cacheItems = []

    def __init__(self, systemConfig, *args, **kwargs):
        for key, (required, rename, typ) in self.initMap.items():
            if key in systemConfig:
                value = systemConfig[key]
                if value is not None:
                    value = _cast(typ, value)
                setattr(self, rename if rename else key, value)
            elif required:
                raise ValueError('Class %s requires parameter \'%s\'' % (type(self).__name__, key))


class BaseSCCache(AttributeMapper):
    def __init__(self, systemConfig, *args, **kwargs):
        AttributeMapper.__init__(self, systemConfig, *args, **kwargs)
        self.systemConfig = {k: systemConfig[k] for k in systemConfig if k not in self.maskKeys}

    @property
    def systemConfig(self):
        return self._systemConfig

    @systemConfig.setter
    def systemConfig(self, value):
        self._systemConfig = value
        self.clearCache()

    def clearCache(self):
        for name in self.cacheItems:
            if name in self.__dict__:
                delattr(self, name)


class SCFilter(object):
    def __init__(self, clslist):
        if not isinstance(clslist, (list, tuple)):
            clslist = [clslist]
        self.required = set()
        self.optional = set()
        for cls in clslist:
            for key, (req, _, _) in cls.initMap.items():
                (self.required if req else self.optional).add(key)

    def __call__(self, systemConfig):
        for key in self.required:
            if key not in systemConfig:
                raise ValueError('%s requires parameter \'%s\'' % (type(self).__name__, key))
        return {k: systemConfig[k] for k in systemConfig if k in self.required or k in self.optional}
