"""systemConfig -> attributes layer.

Zephyr configures every object from one flat `systemConfig` dict; each class declares
`initMap = {key: (required, rename_or_None, cast_or_None)}` and the maps of all base classes are
merged (reference call sites: zephyr/backend/base.py:17-29, discretization.py:23-31,116-124,
distributors.py:28-36).  The reference delegates this to the external `galoshes` package
(setup.py:34), which is not part of its tree; this module is this project's own statement of
that contract: required keys raise ValueError when missing, present keys are cast and stored
under `rename or key` (going through property setters), `maskKeys` are withheld from the
stored `systemConfig` copy, `cacheItems` are dropped by `clearCache()`.
"""
import numpy as np


class ConfigMeta(type):
    def __new__(mcs, name, bases, namespace):
        own_map = dict(namespace.get('initMap', None) or {})
        own_mask = set(namespace.get('maskKeys', None) or ())
        own_cache = list(namespace.get('cacheItems', None) or ())
        cls = super().__new__(mcs, name, bases, namespace)
        merged_map, merged_mask, merged_cache = {}, set(), []
        for base in reversed(cls.__mro__[1:]):
            merged_map.update(base.__dict__.get('_own_init_map', {}))
            merged_mask |= base.__dict__.get('_own_mask_keys', set())
            for item in base.__dict__.get('_own_cache_items', []):
                if item not in merged_cache:
                    merged_cache.append(item)
        cls._own_init_map, cls._own_mask_keys, cls._own_cache_items = own_map, own_mask, own_cache
        merged_map.update(own_map)
        merged_mask |= own_mask
        for item in own_cache:
            if item not in merged_cache:
                merged_cache.append(item)
        cls.initMap, cls.maskKeys, cls.cacheItems = merged_map, merged_mask, merged_cache
        return cls


def cast_value(cast, value):
    if cast is None or value is None:
        return value
    if cast in (np.complex128, np.float64) and isinstance(value, np.ndarray) and value.ndim > 0 and not value.flags.writeable and value.dtype == np.dtype(cast):
        return value                     # (a read-only array of the right type is somebody's private copy already: discretization._spConfigs)
    try:
        return cast(value)
    except TypeError:
        return cast(np.real(value))      # complex scalar given for a float field


class AttributeMapper(metaclass=ConfigMeta):
    initMap = {}

    def __init__(self, systemConfig, *args, **kwargs):
        for key, (required, rename, cast) in self.initMap.items():
            if key in systemConfig:
                setattr(self, rename or key, cast_value(cast, systemConfig[key]))
            elif required:
                raise ValueError('%s requires parameter \'%s\'' % (type(self).__name__, key))


class BaseSCCache(AttributeMapper):
    def __init__(self, systemConfig, *args, **kwargs):
        AttributeMapper.__init__(self, systemConfig, *args, **kwargs)
        self.systemConfig = {k: v for k, v in systemConfig.items() if k not in self.maskKeys}

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
    """Keep only the keys a class (or list of classes) declares; check the required ones."""

    def __init__(self, classes):
        if not isinstance(classes, (list, tuple)):
            classes = [classes]
        self.required, self.known = set(), set()
        for cls in classes:
            for key, (required, _, _) in cls.initMap.items():
                self.known.add(key)
                if required:
                    self.required.add(key)

    def __call__(self, systemConfig):
        missing = [k for k in self.required if k not in systemConfig]
        if missing:
            raise ValueError('missing required parameter(s): %s' % ', '.join(sorted(missing)))
        return {k: v for k, v in systemConfig.items() if k in self.known}
