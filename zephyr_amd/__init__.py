"""zephyr_amd -- MI355X-native frequency-domain Helmholtz forward solver behind Zephyr's
backend operator API (`Disc(systemConfig) * rhs`, `MultiFreq(systemConfig) * rhs`).

Host classes mirror zephyr.backend; all assembly and solves run in libhelm (HIP, gfx950).
"""
from .analytical import AnalyticalHelmholtz
from .base import BaseModelDependent, BaseAnisotropic
from .config import AttributeMapper, BaseSCCache, SCFilter
from .discretization import BaseDiscretization, DiscretizationWrapper, prefactor_many
from .distributors import BaseDist, BaseMPDist, MultiFreq, SerialMultiFreq, ViscoMultiFreq
from .eurus import Eurus, EurusHD
from .helm3d import Helm3D
from .minizephyr import MiniZephyr, MiniZephyrHD, MiniZephyr25D
from .source import (FakeSource, SimpleSource, StackedSimpleSource, SparseKaiserSource, KaiserSource,
                     AnisotropicKaiserSource)

def trim():
    'give the scratch memory libhelm caches between calls (tens of GB after large direct solves) back to the device'
    from . import _lib
    return _lib.load().helm_trim()


__all__ = [
    'AnalyticalHelmholtz', 'BaseModelDependent', 'BaseAnisotropic', 'AttributeMapper', 'BaseSCCache', 'SCFilter',
    'BaseDiscretization', 'DiscretizationWrapper', 'prefactor_many', 'BaseDist', 'BaseMPDist', 'MultiFreq', 'SerialMultiFreq',
    'ViscoMultiFreq', 'Eurus', 'EurusHD', 'Helm3D', 'MiniZephyr', 'MiniZephyrHD', 'MiniZephyr25D', 'FakeSource', 'SimpleSource',
    'StackedSimpleSource', 'SparseKaiserSource', 'KaiserSource', 'AnisotropicKaiserSource', 'trim',
]
