"""Job profiles: mix-ins that select the Problem / Survey / discretisation / input / output of a run.

Interface of zephyr/frontend/jobs.py:13-230.  The reference's `Solver` selection (MUMPS, falling back to SuperLU,
jobs.py:27-32) has no counterpart: the discretisations solve on the GPU through libhelm.
"""
import pickle

from . import omega
from .eurus import EurusHD
from .minizephyr import MiniZephyrHD
from .problem import Helm2DViscoProblem
from .survey import Helm2DSurvey


class Job(object):

    Problem = None
    Survey = None
    SystemWrapper = None
    Disc = None
    projnm = None

    def __init__(self, projnm, supplementalConfig=None, verbose=True):
        self.projnm = projnm
        self.verbose = verbose
        self._say('Setting up composite job "%s":' % (self.__class__.__name__,))
        for item in self.__class__.__mro__[:-1][::-1]:
            self._say('\t%s' % (item.__name__,))
        self._say('')

        systemConfig = self.getSystemConfig(projnm)
        if self.SystemWrapper is not None:
            systemConfig['SystemWrapper'] = self.SystemWrapper
        if self.Disc is not None:
            systemConfig['Disc'] = self.Disc
        if supplementalConfig is not None:
            systemConfig.update(supplementalConfig)
        if 'projnm' not in systemConfig:
            systemConfig['projnm'] = projnm

        self.systemConfig = systemConfig
        self.problem = self.Problem(systemConfig)
        self.survey = self.Survey(systemConfig)
        self.problem.pair(self.survey)

    def _say(self, msg):
        if self.verbose:
            print(msg)

    def getSystemConfig(self, projnm):
        raise NotImplementedError

    def run(self):
        raise NotImplementedError

    def saveData(self, data):
        raise NotImplementedError


class ForwardModelingJob(Job):
    'jobs.py:88-109'

    def run(self):
        self._say('Running %s(%s)...' % (self.__class__.__name__, self.projnm))
        self._say('\t- solving system')
        data = self.survey.dpred()
        data.shape = (self.survey.nrec, self.survey.nsrc, self.survey.nfreq)
        self._say('\t- saving data')
        self.saveData(data)
        self._say('Done!')
        return data


class Visco2DJob(Job):
    Problem = Helm2DViscoProblem
    Survey = Helm2DSurvey


class IsotropicVisco2DJob(Visco2DJob):
    Disc = MiniZephyrHD


class AnisotropicVisco2DJob(Visco2DJob):
    Disc = EurusHD


class IniInputJob(Job):
    def getSystemConfig(self, projnm):
        self.ds = omega.FullwvDatastore(projnm)
        return self.ds.systemConfig


class PythonInputJob(Job):
    def getSystemConfig(self, projnm):
        self.ds = omega.FlatDatastore(projnm)
        return self.ds.systemConfig


class PickleInputJob(Job):
    def getSystemConfig(self, projnm):
        self.ds = omega.PickleDatastore(projnm)
        return self.ds.systemConfig


class UtoutOutputJob(Job):
    def saveData(self, data):
        omega.UtoutWriter(self.systemConfig)(data)


class PickleOutputJob(Job):
    def saveData(self, data):
        with open(self.projnm, 'wb') as fp:
            pickle.Pickler(fp).dump(data)


class OmegaIOJob(IniInputJob, UtoutOutputJob):
    pass


class OmegaJob(IsotropicVisco2DJob, ForwardModelingJob, OmegaIOJob):
    'roughly the default behaviour of OMEGA: ini + SEG-Y in, MiniZephyrHD, .utout out (jobs.py:202-207)'


class PythonUtoutJob(IsotropicVisco2DJob, ForwardModelingJob, PythonInputJob, UtoutOutputJob):
    pass


class AnisoOmegaJob(AnisotropicVisco2DJob, ForwardModelingJob, OmegaIOJob):
    pass


class AnisoPythonUtoutJob(AnisotropicVisco2DJob, ForwardModelingJob, PythonInputJob, UtoutOutputJob):
    pass
