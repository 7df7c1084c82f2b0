"""OMEGA / FULLWV project I/O: `.ini` reader, SEG-Y model reader, `.utout` writer, datastores.

Interfaces of zephyr/middleware/util.py:21-178 (`readini`, `compileDict`), zephyr/middleware/db.py:35-310
(`UtoutWriter`, `FullwvDatastore`, `FlatDatastore`, `PickleDatastore`) and the two pieces of
zephyr/middleware/time.py (:29-49, :79-99, :189-199) they lean on.  The reference reads SEG-Y through the external
`pygeo.segyread.SEGYFile` (not in its tree); `SEGYFile` here follows the SEG-Y rev-1 layout directly.
"""
import glob
import os
import pickle
import re

import numpy as np
import scipy.io as sio

from .config import AttributeMapper


# ---- .ini ------------------------------------------------------------------------------------------------------
def str2bool(v):
    return v.lower() in ('yes', 'true', 't', '1')


class _Lines(object):
    'cursor over the positional lines of an OMEGA ini file: header lines (in <...>) alternate with value lines'

    def __init__(self, lines):
        self.lines = lines

    def fields(self, i, strip_quotes=False):
        line = self.lines[i]
        if strip_quotes:
            line = line.replace('\'', '')
        return line.strip().split()

    def block(self, start, count, per_line=5):
        'count floats laid out per_line to a row starting at line `start`; returns (values, index after the block)'
        nlines = count // per_line + (1 if count % per_line else 0)
        vals = []
        for line in self.lines[start:start + nlines]:
            vals.extend(float(tok) for tok in line.strip().split())
        return np.array(vals), start + nlines

    def table(self, start, count):
        'count rows "<index> v1 v2 ..." -> array of the values without the index column'
        rows = [[float(tok) for tok in self.lines[i].strip().split()[1:]] for i in range(start, start + count)]
        return np.array(rows), start + count


def readini(infile):
    """Parse a (2.5-D) OMEGA `.ini` project file into a flat dict (util.py:21-157).

    The format is positional: every value line is preceded by a `<name> <name> ...` caption line."""
    with open(infile, 'r') as fp:
        L = _Lines(fp.readlines())
    d = {}

    f = L.fields(1)
    d['comment'], d['lessfiles'] = int(f[0]), str2bool(f[1])

    f = L.fields(3)
    d['nx'], d['nz'] = int(f[0]), int(f[1])
    d['dx'], d['dz'], d['xorig'], d['zorig'] = (float(v) for v in f[2:6])

    f = L.fields(5, True)
    d['inv'] = str2bool(f[0])
    d['datain'], d['dataout'] = f[1], f[2]
    d['waveout'] = int(f[3])
    d['usescratch'] = str2bool(f[4])
    d['nom'], d['nsam'] = int(f[5]), int(f[6])
    d['tau'] = float(f[7])
    d['nftout'] = int(f[8])

    f = L.fields(7, True)
    d['we'] = f[0]
    d['param'], d['nky'], d['method'] = int(f[1]), int(f[2]), int(f[3])
    d['vmin'], d['deltatt'] = float(f[4]), float(f[5])
    d['src'] = int(f[6])
    d['wavscale'] = str2bool(f[7])
    d['aniso'], d['freqbase'] = float(f[8]), float(f[9])

    f = L.fields(9)
    d['reduce'] = str2bool(f[0])
    d['redvel'], d['tbegin'] = float(f[1]), float(f[2])
    d['fst'], d['fsr'], d['fsb'], d['fsl'] = (str2bool(v) for v in f[3:7])
    d['sponge'] = str2bool(f[7])
    d['isufx'] = int(f[8])

    d['freqs'], end = L.block(11, d['nom'])
    d['kys'], end = L.block(end + 1, d['nky'])

    d['nslices'] = int(L.fields(end + 1)[0])
    start = end + 3
    slices = []
    for i in range(start, start + d['nslices']):
        f = L.fields(i)
        slices.append([int(f[0]), int(f[1]), float(f[2])] + f[3:])
        d['slices'] = slices
    end = start + d['nslices']

    for count, reg, spread, usewt, table in (('ns', 'isreg', 'sspread', 'useswt', 'srcs'),
                                              ('nr', 'irreg', 'rspread', 'userwt', 'recs'),
                                              ('ng', 'igreg', 'gspread', 'usegwt', 'geos')):
        f = L.fields(end + 1)
        d[count], d[reg], d[spread], d[usewt] = int(f[0]), int(f[1]), float(f[2]), str2bool(f[3])
        d[table], end = L.table(end + 3, d[count])

    f = L.fields(end + 1)
    d['sghost'], d['rghost'], d['gghost'] = (str2bool(v) for v in f[:3])
    d['zgg'] = float(f[3])
    d['zero1'] = [int(v) for v in L.fields(end + 3)]
    d['zero2'] = [int(v) for v in L.fields(end + 4)]
    return d


def compileDict(projnm, exprdict):
    'pre-compile a dict of filename regular expressions, substituting the project name where used (util.py:159-178)'
    out = {}
    for key, expr in exprdict.items():
        try:
            out[key] = re.compile(expr % projnm)
        except TypeError:
            out[key] = re.compile(expr)
    return out


# ---- SEG-Y -----------------------------------------------------------------------------------------------------
def ibm2ieee(words):
    'IBM System/360 single precision, given as big-endian uint32 words -> float64'
    w = np.asarray(words).astype(np.uint32)
    mant = (w & np.uint32(0x00ffffff)).astype(np.float64)
    expo = ((w >> np.uint32(24)) & np.uint32(0x7f)).astype(np.int32) - 64 - 6      # mantissa has 6 hex digits
    out = np.ldexp(mant, 4 * expo)
    return np.where(w >> np.uint32(31), -out, out)


class SEGYFile(object):
    """Minimal SEG-Y reader with the slice access the datastore uses (`sf[:]` -> (ntr, ns) array).

    3200-byte textual header, 400-byte binary header (samples/trace at bytes 3221-3222, format code at 3225-3226),
    240-byte trace headers; format 1 (IBM float) and 5 (IEEE float), big-endian."""

    def __init__(self, filename):
        self.filename = filename
        raw = np.fromfile(filename, dtype=np.uint8)
        if raw.size < 3600:
            raise IOError('%s is too short to be a SEG-Y file' % filename)
        self.ns = int(raw[3220]) << 8 | int(raw[3221])
        self.fmt = int(raw[3224]) << 8 | int(raw[3225])
        if self.fmt not in (1, 5):
            raise NotImplementedError('SEG-Y sample format code %d' % self.fmt)
        tlen = 240 + 4 * self.ns
        self.ntr = (raw.size - 3600) // tlen
        body = raw[3600:3600 + self.ntr * tlen].reshape((self.ntr, tlen))[:, 240:]
        words = np.ascontiguousarray(body).view('>u4')
        self._traces = ibm2ieee(words) if self.fmt == 1 else words.view('>f4').astype(np.float64)

    def __getitem__(self, sl):
        return self._traces[sl]

    def __len__(self):
        return self.ntr


# ---- time <-> frequency helpers ------------------------------------------------------------------------------------
class BaseTimeSensitive(AttributeMapper):
    'time.py:79-99'

    initMap = {
        'freqs':        (True,      None,           list),
        'tau':          (False,     '_tau',         np.float64),
    }

    @property
    def tau(self):
        return getattr(self, '_tau', np.inf)

    @property
    def dampCoeff(self):
        return 1j / self.tau


def dftreal(a, N, M):
    'naive forward DFT of M real column vectors of N samples, e^{+2 pi i nk/N}/N convention (time.py:29-49)'
    n = np.arange(N).reshape((N, 1))
    W = np.exp(2j * np.pi / N) ** (n.T * n)
    return np.dot(W, a[:N, :M]) / N


class UtoutWriter(BaseTimeSensitive):
    """Frequency-domain data -> `<projnm>.utout`: one Fortran sequential record per frequency holding a
    complex64 panel (nsrc, nrec+1) whose first column is omega + i/tau (db.py:35-66)."""

    initMap = {
        'projnm':       (True,      None,           str),
    }

    def __call__(self, data, fid=slice(None), ftype='utout'):
        ofreqs = [(2 * np.pi * freq) + self.dampCoeff for freq in np.asarray(self.freqs)[fid].tolist()]
        outfile = '%s.%s' % (self.projnm, ftype)
        if data.ndim != 3:
            raise Exception('Data must be of shape (nrec, nsrc, nfreq)')
        assert data.shape[2] == len(ofreqs)
        nrec, nsrc = data.shape[:2]
        with sio.FortranFile(outfile, 'w') as ff:
            for i, omega in enumerate(ofreqs):
                panel = np.empty((nsrc, nrec + 1), dtype=np.complex64)
                panel[:, :1] = omega
                panel[:, 1:] = data[:, :, i].T
                ff.write_record(panel.ravel())


def utoutRead(filename, nrec):
    'inverse of UtoutWriter: returns (omegas (nfreq,), data (nrec, nsrc, nfreq)) as complex64'
    panels = []
    with sio.FortranFile(filename, 'r') as ff:
        while True:
            try:
                rec = ff.read_record(np.complex64)
            except Exception:
                break
            panels.append(rec.reshape((-1, nrec + 1)))
    omegas = np.array([p[0, 0] for p in panels])
    data = np.stack([p[:, 1:].T for p in panels], axis=2)
    return omegas, data


# ---- datastores ----------------------------------------------------------------------------------------------------
ftypeRegex = {
    'vp':       r'^%s(?P<iter>[0-9]*)\.vp(?P<freq>[0-9]*\.?[0-9]+)?[^i]*$',
    'qp':       r'^%s(?P<iter>[0-9]*)\.qp(?P<freq>[0-9]*\.?[0-9]+)?.*$',
    'vpi':      r'^%s(?P<iter>[0-9]*)\.vpi(?P<freq>[0-9]*\.?[0-9]+)?.*$',
    'rho':      r'^%s\.rho$',
    'eps2d':    r'^%s\.eps2d$',
    'del2d':    r'^%s\.del2d$',
    'theta':    r'^%s\.theta$',
    'src':      r'^%s\.(new)?src(\.avg)?$',
    'grad':     r'^%s(?P<iter>[0-9]*)\.gvp[a-z]?(?P<freq>[0-9]*\.?[0-9]+)?.*$',
    'data':     r'^%s\.(ut|vz|vx)[ifoOesrcbt]+(?P<freq>[0-9]*\.?[0-9]+).*$',
    'diff':     r'^%s\.ud[ifoOesrcbt]+(?P<freq>[0-9]*\.?[0-9]+).*$',
    'wave':     r'^%s(?P<iter>[0-9]*)\.(wave|bwave)(?P<freq>[0-9]*\.?[0-9]+).*$',
    'slice':    r'^%s\.sl(?P<iter>[0-9]*)',
}


class BaseDatastore(object):

    def __init__(self, projnm):
        pass

    @property
    def systemConfig(self):
        raise NotImplementedError


class FullwvDatastore(BaseDatastore):
    """A FULLWV/OMEGA project directory: `<projnm>.ini` plus SEG-Y model and data files recognised by name
    (db.py:81-271).  `projnm` may carry a directory part; files are looked up next to the ini file."""

    def __init__(self, projnm):
        self.dirname, self.projnm = os.path.split(projnm)
        self._path = projnm
        inifile = '%s.ini' % projnm
        if not os.path.isfile(inifile):
            raise Exception('Project file %s does not exist' % (inifile,))
        self.ini = readini(inifile)

        redict = compileDict(re.escape(self.projnm), ftypeRegex)
        self.keepers = {key: {} for key in redict}
        for path in glob.glob(os.path.join(self.dirname, '*')):
            fn = os.path.basename(path)
            for key in redict:
                match = redict[key].match(fn)
                if match is not None:
                    self.keepers[key][fn] = match.groupdict()
                    break
        self.handled = {}
        for ftype in self.keepers:
            for fn in self.keepers[ftype]:
                self.handled[fn] = self.handle(ftype, fn)

    def sfWrapper(self, filename):
        return SEGYFile(os.path.join(self.dirname, filename))

    def handle(self, ftype, filename):
        return self.sfWrapper(filename)

    def _key(self, key):
        return key if key.find(self.projnm) == 0 else self.projnm + key

    def __getitem__(self, item):
        if isinstance(item, str):
            key, sl = item, slice(None)
        elif isinstance(item, tuple):
            assert len(item) == 2
            key, sl = item
            assert isinstance(key, str) and isinstance(sl, (slice, int))
        else:
            raise TypeError()
        key = self._key(key)
        if key in self.handled:
            return self.handled[key][sl]
        raise KeyError(key)

    def __contains__(self, key):
        return self._key(key) in self.handled

    def keys(self):
        return list(self.handled.keys())

    def __repr__(self):
        return '<%s(%s) comprising %d files>' % (self.__class__.__name__, self.projnm, len(self.handled))

    @property
    def systemConfig(self):
        'the flat configuration dict every Problem/Survey/Disc is built from (db.py:160-236)'
        ini = self.ini
        sc = {key: ini[key] for key in ('nx', 'nz', 'dx', 'dz', 'xorig', 'zorig', 'freqs', 'nky')}
        sc['ireg'] = ini['isreg']
        sc['freqBase'] = ini['freqbase']
        sc['tau'] = ini['tau'] if abs(float(ini['tau']) - 999.999) > 1e-2 else np.inf
        sc['freeSurf'] = (ini['fst'], ini['fsr'], ini['fsb'], ini['fsl'])

        ncol = ini['srcs'].shape[1]
        if ncol <= 3:
            srcGeom, recGeom = ini['srcs'][:, :2], ini['recs'][:, :2]
        elif ncol == 4:
            srcGeom, recGeom = ini['srcs'][:, ::2], ini['recs'][:, ::2]
        else:
            raise Exception('Something went wrong!')
        sc['geom'] = {'src': srcGeom, 'rec': recGeom, 'mode': 'fixed'}

        for fn, key, tf in (('.vp', 'c', None), ('.qp', 'Q', 'inv'), ('.rho', 'rho', None), ('.eps2d', 'eps', None),
                            ('.del2d', 'delta', None), ('.theta', 'theta', None)):
            if fn in self:
                arr = self[fn].T
                sc[key] = 1. / arr if tf == 'inv' else arr

        if '.src' in self:
            src = self['.src']
            nsrc = srcGeom.shape[0]
            ns = 2 * len(sc['freqs'])
            if src.shape[0] != 1 and src.shape[0] != nsrc:
                print('Source nsrc does not match project nsrc; using first term for all sources')
                src = src[:0, :]
            assert src.shape[1] == ns, 'Source ns does not match computed ns'
            a = src.T
            sterms = dftreal(a, a.shape[0], a.shape[1]).T
            sc['sterms'] = sterms[:, 1:ns // 2 + 1].T

        sc['projnm'] = self._path
        return sc

    def dataFiles(self, ftype):
        dKeep = self.keepers['data']
        fns = [fn for fn in dKeep if fn.find(ftype) > -1]
        ffreqs = [float(dKeep[fn]['freq']) for fn in fns]
        order = np.argsort(ffreqs)
        return [fns[i] for i in order], [ffreqs[i] for i in order]

    def spoolData(self, fid=slice(None), ftype='utobs'):
        'generator of complex (nrec, nsrc) observed-data panels for the requested frequencies (db.py:247-260)'
        ifreqs = np.atleast_1d(self.ini['freqs'][fid])
        fns, ffreqs = self.dataFiles(ftype)
        sffreqs = ['%0.3f' % freq for freq in ffreqs]
        try:
            finds = [sffreqs.index('%0.3f' % freq) for freq in ifreqs]
        except ValueError as e:
            raise ValueError('Could not find data from all requested frequencies: %s' % e)
        for fi in finds:
            fdata = self[fns[fi]]
            yield fdata[::2].T + 1j * fdata[1::2].T

    def utoutWrite(self, data, fid=slice(None), ftype='utout'):
        UtoutWriter(self.systemConfig)(data, fid, ftype)


class FlatDatastore(BaseDatastore):
    'configuration from `<projnm>.py`, which must define `systemConfig` (db.py:280-298)'

    def __init__(self, projnm):
        with open('%s.py' % (projnm,), 'r') as fp:
            contents = fp.read()
        scope = {}
        exec(contents, scope)
        self._systemConfig = scope['systemConfig']

    @property
    def systemConfig(self):
        return self._systemConfig

    @systemConfig.setter
    def systemConfig(self, value):
        self._systemConfig = value


class PickleDatastore(FlatDatastore):
    'configuration from `<projnm>.pickle` (db.py:301-310)'

    def __init__(self, projnm):
        with open('%s.pickle' % (projnm,), 'rb') as fp:
            self._systemConfig = pickle.Unpickler(fp).load()
