"""Stand-in for `pygeo.segyread` (external, absent here)  --  test infrastructure.

A deliberately plain SEG-Y rev-0/1 trace reader (3200-byte text header, 400-byte binary header with the
samples-per-trace word at byte 3221 and the format code at byte 3225, 240-byte trace headers), one trace at a
time with `struct`, so that the reference's FullwvDatastore (db.py:81-271) can run in the dev container.  It is
independent of the vectorised decoder in zephyr_amd/omega.py that the tests compare it with.
"""
import struct
import numpy as np


def _ibm(word):
    if word & 0x00ffffff == 0:
        return 0.0
    sign = -1.0 if word >> 31 else 1.0
    return sign * ((word & 0x00ffffff) / 16777216.0) * 16.0 ** (((word >> 24) & 0x7f) - 64)


class SEGYFile(object):
    def __init__(self, filename, *a, **k):
        with open(filename, 'rb') as fp:
            raw = fp.read()
        ns, = struct.unpack('>H', raw[3220:3222])
        fmt, = struct.unpack('>H', raw[3224:3226])
        tlen = 240 + 4 * ns
        ntr = (len(raw) - 3600) // tlen
        out = np.zeros((ntr, ns))
        for t in range(ntr):
            body = raw[3600 + t * tlen + 240:3600 + (t + 1) * tlen]
            if fmt == 1:
                out[t] = [_ibm(w) for w in struct.unpack('>%dI' % ns, body)]
            elif fmt == 5:
                out[t] = struct.unpack('>%df' % ns, body)
            else:
                raise NotImplementedError('SEG-Y format code %d' % fmt)
        self._traces = out
        self.ns, self.ntr = ns, ntr

    def __getitem__(self, sl):
        return self._traces[sl]
