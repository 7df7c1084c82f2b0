class SEGYFile(object):
    def __init__(self, *a, **k):
        raise NotImplementedError('SEG-Y reading is not part of the oracle')
