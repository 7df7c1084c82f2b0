"""`future.standard_library.install_aliases()` is a no-op on Python 3."""


def install_aliases():
    return None
