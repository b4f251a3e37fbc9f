"""mi355_decode: MI355X-native batched-decode engine (host side).

Importing the package binds libmi355_decode.so; that raises if the library has
not been built -- there is no fallback compute path.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

__version__ = "0.1.0"
