"""Import shim: exposes the package directory ``light-loam_amd/`` as the package ``lightloam_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "light-loam_amd")]
__package__ = "lightloam_amd"
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
