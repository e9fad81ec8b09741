"""Import shim: the product package lives in the directory `starkpack-winterfell_amd/` (a name Python cannot
import directly); `import starkpack_winterfell_amd` resolves to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "starkpack-winterfell_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
