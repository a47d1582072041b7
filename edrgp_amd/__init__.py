"""Import alias: ``import edrgp_amd`` loads the package whose files live in
``edr-gp_amd/`` (the directory name the project layout asks for is not a valid
Python identifier).  Everything is imported once, under the ``edrgp_amd`` name."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "edr-gp_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
