"""Import alias: the package lives in the directory `avsr-tf1_amd/` (not a valid Python identifier).

`import avsr_tf1_amd` (optionally `as avsr`) exposes the reference's Python surface
(`AVSR`, `run_experiment`) backed by the MI355X HIP engine.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "avsr-tf1_amd")]
__package__ = __name__           # make the module a package so relative imports inside __init__ work
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
