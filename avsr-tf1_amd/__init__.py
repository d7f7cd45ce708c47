"""avsr_tf1_amd -- MI355X (gfx950) engine for the AVSR seq2seq hot path of georgesterpu/avsr-tf1.

Python host code mirroring the reference's surface (avsr/__init__.py:1-3), hand-written HIP kernels
behind the C ABI in include/avsr_hip.h.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def __getattr__(name):
    """Lazy exports of the reference's public surface (avsr/__init__.py:1-3): AVSR, run_experiment (+ run_experiment_mixedsnrs), LM."""
    if name == "AVSR":
        from .avsr import AVSR
        return AVSR
    if name == "run_experiment":
        from .experiment import run_experiment
        return run_experiment
    if name == "run_experiment_mixedsnrs":
        from .experiment import run_experiment_mixedsnrs
        return run_experiment_mixedsnrs
    if name == "LM":
        from .lm import LM
        return LM
    raise AttributeError(name)
