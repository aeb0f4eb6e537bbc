"""``from models import Mamba, Transformer`` -- the import the reference's analysis/eval_eig.py:24-25 and launch.py
perform -- resolved to the MI355X drop-in models of this repository (package directory has hyphens, hence importlib).

Only the torch families that the analysis path forwards through are provided.  The JAX/flax SSM constructors
(init_LRU / init_S4 / init_S5, BatchClassificationModel) need jax + flax, which this image does not have: asking for
them raises with a pointer to the NumPy initialisers of the analysis module (conf_args["ssm_init"] = "numpy")."""
import importlib as _importlib

_pkg = _importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd.models")

MATCH, MLP, GLU, LAMBDA = _pkg.MATCH, _pkg.MLP, _pkg.GLU, _pkg.LAMBDA
ClassifierHead, TokenEmbeddings = _pkg.ClassifierHead, _pkg.TokenEmbeddings
MHA, MHNA = _pkg.MHA, _pkg.MHNA
Mamba, Transformer = _pkg.Mamba, _pkg.Transformer


def __getattr__(name):
    if name in ("init_LRU", "init_S4", "init_S5", "LRU", "S4", "S5SSM", "BatchClassificationModel"):
        raise ImportError(f"models.{name} is the reference's JAX/flax code path (jax / flax are not installed here). For the "
                          "eigenvalue analysis set `ssm_init: numpy` in the analysis YAML (NumPy draws of log_step / nu / "
                          "theta from the reference's distributions; see analysis.eval_eig.init_layers_ssm_numpy)")
    raise AttributeError(name)
