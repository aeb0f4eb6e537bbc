"""Repository-root shim so that the reference's ``launch.py`` (``from analysis import eval_eig``, launch.py:15) picks
up the MI355X-native implementation unchanged.  The implementation lives in the (hyphen-named, hence importlib)
package ``task-level-insights-from-eigenvalues-across-sequence-models_amd``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_impl = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd.analysis.eval_eig")
eval_eig = _impl.eval_eig

__all__ = ["eval_eig"]
