"""Drop-in for the reference's ``analysis`` package (reference analysis/__init__.py:4-6): ``from analysis import
eval_eig`` keeps working in launch.py (launch.py:15, :176, :183)."""
from .eval_eig import eval_eig

__all__ = ["eval_eig"]
