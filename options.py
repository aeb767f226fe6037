"""Shim: ``from options import args_parser`` as the reference's scripts do (main_both.py:8)."""
from garbage_classification_rca_amd.options import args_parser, build_parser  # noqa: F401
