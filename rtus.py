"""Import alias: ``import rtus`` -> the package in ``ray-tracing-ultrasound_amd/`` (whose directory
name is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("ray-tracing-ultrasound_amd")
