"""Importable alias of the package directory ``fused-4-bit-dequantize-linear-cuda-kernel_amd/``
(its name is not a valid Python identifier).  ``import fused_int4_amd`` gives the package;
``from fused_int4_amd import ops, QuantizedLinear`` and submodule imports work as usual."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "fused-4-bit-dequantize-linear-cuda-kernel_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
