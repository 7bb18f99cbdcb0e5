"""lattigo_amd -- MI355X (gfx950) ring-arithmetic backend for Lattigo-style RNS HE.

Host-side mirror (Python, over ctypes) of the reference's operator interfaces for
the hot path only: ``ring.Ring`` / ``ring.BasisExtender`` (``lattigo_amd.ring``)
and ``rlwe.Evaluator`` plus the CKKS/BGV call sites (``lattigo_amd.rlwe``).
All arithmetic runs in ``libhering.so`` (hand-written HIP); there is no CPU
fallback -- importing works without a GPU, creating a ``Context`` does not.
"""
from ._lib import HeringError, lib_path, load  # noqa: F401
from .ring import BasisExtender, Context, Graph, Poly, Ring, device_count, device_pci_bus_id  # noqa: F401
from .rlwe import Decomposition, EvaluationKey, Evaluator  # noqa: F401

__all__ = ["HeringError", "lib_path", "load", "device_count", "device_pci_bus_id", "Context", "Ring", "Poly", "BasisExtender", "Evaluator",
           "EvaluationKey", "Decomposition"]
