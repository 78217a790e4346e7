"""Import the reference's own PIPS / SamPt modules IN PLACE from /root/reference (read-only).

TEST INFRASTRUCTURE ONLY (see oracle/pips_ref.py header).  Nothing is copied: the reference
packages' ``__init__`` files import cv2 / tensorflow / cotracker (absent here), so empty namespace
modules are pre-registered in ``sys.modules`` and the few needed leaf modules are imported by path —
the recipe verified in SURVEY.md Appendix C.  Only usable where /root/reference exists (the build
container); on the GPU box callers must check ``available()`` first.
"""
from __future__ import annotations

import os
import sys
import types

REF = os.environ.get("SAMPT_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "sam_pt", "point_tracker", "pips"))


def _ns(name: str, rel: str):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = [REF + rel]
    sys.modules[name] = m
    return m


def _link_children():
    for name in list(sys.modules):
        if name.startswith("sam_pt."):
            parent, _, child = name.rpartition(".")
            if parent in sys.modules:
                setattr(sys.modules[parent], child, sys.modules[name])


def load_pips():
    """Returns (Pips class, PipsPointTracker class, PointTracker ABC) of the reference."""
    assert available(), "reference tree not present"
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
    for n, p in [("sam_pt", "/sam_pt"), ("sam_pt.point_tracker", "/sam_pt/point_tracker"),
                 ("sam_pt.point_tracker.utils", "/sam_pt/point_tracker/utils"),
                 ("sam_pt.point_tracker.pips", "/sam_pt/point_tracker/pips"),
                 ("sam_pt.utils", "/sam_pt/utils"), ("sam_pt.modeling", "/sam_pt/modeling")]:
        _ns(n, p)
    import sam_pt.point_tracker.tracker as T
    sys.modules["sam_pt.point_tracker"].PointTracker = T.PointTracker
    import sam_pt.point_tracker.pips.pips as P
    sys.modules["sam_pt.point_tracker.pips"].Pips = P.Pips
    import sam_pt.point_tracker.utils.saverloader as SL
    sys.modules["sam_pt.point_tracker.utils"].saverloader = SL
    _link_children()
    import sam_pt.point_tracker.pips.tracker as PT
    _link_children()
    return P.Pips, PT.PipsPointTracker, T.PointTracker


def load_pips2():
    """Returns (PipsPlusPlus class, PipsPlusPlusPointTracker class) of the reference.  The model hard-codes a
    ``.cuda()`` (pips_plus_plus.py:438) and the tracker several more; on a CPU-only box ``Tensor.cuda`` is shimmed to
    the identity for the lifetime of the process (test infrastructure only)."""
    import torch
    load_pips()
    _ns("sam_pt.point_tracker.pips_plus_plus", "/sam_pt/point_tracker/pips_plus_plus")
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    import sam_pt.point_tracker.utils.basic  # noqa: F401
    import sam_pt.point_tracker.utils.misc  # noqa: F401
    import sam_pt.point_tracker.utils.samp  # noqa: F401
    import sam_pt.point_tracker.pips_plus_plus.pips_plus_plus as P2
    sys.modules["sam_pt.point_tracker.pips_plus_plus"].PipsPlusPlus = P2.PipsPlusPlus
    _link_children()
    import sam_pt.point_tracker.pips_plus_plus.tracker as PT2
    _link_children()
    return P2.PipsPlusPlus, PT2.PipsPlusPlusPointTracker


def load_sam_pt():
    """Returns the reference ``SamPt`` class (sam_pt/modeling/sam_pt.py) with absent third-party imports
    stubbed (segment_anything, skimage, cv2, wandb, sklearn_extra)."""
    load_pips()

    def stub(name, **attrs):
        if name not in sys.modules:
            import importlib.machinery
            m = types.ModuleType(name)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)  # keeps importlib.util.find_spec() working
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(sys.modules[name], k, v)
        return sys.modules[name]

    stub("segment_anything", SamPredictor=object)
    stub("segment_anything.modeling", Sam=object)
    stub("skimage", color=types.SimpleNamespace())
    stub("cv2")
    stub("wandb")
    stub("sklearn_extra")
    stub("sklearn_extra.cluster", KMedoids=object)
    stub("matplotlib") if "matplotlib" not in sys.modules else None

    class _NoSuperGlue:  # isinstance() target only (sam_pt.py:189)
        pass

    sys.modules["sam_pt.point_tracker"].SuperGluePointTracker = _NoSuperGlue
    import importlib
    try:
        U = importlib.import_module("sam_pt.utils.util")
    except Exception:
        # util.py pulls cv2/wandb/matplotlib at import; only the enum is needed on the hot path
        from enum import IntEnum

        class PointVisibilityType(IntEnum):  # values: sam_pt/utils/util.py:267-282
            VISIBLE = 1
            INVISIBLE = 0
            REINIT_FAILED = -1
            OUTSIDE_FRAME = -2
            PATCH_NON_SIMILAR = -3
            REJECTED_AFTER_PATCH_WAS_NON_SIMILAR = -4

        U = stub("sam_pt.utils.util", PointVisibilityType=PointVisibilityType)
    if "sam_pt.utils.query_points" not in sys.modules:
        try:
            importlib.import_module("sam_pt.utils.query_points")
        except Exception:
            def _na(*a, **k):
                raise NotImplementedError("query-point selection needs cv2/sklearn_extra (absent)")
            stub("sam_pt.utils.query_points", extract_kmedoid_points=_na, extract_random_mask_points=_na,
                 extract_corner_points=_na, extract_mixed_points=_na)
    _link_children()
    M = importlib.import_module("sam_pt.modeling.sam_pt")
    return M.SamPt


def load_evaluator():
    """Returns the reference ``SamPtEvaluator`` class (sam_pt/vos_eval/evaluator.py:47-60)."""
    load_sam_pt()
    _ns("sam_pt.vos_eval", "/sam_pt/vos_eval")
    import importlib
    E = importlib.import_module("sam_pt.vos_eval.evaluator")
    _link_children()
    return E.SamPtEvaluator


def load_demo():
    """Returns the reference ``demo.demo`` module (for ``run_inference``, demo/demo.py:114-155) with its control-plane
    imports (hydra, omegaconf, cv2, wandb, matplotlib, the visualisation helper) stubbed."""
    import importlib
    import importlib.machinery
    load_sam_pt()

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(sys.modules[name], k, v)
        return sys.modules[name]

    if "hydra" not in sys.modules:
        stub("hydra", main=lambda **kw: (lambda fn: fn))           # @hydra.main(...) -> identity decorator
        stub("hydra.core")
        stub("hydra.core.hydra_config", HydraConfig=object)
        stub("hydra.utils", instantiate=None)
    if "omegaconf" not in sys.modules:
        stub("omegaconf", OmegaConf=object)
    for name in ("cv2", "wandb"):
        stub(name)
    try:
        importlib.import_module("matplotlib.pyplot")
    except Exception:
        stub("matplotlib")
        stub("matplotlib.pyplot")
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    util = sys.modules.get("sam_pt.utils.util")
    if util is not None and not hasattr(util, "visualize_predictions"):
        util.visualize_predictions = None
    _ns("demo", "/demo")
    D = importlib.import_module("demo.demo")
    return D
