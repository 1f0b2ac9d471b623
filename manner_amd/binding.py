"""``manner_amd.install()`` — bind the HIP operator classes INTO the reference's own package.

The reference selects its operators by Python import path: its LightningModules say
``from manner.models.components.news_encoder import MannerNewsEncoder`` (reference manner/models/cr_module.py:13-16,
a_module.py:15, ensemble_module.py:13, baselines/nrms_plm_module.py:15-16).  ``install()`` imports the REFERENCE's
``manner.models.components.{news_encoder, attention, user_encoder, click_predictors}`` modules as they are and replaces, inside
them, exactly the classes this build mirrors:

    news_encoder      MannerTextEncoder, MannerEntityEncoder, MannerNewsEncoder, PLMTextEncoder
    attention         AdditiveAttention
    user_encoder      NAMLUserEncoder, NRMSUserEncoder
    click_predictors  DotProduct

Everything else of those files — ``NAMLNewsEncoder``, ``LSTURNewsEncoder``, ``MINERNewsEncoder``, ``CAUMNewsEncoder``,
``PolyAttention``, ``TargetAwareAttention``, ``DenseAttention``, ``LSTURUserEncoder``, ``CAUMUserEncoder``, ``MINSUserEncoder`` —
stays the reference's own torch code, and so do the names those classes captured at import time (the reference's
``NAMLNewsEncoder`` keeps using the reference's ``AdditiveAttention``: its module-level binding inside ``news_encoder`` is an import
of the ORIGINAL class object and is only replaced where it is one of the listed names of that module).  ``manner.utils``,
``manner.data``, ``manner.models.cr_module`` … are never touched: there is no ``manner`` package in this repository that could
shadow them (rounds 1-3 shipped a four-file shim package ``manner/``; it shadowed the reference tree and was removed).

Modules of the reference imported BEFORE ``install()`` (say ``manner.models.cr_module``, which has already executed its
``from … import MannerNewsEncoder``) are patched too: every module named ``manner`` / ``manner.*`` whose globals hold one of the
replaced class objects — under any alias, e.g. ``UserEncoder`` — is rebound, so the call order does not matter.

Use (no reference source line changes):

    python -m manner_amd.run manner/train.py experiment=cr_module_mind_title_scl_lf      # = install() + runpy of the script

or two lines at the top of ``manner/train.py`` / ``manner/eval.py``:  ``import manner_amd; manner_amd.install()``.
"""
from __future__ import annotations

import importlib
import sys
from typing import Dict, List, Optional

TARGETS = {
    "news_encoder": ("MannerTextEncoder", "MannerEntityEncoder", "MannerNewsEncoder", "PLMTextEncoder"),
    "attention": ("AdditiveAttention",),
    "user_encoder": ("NAMLUserEncoder", "NRMSUserEncoder"),
    "click_predictors": ("DotProduct",),
}
_REF_PKG = "manner.models.components"
_MIRROR_PKG = "manner_amd.models.components"
_installed: Dict[str, Dict[str, type]] = {}          # module name -> {class name: the reference's original class}


def install(reference_root: Optional[str] = None) -> Dict[str, List[str]]:
    """Rebind the mirrored classes inside the reference's ``manner.models.components`` modules.  ``reference_root``: a checkout
    of andreeaiana/manner to put on ``sys.path`` when ``manner`` is not importable yet.  Returns {module: [rebound names]} (also
    the aliases patched in already-imported ``manner.*`` modules).  Idempotent; ``uninstall()`` restores the originals."""
    if reference_root and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    report: Dict[str, List[str]] = {}
    swaps = {}                                       # id(original class) -> mirror class
    for leaf, names in TARGETS.items():
        try:
            ref_mod = importlib.import_module(f"{_REF_PKG}.{leaf}")
        except ModuleNotFoundError as e:
            if (e.name or "").split(".")[0] == "manner":
                raise ModuleNotFoundError(
                    f"manner_amd.install(): the reference package `manner` is not importable ({e}); put a checkout of andreeaiana/manner on "
                    "PYTHONPATH or pass reference_root=") from e
            raise
        if getattr(ref_mod, "__file__", "").startswith(__file__.rsplit("/manner_amd/", 1)[0] + "/manner/"):
            raise RuntimeError("manner_amd.install(): `manner` resolves to a stale shim package inside this repository, not to the reference")
        mirror_mod = importlib.import_module(f"{_MIRROR_PKG}.{leaf}")
        kept = _installed.setdefault(ref_mod.__name__, {})
        for n in names:
            mirror = getattr(mirror_mod, n)
            cur = getattr(ref_mod, n)
            if cur is mirror:
                continue
            kept.setdefault(n, cur)
            swaps[id(cur)] = mirror
            setattr(ref_mod, n, mirror)
            report.setdefault(ref_mod.__name__, []).append(n)
    # aliases captured by reference modules that were imported before install(): patch by object identity
    if swaps:
        own = {f"{_REF_PKG}.{leaf}" for leaf in TARGETS}
        for name, mod in list(sys.modules.items()):
            if mod is None or not (name == "manner" or name.startswith("manner.")) or name in own:
                continue
            for attr, val in list(vars(mod).items()):
                if isinstance(val, type) and id(val) in swaps:
                    _installed.setdefault(name, {}).setdefault(attr, val)
                    setattr(mod, attr, swaps[id(val)])
                    report.setdefault(name, []).append(attr)
    return report


def uninstall() -> None:
    """Put the reference's own classes back (tests)."""
    for modname, kept in _installed.items():
        mod = sys.modules.get(modname)
        if mod is not None:
            for n, orig in kept.items():
                setattr(mod, n, orig)
    _installed.clear()


def installed() -> Dict[str, List[str]]:
    return {m: sorted(k) for m, k in _installed.items() if k}
