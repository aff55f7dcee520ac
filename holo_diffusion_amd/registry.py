"""The Implicitron registry/config surface the plugins of this package hang off.

The reference plugs its denoiser / implicit function / renderer / model into
``pytorch3d.implicitron.tools.config`` (``@registry.register``, ``registry.get(Base, name)``,
``ReplaceableBase``, ``Configurable``, ``<member>_class_type`` + ``<member>_<Type>_args``;
e.g. /root/reference/holo_diffusion/utils/diffusion_utils.py:41-42,
holo_diffusion_model.py:44-55,118-130).

Two situations, decided once at import time:

* **PyTorch3D importable** - ``Configurable`` / ``ReplaceableBase`` ARE PyTorch3D's classes and the plugin base classes
  (``ImplicitronModelBase``, ``BaseRenderer``, ``ImplicitFunctionBase``) are PyTorch3D's own when their modules import
  (see :func:`pt3d_base`), so the plugin classes are real members of the Implicitron class tree.
  ``registry.register`` then ALSO registers the class in PyTorch3D's registry - errors propagate, nothing is swallowed -
  and ``experiment.py`` / ``generate_samples.py`` resolve ``*_class_type: HoloDiffusionModel`` etc. to the HIP-backed
  classes.  PyTorch3D dataclass-processes a ``Configurable`` on first instantiation (``expand_args_fields``); the
  plugin classes survive that because they define their own ``__init__(**config)`` (``dataclasses`` never overwrites a
  class-defined ``__init__``) and all their config fields are annotated class attributes with immutable defaults.
* **PyTorch3D absent** (the build and GPU boxes today) - minimal local stand-ins with the same names and the same
  lookup semantics.

Either way the package-level ``registry`` keeps its own ``(base class -> name -> class)`` table, which is what the
classes of this package use to resolve each other.
"""
from __future__ import annotations

import dataclasses
import enum
import importlib
from typing import Any, Dict, Optional, Type


def _import_or_none(module: str):
    try:
        return importlib.import_module(module)
    except Exception:  # ImportError, or a half-installed PyTorch3D failing in its own imports
        return None


_P3D_CONFIG = _import_or_none("pytorch3d.implicitron.tools.config")
HAVE_PYTORCH3D = _P3D_CONFIG is not None and all(
    hasattr(_P3D_CONFIG, n) for n in ("registry", "ReplaceableBase", "Configurable"))

if HAVE_PYTORCH3D:
    Configurable = _P3D_CONFIG.Configurable
    ReplaceableBase = _P3D_CONFIG.ReplaceableBase
else:
    class Configurable:  # type: ignore[no-redef]
        """Marker base: classes whose public annotated fields are configuration (dataclass-like)."""

    class ReplaceableBase:  # type: ignore[no-redef]
        """Marker base for pluggable implementations selected by ``<member>_class_type``."""


def pt3d_base(module: str, name: str, bases=(ReplaceableBase,)) -> Type:
    """The plugin base class ``name``: PyTorch3D's own (``pytorch3d.<module>.<name>``) when importable, so that
    ``registry.get(<pt3d base>, "<plugin>")`` issued by Implicitron's factories finds the plugin; otherwise a local
    stand-in deriving DIRECTLY from ``ReplaceableBase`` (Implicitron's registry keys on exactly such classes)."""
    if HAVE_PYTORCH3D:
        m = _import_or_none("pytorch3d." + module)
        if m is not None and hasattr(m, name):
            return getattr(m, name)
    return type(name, tuple(bases), {"__doc__": f"stand-in for pytorch3d.{module}.{name}", "__module__": __name__})


def _is_config_class(klass: Any) -> bool:
    return isinstance(klass, type) and (issubclass(klass, Configurable) or issubclass(klass, ReplaceableBase)) \
        and klass not in (Configurable, ReplaceableBase)


def config_fields(cls: Type) -> Dict[str, Any]:
    """Public annotated class attributes of the Configurable / ReplaceableBase part of ``cls``'s MRO, base-first:
    the configuration of the class (name -> default)."""
    out: Dict[str, Any] = {}
    for klass in reversed(cls.__mro__):
        if not _is_config_class(klass):
            continue
        for k in klass.__dict__.get("__annotations__", {}):
            if k.startswith("_"):
                continue
            if k in klass.__dict__:
                v = klass.__dict__[k]
                if isinstance(v, dataclasses.Field):  # after PyTorch3D's dataclass processing
                    v = v.default if v.default is not dataclasses.MISSING else (
                        v.default_factory() if v.default_factory is not dataclasses.MISSING else None)
                out[k] = v
            elif k not in out and hasattr(klass, k):
                out[k] = getattr(klass, k)
    return out


def _replaceable_bases(cls: Type):
    """Classes in ``cls``'s MRO that derive DIRECTLY from ReplaceableBase: the keys Implicitron registers under."""
    return [b for b in cls.__mro__[1:] if isinstance(b, type) and ReplaceableBase in getattr(b, "__bases__", ())]


class _Registry:
    def __init__(self):
        self._by_base: Dict[Type, Dict[str, Type]] = {}

    def register_local(self, cls: Type) -> Type:
        """Package-local registration ONLY.  For config-carrying classes that share their name with a PyTorch3D class
        they do not replace (e.g. ``AngleWeightedReductionFeatureAggregator``: here a parameter holder of the fused
        view-pooling kernel, in PyTorch3D an ``nn.Module`` other Implicitron models use).  PyTorch3D's registry maps
        (base, name) -> class and overwrites silently, so such a class must never be pushed into it."""
        return self.register(cls, _pytorch3d=False)

    def register(self, cls: Type, _pytorch3d: bool = True) -> Type:
        bases = _replaceable_bases(cls)
        if not bases:
            raise ValueError(f"{cls.__name__} does not derive from a direct subclass of ReplaceableBase")
        for b in bases:
            self._by_base.setdefault(b, {})[cls.__name__] = cls
        if HAVE_PYTORCH3D and _pytorch3d:
            # the real Implicitron registry: a failure here means experiment.py would NOT resolve to this class,
            # so it is an error, not something to hide
            _P3D_CONFIG.registry.register(cls)
        return cls

    def get(self, base: Type, name: str) -> Type:
        try:
            return self._by_base[base][name]
        except KeyError:
            pass
        for b, table in self._by_base.items():  # a subclass of a registered base may be asked for as well
            if name in table and isinstance(base, type) and issubclass(table[name], base):
                return table[name]
        raise ValueError(f"{name} has not been registered as a {getattr(base, '__name__', base)}")

    def get_all(self, base: Type):
        return list(self._by_base.get(base, {}).values())


registry = _Registry()


def get_default_args(cls: Type) -> Dict[str, Any]:
    """Default config of a Configurable class as a plain dict (OmegaConf-free)."""
    out = {}
    for k, v in config_fields(cls).items():
        if dataclasses.is_dataclass(v) and not isinstance(v, type):
            v = dataclasses.asdict(v)
        out[k] = list(v) if isinstance(v, tuple) else v
    return out


def apply_config(obj: Any, kwargs: Dict[str, Any]) -> None:
    """Set config fields from kwargs, rejecting unknown keys like the Implicitron dataclasses do."""
    fields = config_fields(type(obj))
    for k, v in kwargs.items():
        if k not in fields:
            raise TypeError(f"{type(obj).__name__} got an unexpected config field '{k}'")
        default = fields[k]
        if isinstance(default, tuple) and isinstance(v, (list, tuple)):
            v = tuple(v)
        if isinstance(default, enum.Enum) and isinstance(v, str):  # YAML carries enum fields by member name
            v = type(default)[v]
        if v is not None and not isinstance(v, (str, bytes)) and hasattr(v, "items") and not isinstance(v, dict):
            v = {kk: vv for kk, vv in v.items()}  # OmegaConf DictConfig -> plain dict
        setattr(obj, k, v)
    for k, v in fields.items():
        if k not in kwargs:
            setattr(obj, k, v)


def pytorch3d_registered(cls: Type) -> Optional[bool]:
    """True/False: ``cls`` resolves through PyTorch3D's registry under each of its replaceable bases; None without
    PyTorch3D."""
    if not HAVE_PYTORCH3D:
        return None
    try:
        return all(_P3D_CONFIG.registry.get(b, cls.__name__) is cls for b in _replaceable_bases(cls))
    except Exception:
        return False
